"""CPU tests of the multi-GPU data path minus the GPU: the pure tile partition, the slab layout and the
all-gather + unpack, run with torch.distributed's gloo backend at world_size 2 (and 3 ranks emulated in
one process).  The per-rank slab contents come from the CPU oracle standing in for the HIP kernel: what
is under test is the product's partition / layout / unpack logic, which is host code."""
import os
import socket

import numpy as np
import pytest

from audio_pattern_discovery_amd import sharding, synth


def oracle_slab(oracle, frames, offsets, pct, rank, world):
    """What apd_align_tiles_async writes, computed by the oracle: [tiles][2][16][16]."""
    n = len(offsets) - 1
    t = sharding.tile_size()
    slab = np.zeros(sharding.slab_floats(n, world), dtype=np.float32)
    seqs = synth.split(frames, offsets)
    order = sharding.length_order(offsets)                       # tiles are cut over the length-sorted positions
    for k, (ta, tb) in enumerate(sharding.rank_tiles(n, rank, world)):
        blk = slab[k * 2 * t * t:(k + 1) * 2 * t * t].reshape(2, t, t)
        for sa in range(t):
            for sb in range(t):
                pa, pb = int(ta) * t + sa, int(tb) * t + sb
                if pa < pb < n:
                    a, b = int(order[pa]), int(order[pb])
                    band = oracle.warping_band(pct, max(len(seqs[a]), len(seqs[b])))
                    blk[0, sa, sb] = oracle.dtw_pair(seqs[a], seqs[b], band)
                    blk[1, sa, sb] = oracle.dtw_pair(seqs[b], seqs[a], band)
    return slab


@pytest.mark.parametrize("n_seq,world", [(1, 1), (16, 1), (17, 2), (40, 3), (70, 8), (33, 5)])
def test_partition_is_a_partition(n_seq, world):
    t = sharding.tile_size()
    side = (n_seq + t - 1) // t
    assert sharding.num_tiles(n_seq) == side * (side + 1) // 2
    seen = []
    for r in range(world):
        tiles = sharding.rank_tiles(n_seq, r, world)
        assert len(tiles) * 2 * t * t <= sharding.slab_floats(n_seq, world)
        seen += [tuple(x) for x in tiles.tolist()]
    assert sorted(seen) == [(a, b) for a in range(side) for b in range(a, side)]     # every tile exactly once
    counts = [len(sharding.rank_tiles(n_seq, r, world)) for r in range(world)]
    assert max(counts) - min(counts) <= 1                                              # cyclic => balanced


def test_emulated_three_ranks_reassemble_the_matrix(oracle):
    frames, offsets = synth.make_sequences(37, 20, 5, seed=3, jitter=6)
    want = oracle.align_all(frames, offsets, 0.25, workers=4)
    gathered = np.concatenate([oracle_slab(oracle, frames, offsets, 0.25, r, 3) for r in range(3)])
    got = sharding.unpack_host(offsets, 3, gathered)
    assert np.array_equal(got, want)


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from oracle import binding as oracle
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    frames, offsets = synth.make_sequences(35, 18, 4, seed=11, jitter=5)      # same seed on every rank
    slab = torch.from_numpy(oracle_slab(oracle, frames, offsets, 0.0625, rank, world))
    gathered = torch.empty(world * slab.numel(), dtype=torch.float32)
    dist.all_gather_into_tensor(gathered, slab)                                 # the one collective of the path
    if rank == 0:
        got = sharding.unpack_host(offsets, world, gathered.numpy())
        want = oracle.align_all(frames, offsets, 0.0625, workers=2)
        q.put(bool(np.array_equal(got, want)))
    dist.barrier()
    dist.destroy_process_group()


def test_gloo_world_size_2_all_gather_and_unpack():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_length_order_is_longest_first_and_stable():
    lens = np.array([5, 9, 5, 1, 9, 7, 5], dtype=np.uint64)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    order = sharding.length_order(offsets)
    assert order.tolist() == [1, 4, 5, 0, 2, 6, 3]            # equal lengths keep ascending index
    assert sharding.length_order(np.zeros(1, dtype=np.uint64)).tolist() == []


def test_file_rendezvous_control_plane(tmp_path):
    """bench.py's process-per-GPU mode hands the 128-byte communicator id and the per-rank status lines around through a
    directory in /tmp (no torch, no sockets).  Three rank processes of one parent: everyone reads rank 0's blob, everyone sees
    everyone's status in rank order, the directory is gone afterwards."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import os, sys; sys.path.insert(0, %r); import bench\n"
        "rank, world = int(sys.argv[1]), 3\n"
        "r = bench.FileRendezvous(rank, world)\n"
        "if rank == 0: r.put('id', bytes(range(128)))\n"
        "blob = r.get('id', 60)\n"
        "st = r.gather_status('' if rank != 1 else 'rank 1: boom', timeout=60)\n"
        "print(len(blob), blob[5], st, r.dir)\n"
        "r.finish()\n" % root)
    env = dict(os.environ, MASTER_PORT="45991", APD_RDZV_DIR=str(tmp_path))
    env.pop("TORCHELASTIC_RUN_ID", None)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r)], stdout=subprocess.PIPE, text=True, env=env) for r in range(3)]
    outs = [p.communicate(timeout=120)[0].strip() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    dirs = set()
    for o in outs:
        n, b5, st, d = o.split(" ", 3)[0], o.split(" ", 3)[1], o[o.index("["):o.rindex("]") + 1], o.rsplit(" ", 1)[1]
        assert (n, b5) == ("128", "5") and st == "['', 'rank 1: boom', '']"
        dirs.add(d)
    assert len(dirs) == 1 and not os.path.exists(dirs.pop())            # one directory for the three ranks, removed by rank 0

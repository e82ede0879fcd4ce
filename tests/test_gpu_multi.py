"""The persistent multi-device handle (include/apd.h: apd_multi_*) on the one GPU a test box has: contexts, RCCL
communicators (ncclCommInitAll), worker threads, gather buffers and resident batches are made once; the second align_all
pays kernels + one all-gather + unpack only.  Replaces AlignmentWorkers { data, result } + align_all's thread fan-out
(reference src/alignments.rs:11-67) for N GPUs; the N > 1 branch proper runs on the driver's 8-GPU node (bench.py --gpus N)."""
import ctypes as C
import os
import time

import numpy as np
import pytest

from audio_pattern_discovery_amd import synth

pytestmark = pytest.mark.gpu


def assert_parity(got, want):
    assert np.array_equal(np.isfinite(got), np.isfinite(want))
    f = np.isfinite(want) & (want != 0)
    assert np.all(got[np.isfinite(want) & (want == 0)] == 0)
    assert np.max(np.abs(got[f] - want[f]) / np.abs(want[f])) <= 1e-4          # the north star's tolerance


def test_create_once_align_many(apd, oracle):
    """First call = handle + batch + alignment; the second alignment on the same handle must give the identical bits and be
    at least 3x faster (no communicator, context, batch or plan is rebuilt)."""
    from audio_pattern_discovery_amd import sharding
    from audio_pattern_discovery_amd.alignments import Batch
    n = 96
    frames, offsets = synth.make_sequences(n, 80, 13, seed=21)
    cfg = apd.AlignConfig(0.0625, 1.0, 1.0, 1.0)
    t0 = time.perf_counter()
    m = sharding.Multi([0])
    mb = m.batch(offsets, 13, frames=frames)
    first = m.align_all(mb, cfg)
    t_first = time.perf_counter() - t0
    t0 = time.perf_counter()
    second = m.align_all(mb, cfg)
    t_second = time.perf_counter() - t0
    assert np.array_equal(first.view(np.uint32), second.view(np.uint32))
    assert t_second * 3 <= t_first, "second call %.4f s vs first %.4f s" % (t_second, t_first)
    assert m.ranks_seen() == 1 and m.collective().startswith("rccl")
    assert_parity(first, oracle.align_all(frames, offsets, 0.0625, workers=8))
    # the same bits as the single-device entry point
    ctx = apd.Context(0)
    single = np.empty((n, n), np.float32)
    b = Batch(ctx, frames, offsets, 13)
    apd.check(apd.lib().apd_align_all(ctx.handle, b.handle, C.byref(cfg), single.ctypes.data_as(C.POINTER(C.c_float))), ctx.handle)
    assert np.array_equal(first.view(np.uint32), single.view(np.uint32))
    mb.close()
    m.close()


def test_device_resident_frames_refill_and_async(apd, oracle):
    """Frames resident on the handle's device (apd_multi_context + apd_device_alloc), refilled with new values of the same
    lengths, aligned asynchronously into a caller's buffer and into the handle's own result."""
    from audio_pattern_discovery_amd import sharding
    n = 50
    f1, offsets = synth.make_sequences(n, 64, 13, seed=5)
    f2 = (f1[::-1] * np.float32(0.5)).copy()
    cfg = apd.AlignConfig(0.0625, 1.0, 1.0, 1.0)
    m = sharding.Multi([0])
    c0 = m.contexts[0]
    d_f = c0.upload(f1)
    mb = m.batch(offsets, 13, d_frames=[d_f.ptr])
    d_out = c0.alloc(4 * n * n)
    m.align_all_async(mb, cfg, d_out.ptr)
    m.synchronize()
    assert_parity(d_out.to_numpy(np.float32).reshape(n, n), oracle.align_all(f1, offsets, 0.0625, workers=8))
    d_f.copy_from(f2)
    mb.refill(d_frames=[d_f.ptr])
    m.align_all_async(mb, cfg)                                          # result owned by the handle
    m.synchronize()
    assert m.result_ptr() != 0
    got = np.empty(n * n, np.float32)
    apd.check(apd.lib().apd_copy_to_host(c0.handle, C.c_void_p(got.ctypes.data), C.c_void_p(m.result_ptr()), got.nbytes), c0.handle)
    assert_parity(got.reshape(n, n), oracle.align_all(f2, offsets, 0.0625, workers=8))
    mb.refill(frames=f1)                                                # host frames work for a refill too
    assert_parity(m.align_all(mb, cfg), oracle.align_all(f1, offsets, 0.0625, workers=8))
    # per-device settings go through the handle's contexts: strict mode gives the oracle's bits
    c0.set_distance_mode("strict")
    strict = m.align_all(mb, cfg)
    assert np.array_equal(strict.view(np.uint32), oracle.align_all(f1, offsets, 0.0625, workers=8).view(np.uint32))
    m.close()                                                           # destroys the batch it still holds
    mb.close()


def test_peer_copy_fallback_gives_the_same_bits(apd):
    """APD_MULTI_COLLECTIVE=peer takes the path the handle falls back to when ncclCommInitAll fails: slabs gathered with
    hipMemcpyPeerAsync.  Same matrix, and the handle says which collective it runs."""
    from audio_pattern_discovery_amd import sharding
    frames, offsets = synth.make_sequences(70, 60, 13, seed=4)
    cfg = apd.AlignConfig(0.0625, 1.0, 1.0, 1.0)
    m = sharding.Multi([0])
    want = m.align_all(m.batch(offsets, 13, frames=frames), cfg)
    m.close()
    os.environ["APD_MULTI_COLLECTIVE"] = "peer"
    try:
        p = sharding.Multi([0])
    finally:
        del os.environ["APD_MULTI_COLLECTIVE"]
    assert p.collective().startswith("peer-copy fallback: APD_MULTI_COLLECTIVE=peer") and p.ranks_seen() == 1
    pb = p.batch(offsets, 13, frames=frames)
    got = p.align_all(pb, cfg)
    again = p.align_all(pb, cfg)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and np.array_equal(again.view(np.uint32), want.view(np.uint32))
    p.close()


def test_errors_come_back_with_the_device_and_the_text(apd):
    from audio_pattern_discovery_amd import sharding
    with pytest.raises(apd.ApdError) as e:
        sharding.Multi([0, 0])                                           # RCCL refuses two ranks on one device
    assert e.value.status == apd.APD_ERR_INVALID_ARG
    with pytest.raises(apd.ApdError) as e:
        sharding.Multi([0, 4096])
    assert e.value.status == apd.APD_ERR_NO_DEVICE
    m = sharding.Multi([0])
    frames, offsets = synth.make_sequences(8, 30, 13, seed=1)
    offsets = offsets.copy()
    offsets[3] = offsets[2]                                              # a zero-length sequence: alignments.rs:120 underflows
    mb = m.batch(offsets, 13, frames=frames)
    with pytest.raises(apd.ApdError) as e:
        m.align_all(mb, apd.AlignConfig(1.0, 1.0, 1.0, 1.0))
    assert e.value.status == apd.APD_ERR_EMPTY_SEQUENCE and "device 0" in str(e.value)
    m.close()


def test_one_shot_entry_point_is_the_handle(apd, oracle):
    from audio_pattern_discovery_amd import sharding
    from audio_pattern_discovery_amd.discovery import Discovery
    frames, offsets = synth.make_sequences(40, 50, 8, seed=9)
    cfg = Discovery(warping_band_percentage=0.25).align_config()
    got, seen = sharding.align_all_multi([0], frames, offsets, 8, cfg)
    assert seen == 1
    assert_parity(got, oracle.align_all(frames, offsets, 0.25, workers=8))


def test_several_ranks_on_one_gpu_through_the_peer_collective(apd, oracle):
    """The N > 1 machinery of the handle -- one worker thread, context, stream, tile share and slab per rank, the cross-stream
    ordering of the gather, the world-N unpack -- rehearsed on ONE GPU: with the peer-copy collective forced a device may be
    named several times (RCCL itself refuses two ranks on a device).  3 and 8 ranks, repeated calls, a refill in between: every
    matrix bitwise equal to the one-rank result."""
    from audio_pattern_discovery_amd import sharding
    n = 150
    f1, offsets = synth.make_sequences(n, 90, 13, seed=31, jitter=20)
    f2 = (f1 * np.float32(1.25) + np.float32(0.5)).astype(np.float32)
    cfg = apd.AlignConfig(0.0625, 1.0, 1.0, 1.0)
    one = sharding.Multi([0])
    ob = one.batch(offsets, 13, frames=f1)
    want1 = one.align_all(ob, cfg)
    ob.refill(frames=f2)
    want2 = one.align_all(ob, cfg)
    one.close()
    assert not np.array_equal(want1, want2)
    os.environ["APD_MULTI_COLLECTIVE"] = "peer"
    try:
        for world in (3, 8):
            m = sharding.Multi([0] * world)
            assert m.ranks_seen() == world and len(m.contexts) == world
            mb = m.batch(offsets, 13, frames=f1)
            for _ in range(3):
                assert np.array_equal(m.align_all(mb, cfg).view(np.uint32), want1.view(np.uint32))
            mb.refill(frames=f2)
            assert np.array_equal(m.align_all(mb, cfg).view(np.uint32), want2.view(np.uint32))
            # asynchronous form back to back (the slab of a rank must not be poisoned again before rank 0 has copied it)
            d_out = m.contexts[0].alloc(4 * n * n)
            for _ in range(4):
                m.align_all_async(mb, cfg, d_out.ptr)
            m.synchronize()
            assert np.array_equal(d_out.to_numpy(np.uint32).reshape(n, n), want2.view(np.uint32))
            m.close()
    finally:
        del os.environ["APD_MULTI_COLLECTIVE"]
    assert_parity(want1, oracle.align_all(f1, offsets, 0.0625, workers=8))


def test_more_ranks_than_tiles_and_degenerate_batches(apd, oracle):
    """24 sequences are 3 pair tiles: with 8 ranks five of them own nothing (their slabs stay poisoned and are never read);
    one sequence is a 1 x 1 zero matrix; an empty batch is a no-op."""
    from audio_pattern_discovery_amd import sharding
    cfg = apd.AlignConfig(1.0, 1.0, 1.0, 1.0)
    os.environ["APD_MULTI_COLLECTIVE"] = "peer"
    try:
        m = sharding.Multi([0] * 8)
    finally:
        del os.environ["APD_MULTI_COLLECTIVE"]
    frames, offsets = synth.make_sequences(24, 40, 13, seed=8)
    got = m.align_all(m.batch(offsets, 13, frames=frames), cfg)
    want = oracle.align_all(frames, offsets, 1.0, workers=8)
    assert_parity(got, want)
    one = m.align_all(m.batch(offsets[:2], 13, frames=frames[:int(offsets[1])]), cfg)
    assert one.shape == (1, 1) and one[0, 0] == 0.0
    empty = m.batch(np.zeros(1, np.uint64), 13, frames=np.zeros((0, 13), np.float32))
    assert m.align_all(empty, cfg).shape == (0, 0)
    m.close()


def test_matrix_stays_on_rank_0_for_the_upgma_leg(apd, oracle):
    """The N-GPU flow of main.rs:187-203: align_all over the handle leaves the matrix in devices[0]'s HBM (apd_multi_result) and
    apd_clustering runs there on the handle's own context -- nothing crosses PCIe between the two.  Against the oracle pipeline."""
    from audio_pattern_discovery_amd import sharding
    n = 64
    frames, offsets = synth.make_sequences(n, 60, 13, seed=12)
    cfg = apd.AlignConfig(0.0625, 1.0, 1.0, 1.0)
    os.environ["APD_MULTI_COLLECTIVE"] = "peer"
    try:
        m = sharding.Multi([0, 0, 0, 0])                                # four ranks on the one GPU of a test box
    finally:
        del os.environ["APD_MULTI_COLLECTIVE"]
    for c in m.contexts:
        c.set_distance_mode("strict")                                   # the oracle's bits, so that the dendrograms must agree exactly
    mb = m.batch(offsets, 13, frames=frames)
    m.align_all_async(mb, cfg)
    m.synchronize()
    c0 = m.contexts[0]
    ops = (apd.ClusterOp * n)()
    roots = np.zeros(n, dtype=np.uint32)
    n_ops, n_roots, thr = C.c_uint32(0), C.c_uint32(0), C.c_float(0)
    apd.check(apd.lib().apd_clustering(c0.handle, C.c_void_p(m.result_ptr()), 1, n, 0.3, ops, C.byref(n_ops),
                                       roots.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(n_roots), C.byref(thr)), c0.handle)
    want_d = oracle.align_all(frames, offsets, 0.0625, workers=8)
    want_ops, want_roots, want_thr = oracle.clustering(want_d, n, 0.3)
    assert [(ops[k].merge_i, ops[k].merge_j, ops[k].into) for k in range(n_ops.value)] == [(o["merge_i"], o["merge_j"], o["into"]) for o in want_ops]
    assert np.array_equal(np.array([ops[k].distance for k in range(n_ops.value)], np.float32).view(np.uint32),
                          np.array([o["distance"] for o in want_ops], np.float32).view(np.uint32))
    assert roots[:n_roots.value].tolist() == sorted(want_roots) and thr.value == want_thr
    m.close()


def test_alignment_workers_mirror_over_devices(apd, oracle):
    """The Python mirror of AlignmentWorkers with `devices`: new() makes the handle and the resident corpus once, align_all()
    is called twice (main.rs:187-195 in a loop) -- same bits as the single-GPU mirror."""
    from audio_pattern_discovery_amd.alignments import AlignmentWorkers, NDSequence
    from audio_pattern_discovery_amd.discovery import Discovery
    frames, offsets = synth.make_sequences(48, 70, 13, seed=14)
    seqs = [NDSequence(s) for s in synth.split(frames, offsets)]
    cfg = Discovery(warping_band_percentage=0.0625)
    single = AlignmentWorkers.new(seqs, apd.Context(0)).align_all(cfg).copy()
    w = AlignmentWorkers.new(seqs, devices=[0])
    first = w.align_all(cfg).copy()
    second = w.align_all(cfg).copy()
    w.close()
    assert np.array_equal(first.view(np.uint32), single.view(np.uint32)) and np.array_equal(second.view(np.uint32), single.view(np.uint32))
    assert_parity(first.reshape(48, 48), oracle.align_all(frames, offsets, 0.0625, workers=8))


def test_affinity_check_is_on_and_tells_contexts_on_one_device_apart(apd):
    """APD_DEBUG_AFFINITY=1 (tests/conftest.py): the library checks at every allocation / event / launch that the calling thread is
    bound to the CONTEXT it works on.  The probe binds one context and checks another: two contexts on the same GPU -- the
    rehearsal's "ranks" -- are told apart, which is what makes a worker thread that forgot to bind visible on a one-GPU box.
    Every other test of this file (3, 4 and 8 ranks on device 0) runs with the check armed and must not trip it."""
    L = apd.lib()
    a, b = apd.Context(0), apd.Context(0)
    on = C.c_int(0)
    assert L.apd_debug_affinity_probe(a.handle, a.handle, C.byref(on)) == apd.APD_OK and on.value == 1
    assert L.apd_debug_affinity_probe(a.handle, b.handle, C.byref(on)) == apd.APD_ERR_HIP
    assert b"APD_DEBUG_AFFINITY" in L.apd_last_error(b.handle) and b"another context" in L.apd_last_error(b.handle)
    # ... and ordinary use of both contexts, interleaved from one thread, passes: every entry point binds for itself
    x = np.arange(64, dtype=np.float32)
    da, db = a.upload(x), b.upload(x * 2)
    assert np.array_equal(da.to_numpy(np.float32), x) and np.array_equal(db.to_numpy(np.float32), x * 2)
    a.close()
    b.close()


def test_peer_collective_growing_batch_back_to_back(apd):
    """A small batch, then a larger one, through align_all_async with no synchronisation in between (peer-copy collective, 4 ranks
    on device 0): the larger slab makes every rank re-allocate its gather buffer while devices[0] may still be copying the previous
    slabs out of the old ones -- the handle waits for that gather first (csrc/comm.hip).  Both matrices must be right."""
    from audio_pattern_discovery_amd import sharding
    cfg = apd.AlignConfig(0.0625, 1.0, 1.0, 1.0)
    f_small, off_small = synth.make_sequences(40, 60, 13, seed=41)
    f_big, off_big = synth.make_sequences(400, 120, 13, seed=42)
    one = sharding.Multi([0])
    want_small = one.align_all(one.batch(off_small, 13, frames=f_small), cfg)
    want_big = one.align_all(one.batch(off_big, 13, frames=f_big), cfg)
    one.close()
    os.environ["APD_MULTI_COLLECTIVE"] = "peer"
    try:
        m = sharding.Multi([0] * 4)
    finally:
        del os.environ["APD_MULTI_COLLECTIVE"]
    b_small, b_big = m.batch(off_small, 13, frames=f_small), m.batch(off_big, 13, frames=f_big)
    d_small, d_big = m.contexts[0].alloc(4 * 40 * 40), m.contexts[0].alloc(4 * 400 * 400)
    for _ in range(3):
        m.align_all_async(b_small, cfg, d_small.ptr)
        m.align_all_async(b_big, cfg, d_big.ptr)                    # grows the gather buffers right behind the small call
    m.synchronize()
    assert np.array_equal(d_small.to_numpy(np.uint32).reshape(40, 40), want_small.view(np.uint32))
    assert np.array_equal(d_big.to_numpy(np.uint32).reshape(400, 400), want_big.view(np.uint32))
    m.close()

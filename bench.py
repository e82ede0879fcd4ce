#!/usr/bin/env python3
"""Benchmark of the hot path: all-pairs Sakoe-Chiba-banded DTW distance matrix (reference
src/alignments.rs:31-67) on synthetic MFCC-like sequences, one process per GPU.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one full pass of the path over the resident batch: repack the [sum len][D] frames into
the kernels' padded layout (apd_batch_refill), fused-pair DTW over this rank's pair tiles, ONE
ncclAllGather of the packed tile slabs, unpack into the N x N matrix -- the last three are ONE call,
apd_align_all_sharded_async, on a communicator the LIBRARY owns (apd_comm_create: RCCL over xGMI,
include/apd.h).  The pair set is fixed as ranks are added (strong scaling).  torch is plumbing
only: device buffers, the stream, and a gloo process group for the control plane (handing the
128-byte communicator id to the ranks, barriers, the max over ranks of the elapsed time); the
compute AND the data-path collective are libapd_hip.so through its C ABI.  If the library's
communicator cannot be made on some rank, every rank falls back to torch.distributed's "nccl"
all_gather_into_tensor and the JSON line says so ("collective").

Prints ONE JSON line on rank 0 (metric: DTW cell-updates/s, whole job), with
  roofline     -- algorithmic bytes (4*D*(n+m)+4 per ordered pair) / measured kernel time vs 8 TB/s
  cpu_baseline -- the CPU oracle (a port of the reference's algorithm) timed on a bounded sample
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# BASELINE.json configs: name -> (n_seq, nominal_len, dim, band_pct)
WORKLOADS = {
    "cfg1": dict(n_seq=64, length=256, dim=13, pct=1.0,
                 desc="64 seq len~256 D=13 full DTW (reference CPU-runnable case)"),
    "cfg2": dict(n_seq=1024, length=512, dim=13, pct=0.0625,
                 desc="1024 seq len~512 D=13 MFCC, Sakoe-Chiba band=32"),
    "cfg3": dict(n_seq=4096, length=1024, dim=13, pct=0.0625,
                 desc="4096 seq len~1024 D=13 MFCC, band=64, all-pairs distance matrix"),
    "cfg4": dict(n_seq=4096, length=1024, dim=8, pct=0.0625, encode_from=13,
                 desc="4096 seq len~1024, 13-dim frames -> 8-dim autoencoder embeddings on-device (neural.rs:55-71) -> DTW, band=64"),
}
WORKLOADS["cfg5s"] = dict(n_seq=512, length=2048, dim=13, pct=0.0625,
                          desc="512 seq len~2048 D=13, band=128 (cfg 5's per-pair shape at 1/32 of its sequence count; not a BASELINE config)")
WORKLOADS["cfg5e"] = dict(n_seq=256, length=2048, dim=13, pct=0.0625, audio=True,
                          desc="cfg 5's end-to-end data path at 1/64 of its sequence count: 256 recordings x 262400 i16 samples -> on-device "
                               "cepstrum (spectrogram.rs:31-80, dft_win 256, step 128, ceps_filter 18 -> 13 bins) -> DTW band=128 (not a BASELINE config)")
WORKLOADS["cfg5"] = dict(n_seq=16384, length=2048, dim=13, pct=0.0625, audio=True,
                         desc="cfg5: 16384 recordings x 262400 i16 samples -> on-device cepstrum (13 bins) -> DTW band=128 "
                              "(add --cluster for the UPGMA leg); ~1 min per step on one GPU")
WORKLOADS["ship"] = dict(n_seq=512, length=525, jitter=375, dim=10, pct=1.0, encode_from=26,
                         desc="the reference's shipped Discovery.toml shape: 512 ragged VAT slices of 150..900 frames, 26-bin cepstra -> "
                              "10-dim autoencoder embeddings on-device -> full DTW (warping_band_percentage = 1.0); not a BASELINE config")
WORKLOADS["wide7"] = dict(n_seq=128, length=4000, dim=13, pct=0.1,
                          desc="128 seq len~4000 D=13, band=400: a binding band wider than one wavefront (dtw_fused_wide; not a BASELINE config)")
WORKLOADS["wide11"] = dict(n_seq=128, length=3000, dim=13, pct=0.4,
                           desc="128 seq len~3000 D=13, band=1200: a binding band of 2400 offsets, 80 % of the sequence (banded strips vs 8-wave band form; not a BASELINE config)")
WORKLOADS["short12"] = dict(n_seq=2048, length=220, jitter=70, dim=10, pct=0.1,
                            desc="2048 short slices of 150..290 frames, D=10, band 10 % (w follows |n-m|): band-form kernels and banded strips mixed (not a BASELINE config)")
WORKLOADS["rag10"] = dict(n_seq=512, length=525, jitter=375, dim=10, pct=0.1,
                          desc="512 ragged slices of 150..900 frames, D=10, band 10 %: w = max(band, |n-m|) + 2 spans every kernel family (not a BASELINE config)")
WORKLOADS["short9"] = dict(n_seq=2048, length=220, jitter=70, dim=10, pct=1.0,
                           desc="2048 short slices of 150..290 frames, D=10, full DTW (the shipped vat_min_len = 150 end of the range; not a BASELINE config)")
WORKLOADS["full8"] = dict(n_seq=96, length=3000, jitter=1000, dim=10, pct=1.0,
                          desc="96 long slices of 2000..4000 frames, D=10, full DTW: several 64*CW-column passes per pair (not a BASELINE config)")
WORKLOADS["full6"] = dict(n_seq=256, length=600, dim=13, pct=1.0,
                          desc="256 seq len~600 D=13, full DTW (the reference's shipped warping_band_percentage = 1.0 on long slices; not a BASELINE config)")
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("APD_WORKLOAD", "cfg3"), choices=sorted(WORKLOADS))
    ap.add_argument("--variant", type=int, default=0, help="kernel variant (0 = auto)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget per leg (0 = skip)")
    ap.add_argument("--verify", type=int, default=64, help="entries re-checked against the oracle after timing")
    ap.add_argument("--distance", default="hybrid", choices=["hybrid", "exact", "strict"], help="local-distance form (apd_set_distance_mode); strict = the reference's arithmetic, bit-identical")
    ap.add_argument("--tau", type=float, default=0.0, help="hybrid recomputation threshold (0 = library default 1/64)")
    ap.add_argument("--cluster", action="store_true", help="also time percentile + UPGMA (rank 0, outside the timed region)")
    ap.add_argument("--backend", default="apd", choices=["apd", "nccl", "gloo"],
                    help="data-path collective: apd = the library's own RCCL communicator (default); nccl = torch.distributed's; "
                         "gloo (staged through host memory) only to rehearse N>1 on a box with fewer GPUs")
    return ap.parse_args()


def host_threads():
    """CPU threads for the oracle legs: the GPU box grants 16 cores per GPU."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    return max(1, min(avail, 16))


def cpu_baseline(frames, offsets, wl, seconds):
    """Times the CPU oracle (oracle/apd_oracle.c, kind = "port") on a bounded random sample of
    ordered pairs with every host core; returns the JSON object."""
    from oracle import binding as oracle
    cores = host_threads()
    rng = np.random.default_rng(1234)
    n = wl["n_seq"]

    def sample(k):
        pi = rng.integers(0, n, k).astype(np.uint32)
        pj = (pi + 1 + rng.integers(0, n - 1, k)).astype(np.uint32) % n
        return pi, pj

    out = {}
    for name, hm in (("dense", False), ("hashmap", True)):
        pi, pj = sample(cores * 4)
        t0 = time.perf_counter()
        _, cells = oracle.align_sample(frames, offsets, pi, pj, wl["pct"], workers=cores, hashmap=hm)
        dt = max(time.perf_counter() - t0, 1e-6)
        k = int(max(cores * 4, min(seconds / dt * len(pi) * 0.8, 2_000_000)))
        pi, pj = sample(k)
        t0 = time.perf_counter()
        _, cells = oracle.align_sample(frames, offsets, pi, pj, wl["pct"], workers=cores, hashmap=hm)
        dt = time.perf_counter() - t0
        out[name] = (cells / dt, k, dt)
    return {
        "value": out["dense"][0], "unit": "cell-updates/s", "cores": cores, "kind": "port",
        "sample": "%d random ordered pairs of the same workload, dense rolling-row oracle, %.1f s"
                  % (out["dense"][1], out["dense"][2]),
        "reference_like_value": out["hashmap"][0],
        "reference_like_sample": "%d pairs with the reference's per-pair hash-map cost structure, %.1f s"
                                 % (out["hashmap"][1], out["hashmap"][2]),
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run (see the docstring)")
        args.gpus = world

    import torch
    import torch.distributed as dist
    from audio_pattern_discovery_amd import _lib, synth
    from audio_pattern_discovery_amd.alignments import align_work

    dev_index = int(os.environ.get("APD_FORCE_DEVICE", local_rank))      # rehearsal: several ranks on one GPU
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    nccl_group = None
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)       # control plane only
        if args.backend == "nccl":
            nccl_group = dist.new_group(backend="nccl", device_id=dev)

    wl = WORKLOADS[args.workload]
    n, dim = wl["n_seq"], wl["dim"]
    src_dim = wl.get("encode_from", dim)
    audio = None
    if wl.get("audio"):
        # synthetic recordings; the frames the CPU legs use come from the oracle's cepstrum of the same audio
        from oracle import binding as _orc
        rng_a = np.random.default_rng(0xA0D10)
        n_samp = 256 + 128 * wl["length"]
        base = [synth.make_audio(n_samp, seed=1000 + k) for k in range(16)]
        audio = [np.clip(base[k % 16].astype(np.int32) + rng_a.integers(-200 * min(k // 16, 40), 200 * min(k // 16, 40) + 1, n_samp, dtype=np.int16),
                         -32768, 32767).astype(np.int16) for k in range(n)]
    frames, offsets = (None, None) if audio is not None else synth.make_sequences(n, wl["length"], src_dim, seed=0xA9D0 + sum(map(ord, args.workload)) % 97, jitter=wl.get("jitter"))
    if audio is not None:
        s_off = np.concatenate([[0], np.cumsum([len(a) for a in audio])]).astype(np.uint64)
        offsets = np.zeros(n + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum([(len(a) - 256 + 127) // 128 for a in audio])
    enc_w = enc_b = None
    if src_dim != dim:                                               # Mat::seeded-scale random encoder (numerics.rs:178-186)
        rng = np.random.default_rng(0xE1C)
        enc_w = ((rng.random((src_dim, dim)) - 0.5) / dim).astype(np.float32)
        enc_b = ((rng.random(dim) - 0.5) / dim).astype(np.float32)
    L = _lib.lib()
    cfg = _lib.AlignConfig(wl["pct"], 1.0, 1.0, 1.0)
    pairs_all, cells_all, bytes_all = align_work(offsets, dim, cfg, 0, 1)
    pairs_r, cells_r, bytes_r = align_work(offsets, dim, cfg, rank, world)

    ctx = _lib.Context(dev_index, stream=torch.cuda.current_stream().cuda_stream)
    ctx.selftest()
    ctx.set_variant(args.variant)
    ctx.set_distance_mode(args.distance, args.tau)
    ctx.set_timing(True)
    total_frames = int(offsets[-1])
    if audio is not None:
        d_audio = torch.from_numpy(np.concatenate(audio)).to(dev)    # inputs resident in HBM
        d_src = torch.empty(total_frames * dim, dtype=torch.float32, device=dev)
        f_off = np.zeros(n + 1, dtype=np.uint64)
        nb = C.c_uint32(0)
    else:
        d_src = torch.from_numpy(frames).to(dev)                     # inputs resident in HBM
    d_frames = d_src if enc_w is None else torch.empty(total_frames * dim, dtype=torch.float32, device=dev)
    f32p = C.POINTER(C.c_float)
    off_c = np.ascontiguousarray(offsets, dtype=np.uint64)
    slab_floats = int(L.apd_slab_floats(n, world))
    d_out = torch.empty(n * n, dtype=torch.float32, device=dev)
    kernel_ms = []
    # the library-owned communicator (one process per GPU: rank 0 makes the id, the control plane hands it out)
    comm, ranks_seen, collective = None, 1, "none (1 rank)"
    if world > 1 and args.backend == "apd":
        from audio_pattern_discovery_amd import sharding
        box = [sharding.Comm.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ok = 1
        try:
            comm = sharding.Comm(ctx, box[0], rank, world)
            ranks_seen = comm.count()
            # one small all-gather now: RCCL sets its channels up on first use, which must not land in a timed step (--warmup 0)
            warm_src = torch.full((256,), float(rank), dtype=torch.float32, device=dev)
            warm_dst = torch.empty(256 * world, dtype=torch.float32, device=dev)
            _lib.check(L.apd_all_gather_async(ctx.handle, comm.handle, C.c_void_p(warm_src.data_ptr()), C.c_void_p(warm_dst.data_ptr()), 256),
                       ctx.handle)
            ctx.synchronize()
            if not torch.equal(warm_dst.view(world, 256)[:, 0].cpu(), torch.arange(world, dtype=torch.float32)):
                raise RuntimeError("all-gather returned the slabs out of rank order")
        except Exception as exc:                                         # noqa: BLE001 -- any failure: agree on the fallback
            sys.stderr.write("[bench] rank %d: apd_comm_create failed (%s); falling back to torch.distributed nccl\n" % (rank, exc))
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            collective = "apd_comm: library-owned RCCL communicator, ncclAllGather inside apd_align_all_sharded_async"
        else:
            if comm is not None:
                comm.close()
                comm = None
            args.backend = "nccl"
            nccl_group = dist.new_group(backend="nccl", device_id=dev)
    if world > 1 and args.backend == "nccl":
        collective = "torch.distributed all_gather_into_tensor (nccl = RCCL)"
        ranks_seen = dist.get_world_size(nccl_group)
    elif world > 1 and args.backend == "gloo":
        collective = "gloo through host memory (rehearsal only)"
    if comm is None:
        d_slab = torch.zeros(slab_floats, dtype=torch.float32, device=dev)
        d_gathered = torch.zeros(slab_floats * world, dtype=torch.float32, device=dev) if world > 1 else d_slab
    # AlignmentWorkers::new once; every step refills the resident copy from the features it just made (same lengths), so the
    # device buffers and the tile plans are made once, not per step
    batch = C.c_void_p()

    def step():
        if audio is not None:                                        # NDSequence::new on the whole corpus, in HBM
            _lib.check(L.apd_cepstrum_batch(ctx.handle, C.c_void_p(d_audio.data_ptr()), s_off.ctypes.data_as(C.POINTER(C.c_uint64)), n,
                                            256, 128, 18, 1, C.c_void_p(d_src.data_ptr()), f_off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                            C.byref(nb)), ctx.handle)
        if enc_w is not None:                                        # NDSequence::encoded on the whole corpus, in HBM
            _lib.check(L.apd_encode(ctx.handle, C.c_void_p(d_src.data_ptr()), total_frames, src_dim, enc_w.ctypes.data_as(f32p),
                                    enc_b.ctypes.data_as(f32p), dim, 1, C.c_void_p(d_frames.data_ptr())), ctx.handle)
        if not batch:
            _lib.check(L.apd_batch_create(ctx.handle, C.c_void_p(d_frames.data_ptr()), off_c.ctypes.data_as(C.POINTER(C.c_uint64)),
                                          n, dim, 1, C.byref(batch)), ctx.handle)
        else:
            _lib.check(L.apd_batch_refill(ctx.handle, batch, C.c_void_p(d_frames.data_ptr()), 1), ctx.handle)
        if comm is not None or world == 1:
            # tiles of this rank + the one ncclAllGather (RCCL over xGMI) + unpack, all inside the library
            _lib.check(L.apd_align_all_sharded_async(ctx.handle, comm.handle if comm is not None else None, batch, C.byref(cfg),
                                                     C.c_void_p(d_out.data_ptr())), ctx.handle)
            return
        _lib.check(L.apd_align_tiles_async(ctx.handle, batch, C.byref(cfg), rank, world, C.c_void_p(d_slab.data_ptr())),
                   ctx.handle)
        if args.backend == "nccl":
            dist.all_gather_into_tensor(d_gathered, d_slab, group=nccl_group)
        else:
            host = torch.empty(slab_floats * world, dtype=torch.float32)
            dist.all_gather_into_tensor(host, d_slab.cpu())          # rehearsal path only
            d_gathered.copy_(host)
        _lib.check(L.apd_unpack_tiles_async(ctx.handle, batch, world, C.c_void_p(d_gathered.data_ptr()),
                                            C.c_void_p(d_out.data_ptr())), ctx.handle)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 0)):
        step()
    if not batch:                                                    # --warmup 0: AlignmentWorkers::new is not part of a step
        _lib.check(L.apd_batch_create(ctx.handle, C.c_void_p(d_frames.data_ptr()), off_c.ctypes.data_as(C.POINTER(C.c_uint64)),
                                      n, dim, 1, C.byref(batch)), ctx.handle)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        ms = ctx.last_kernel_ms()                                    # syncs on the kernel's end event only
        kernel_ms.append(ms)
    fence()
    elapsed = time.perf_counter() - t0
    ctx.synchronize()                                                # raises APD_ERR_INCOMPLETE if any unpack met an unwritten score
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        result = d_out.cpu().numpy().reshape(n, n)
        verify = None
        if args.verify > 0:
            from oracle import binding as oracle
            rng = np.random.default_rng(7)
            pi = rng.integers(0, n, args.verify).astype(np.uint32)
            pj = (pi + 1 + rng.integers(0, n - 1, args.verify)).astype(np.uint32) % n
            if audio is not None:                                    # oracle cepstrum of the sampled recordings only
                need = sorted(set(pi.tolist()) | set(pj.tolist()))
                remap = {s_: k for k, s_ in enumerate(need)}
                feats = [oracle.cepstrum(audio[s_], 256, 128, 18) for s_ in need]
                ref_frames = np.concatenate(feats)
                ref_off = np.concatenate([[0], np.cumsum([len(f) for f in feats])]).astype(np.uint64)
                want, _ = oracle.align_sample(ref_frames, ref_off, np.array([remap[v] for v in pi.tolist()], np.uint32),
                                              np.array([remap[v] for v in pj.tolist()], np.uint32), wl["pct"], workers=host_threads())
            else:
                ref_frames = frames if enc_w is None else oracle.encode(frames, enc_w, enc_b)
                want, _ = oracle.align_sample(ref_frames, offsets, pi, pj, wl["pct"], workers=host_threads())
            verify = float(np.max(np.abs(result[pi, pj] - want) / np.maximum(np.abs(want), 1e-30)))
        traffic = valu_insts = None
        try:
            rec = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).get(args.workload)
            if rec and rec.get("n_gpus") == world:
                traffic = rec["traffic_bytes"]        # measured by rocprofv3 PMC passes of this command (profiles/)
                valu_insts = rec.get("sq_insts_valu")  # wavefront VALU instructions per launch (SQ_INSTS_VALU)
        except (OSError, ValueError):
            pass
        k_ms = float(np.mean(kernel_ms))
        achieved = bytes_r / (k_ms * 1e-3) / 1e9
        line = {
            "metric": "DTW cell-updates/sec (whole node)", "value": cells_all * args.steps / elapsed,
            "unit": "cell-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.workload, wl["desc"]), "n_seq": n, "nominal_len": wl["length"],
                       "dim": dim, "warping_band_percentage": wl["pct"], "ordered_pairs": pairs_all,
                       "cells_per_step": cells_all, "sharding": "pair tiles 16x16, cyclic over %d ranks, 1 all-gather" % world,
                       "collective": collective, "ranks_seen": ranks_seen,
                       "kernel_variant": args.variant, "distance_form": args.distance},
            "wall_clock_matrix_s": elapsed / args.steps,
            "pairs_per_s": pairs_all * args.steps / elapsed,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": None if traffic is None else
                         "profiles/hbm_traffic.json: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE of this command, committed; "
                         "a recorded constant, not re-measured in this run",
                         "note": "rows n / columns m of every pair are not swept (score() reads cell (n-1, m-1), alignments.rs:120): "
                                 "the kernel touches ~0.2 % fewer cells than `value` counts; `value` counts the reference's cells",
                         "kernel": "dtw_fused (rank 0 share: %d ordered pairs)" % pairs_r,
                         "kernel_ms": k_ms, "alg_bytes_per_launch": bytes_r,
                         "kernel_cells_per_s": cells_r / (k_ms * 1e-3)},
            # the kernel is VALU-bound, so the HBM fraction above is reported next to the vector-issue figure:
            # lane-instructions per cell from the SQ_INSTS_VALU counter of this workload, times the live cell rate, over the
            # 7.86e13 lane-instr/s of 1024 SIMDs x 32 lanes x 2.4 GHz (gfx950 SIMDs are 32 lanes wide: 157.3 TFLOP/s f32 / 2)
            "valu": None if not valu_insts else {
                "lane_instr_per_cell": valu_insts * 64.0 / cells_r,
                "lane_instr_per_s": valu_insts * 64.0 / (k_ms * 1e-3),
                "peak_lane_instr_per_s": 1024 * 32 * 2.4e9,
                "frac": valu_insts * 64.0 / (k_ms * 1e-3) / (1024 * 32 * 2.4e9),
                "source": "profiles/hbm_traffic.json (rocprofv3 --pmc SQ_INSTS_VALU of this command, committed; not re-measured in this run)"},
            "max_rel_err_vs_oracle": verify, "parity_ok": (verify is None) or bool(verify <= (1e-3 if audio is not None else 1e-4)),
        }
        if frames is not None and enc_w is None and world == 1:
            # the host-buffer entry of the boundary (AlignmentWorkers::new + align_all on host Vec<f32>s): H2D of the frames,
            # repack, kernel, unpack, D2H of the matrix -- reported beside `value`, never as it
            host_out = np.empty(n * n, dtype=np.float32)
            hb = C.c_void_p()
            t0 = time.perf_counter()
            _lib.check(L.apd_batch_create(ctx.handle, frames.ctypes.data_as(f32p), off_c.ctypes.data_as(C.POINTER(C.c_uint64)), n, dim, 0,
                                          C.byref(hb)), ctx.handle)
            _lib.check(L.apd_align_all(ctx.handle, hb, C.byref(cfg), host_out.ctypes.data_as(f32p)), ctx.handle)
            dt = time.perf_counter() - t0
            L.apd_batch_destroy(hb)
            line["pcie_inclusive"] = {"seconds": dt, "value": cells_all / dt, "unit": "cell-updates/s",
                                      "h2d_bytes": int(frames.nbytes), "d2h_bytes": int(host_out.nbytes),
                                      "bitwise_equal_to_resident_path": bool(np.array_equal(host_out.reshape(n, n), result)),
                                      "note": "apd_batch_create(host frames) + apd_align_all(host out), pageable host memory, one call"}
        if args.cluster:
            ops = (_lib.ClusterOp * n)()
            roots = np.zeros(n, dtype=np.uint32)
            n_ops, n_roots, thr = C.c_uint32(0), C.c_uint32(0), C.c_float(0)
            torch.cuda.synchronize()
            cl_ctx = _lib.Context(dev_index)                         # a context with its own stream: the merge loop is replayed as a hipGraph,
            t0 = time.perf_counter()                                 # which the legacy default stream (torch's current one) cannot capture
            _lib.check(L.apd_clustering(cl_ctx.handle, C.c_void_p(d_out.data_ptr()), 1, n, 0.05, ops, C.byref(n_ops),
                                        roots.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(n_roots), C.byref(thr)), cl_ctx.handle)
            line["clustering"] = {"seconds": time.perf_counter() - t0, "merges": int(n_ops.value), "roots": int(n_roots.value),
                                  "threshold": float(thr.value), "percentile": 0.05,
                                  "note": "apd_clustering on the resident matrix: radix-select threshold + UPGMA (clustering.rs:81-110)"}
        if world == 1 and args.cpu_seconds > 0 and audio is not None:
            line["cpu_baseline"] = None                              # the CPU legs would need the whole corpus' cepstra on the host
        elif world == 1 and args.cpu_seconds > 0:
            from oracle import binding as _o
            line["cpu_baseline"] = cpu_baseline(frames if enc_w is None else _o.encode(frames, enc_w, enc_b), offsets, wl, args.cpu_seconds)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if batch:
        L.apd_batch_destroy(batch)
    if comm is not None:
        comm.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

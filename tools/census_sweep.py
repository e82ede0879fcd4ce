"""How often does the default distance form leave the north star's 1e-4?  Full-matrix census (default mode vs strict mode, which is
bit-identical to the CPU arithmetic) over K independent synthetic corpora of a BASELINE shape:
    python tools/census_sweep.py cfg3|cfg4|cfg2 [K] [first_seed]
Prints one line per corpus and a total.  Run on the GPU box; ~8 s per cfg3 / cfg4 corpus (5 s of it host-side generation)."""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from audio_pattern_discovery_amd import _lib, synth

shape = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 5000
n, length, dim = {"cfg2": (1024, 512, 13), "cfg3": (4096, 1024, 13), "cfg4": (4096, 1024, 8)}[shape]
L = _lib.lib()
ctx = _lib.Context(0)
u64p, f32p = C.POINTER(C.c_uint64), C.POINTER(C.c_float)
cfg = _lib.AlignConfig(0.0625, 1.0, 1.0, 1.0)
tot_entries = tot_over = 0
worst = 0.0
hist = np.zeros(8, dtype=np.int64)           # entries with rel > 1e-7, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3
edges = [1e-7, 1e-6, 3e-6, 1e-5, 3e-5, 1e-4, 3e-4, 1e-3]
t0 = time.time()
for k in range(K):
    seed = seed0 + k
    frames, offsets = synth.make_sequences(n, length, 13, seed=seed)
    d_src = ctx.upload(frames)
    d_frames = d_src
    if shape == "cfg4":                      # 13 -> 8 through the on-device encoder, as bench.py's cfg4 does
        rng = np.random.default_rng(0xE1C + seed)
        w = ((rng.random((13, 8)) - 0.5) / 8).astype(np.float32)
        b = ((rng.random(8) - 0.5) / 8).astype(np.float32)
        total = int(offsets[-1])
        d_frames = ctx.alloc(4 * total * 8)
        _lib.check(L.apd_encode(ctx.handle, d_src.at(), total, 13, w.ctypes.data_as(f32p), b.ctypes.data_as(f32p), 8, 1, d_frames.at()), ctx.handle)
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    batch = C.c_void_p()
    _lib.check(L.apd_batch_create(ctx.handle, d_frames.at(), off.ctypes.data_as(u64p), n, dim, 1, C.byref(batch)), ctx.handle)
    d_out = ctx.alloc(4 * n * n)
    mats = []
    for mode in ("hybrid", "strict"):
        ctx.set_distance_mode(mode)
        _lib.check(L.apd_align_all_device_async(ctx.handle, batch, C.byref(cfg), d_out.at()), ctx.handle)
        ctx.synchronize()
        mats.append(d_out.to_numpy(np.float32).reshape(n, n))
    L.apd_batch_destroy(batch)
    default, strict = mats
    assert np.array_equal(np.isfinite(default), np.isfinite(strict)) and np.array_equal(default == 0, strict == 0)
    m = np.isfinite(strict) & (strict != 0)
    rel = np.abs(default[m] - strict[m]) / np.abs(strict[m])
    over = int((rel > 1e-4).sum())
    for e, edge in enumerate(edges):
        hist[e] += int((rel > edge).sum())
    tot_entries += int(m.sum()); tot_over += over; worst = max(worst, float(rel.max()))
    print("%s seed %d: %d entries, max rel %.3e, > 1e-4: %d   (%.0f s)" % (shape, seed, int(m.sum()), float(rel.max()), over, time.time() - t0), flush=True)
    del d_out, d_src, d_frames
print("TOTAL %s: %d corpora, %d entries, %d beyond 1e-4 (%.2e of the entries), worst %.3e" % (shape, K, tot_entries, tot_over, tot_over / max(tot_entries, 1), worst))
print("entries with relative difference above", ", ".join("%g: %d" % (edge, hist[e]) for e, edge in enumerate(edges)))

#!/usr/bin/env python3
"""GPU box: strict mode against the CPU oracle over EVERY ordered pair of a workload's corpus, bitwise (not a sample).
usage: python tools/strict_full_check.py [workload=cfg3] [rows per chunk=64]
cfg3 = 16.8 M pairs = 2.2e12 oracle cell updates: ~13 minutes on the box's 16 host cores; prints one progress line per chunk."""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from audio_pattern_discovery_amd import _lib  # noqa: E402
from oracle import binding as oracle  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 64
inp = bench.make_inputs(name)
assert inp["audio"] is None and inp["enc_w"] is None, "host-frame workloads only"
n, dim, pct, offsets, frames = inp["n"], inp["dim"], inp["wl"]["pct"], inp["offsets"], inp["frames"]
ctx = _lib.Context(0)
L = _lib.lib()
ctx.set_distance_mode("strict")
batch = C.c_void_p()
d_frames = ctx.upload(frames)
_lib.check(L.apd_batch_create(ctx.handle, d_frames.at(), offsets.ctypes.data_as(C.POINTER(C.c_uint64)), n, dim, 1, C.byref(batch)), ctx.handle)
d_out = ctx.alloc(n * n * 4)
cfg = _lib.AlignConfig(pct, 1.0, 1.0, 1.0)
_lib.check(L.apd_align_all_device_async(ctx.handle, batch, C.byref(cfg), d_out.at()), ctx.handle)
ctx.synchronize()
got = d_out.to_numpy(np.float32).reshape(n, n)
print("%s: strict-mode matrix on the GPU done (%d x %d); oracle on %d host threads ..." % (name, n, n, bench.host_threads()), flush=True)
t0, differing, checked, cells = time.time(), 0, 0, 0
cols = np.arange(n, dtype=np.uint32)
for r0 in range(0, n, chunk):
    rows = np.arange(r0, min(r0 + chunk, n), dtype=np.uint32)
    pi = np.repeat(rows, n)
    pj = np.tile(cols, len(rows))
    keep = pi != pj
    pi, pj = pi[keep], pj[keep]
    want, c = oracle.align_sample(frames, offsets, pi, pj, pct, workers=bench.host_threads())
    cells += c
    bad = got[pi, pj].view(np.uint32) != want.view(np.uint32)
    differing += int(bad.sum())
    checked += len(pi)
    if bad.any():
        k = int(np.flatnonzero(bad)[0])
        print("  DIFFERENCE at pair (%d, %d): gpu %r oracle %r" % (pi[k], pj[k], got[pi[k], pj[k]], want[k]), flush=True)
    print("rows %5d..%5d: %9d pairs checked, %d differing, %.0f s, oracle %.2e cells/s" % (r0, rows[-1], checked, differing, time.time() - t0, cells / (time.time() - t0)), flush=True)
print("RESULT %s: %d of %d ordered pairs differ bitwise from the CPU oracle (diagonal: %s)" % (name, differing, checked, "all zero" if np.all(np.diag(got) == 0) else "NOT zero"))
sys.exit(1 if differing else 0)

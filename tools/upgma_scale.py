"""Scale check of apd_clustering on a resident N x N matrix (run on the GPU box):
python tools/upgma_scale.py N [percentile] [centres] [chain|blob] [dims]"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from audio_pattern_discovery_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
perc = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
rng = np.random.default_rng(0)
dims = int(sys.argv[5]) if len(sys.argv) > 5 else 8                 # 64+: distances concentrate, as DTW costs of long noisy takes do
x = rng.standard_normal((n, dims)).astype(np.float32)
n_centers = int(sys.argv[3]) if len(sys.argv) > 3 else 64          # 16: the family structure of bench.py's cfg 5 audio
centers = rng.standard_normal((n_centers, dims)).astype(np.float32) * 4
if len(sys.argv) > 4 and sys.argv[4] == "chain":                    # member k of a family sits at noise radius ~ k / 16: clusters grow one by one
    k = np.arange(n)
    x = x * (0.02 + 2.0 * (k // n_centers).astype(np.float32)[:, None] / (n / n_centers)) + centers[k % n_centers]
else:
    x = x * 0.5 + centers[rng.integers(0, n_centers, n)]
x = x.astype(np.float32)
sq = (x * x).sum(1)
d = np.sqrt(np.maximum(sq[:, None] + sq[None, :] - 2.0 * (x @ x.T), 0.0)).astype(np.float32)
np.fill_diagonal(d, 0.0)
ctx = _lib.Context(0)
L = _lib.lib()
ops = (_lib.ClusterOp * n)()
roots = np.zeros(n, dtype=np.uint32)
n_ops, n_roots, thr = C.c_uint32(0), C.c_uint32(0), C.c_float(0)
d_dev = ctx.upload(d)
ctx.synchronize()
t0 = time.perf_counter()
_lib.check(L.apd_clustering(ctx.handle, d_dev.at(), 1, n, perc, ops, C.byref(n_ops),
                            roots.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(n_roots), C.byref(thr)), ctx.handle)
dt = time.perf_counter() - t0
print("n=%d perc=%.2f merges=%d roots=%d threshold=%.4f seconds=%.3f us/merge=%.1f" %
      (n, perc, n_ops.value, n_roots.value, thr.value, dt, dt / max(n_ops.value, 1) * 1e6))

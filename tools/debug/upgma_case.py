"""One clustering case, with progress on stderr (APD_DEBUG_UPGMA=1): python tools/debug/upgma_case.py N KIND PERC [torch]
`torch` as 4th argument imports torch first (its bundled HIP runtime instead of /opt/rocm's)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 4 and sys.argv[4] == "torch":
    import torch  # noqa: F401
import numpy as np
from audio_pattern_discovery_amd import _lib, synth
from audio_pattern_discovery_amd.clustering import AgglomerativeClustering
from oracle import binding as oracle
n, kind, perc = int(sys.argv[1]), sys.argv[2], float(sys.argv[3])
print(_lib.runtime_info(), flush=True)
d = synth.make_distance_matrix(n, kind, seed=n + len(kind))
t0 = time.time()
want_ops, want_roots, want_thr = oracle.clustering(d, n, perc, fast=True)
print("oracle %.1fs, %d ops" % (time.time() - t0, len(want_ops)), flush=True)
ctx = _lib.Context(0)
t0 = time.time()
ops, roots, thr = AgglomerativeClustering.clustering(d, n, perc, ctx, return_threshold=True)
print("gpu %.2fs, %d ops, thr equal %s" % (time.time() - t0, len(ops), thr == want_thr), flush=True)
same = [(o.merge_i, o.merge_j, o.into) for o in ops] == [(o["merge_i"], o["merge_j"], o["into"]) for o in want_ops]
gd = np.array([o.distance for o in ops], np.float32); wd = np.array([o["distance"] for o in want_ops], np.float32)
print("sequence equal", same, "linkage bits equal", bool(len(gd) == len(wd) and np.array_equal(gd.view(np.uint32), wd.view(np.uint32))))

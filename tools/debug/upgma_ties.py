"""GPU box: tie-laden integer matrices through apd_clustering vs the oracle, several n (vector and scalar scan paths)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from audio_pattern_discovery_amd import _lib
from audio_pattern_discovery_amd.clustering import AgglomerativeClustering
from oracle import binding as oracle
ctx = _lib.Context(0)
for n in (20, 21, 22, 23, 24, 40, 41):
    for seed in range(3):
        rng = np.random.default_rng(seed)
        d = rng.integers(1, 6, size=(n, n)).astype(np.float32)
        np.fill_diagonal(d, 0.0)
        want, _, _ = oracle.clustering(d, n, 0.95)
        got, _ = AgglomerativeClustering.clustering(d, n, 0.95, ctx)
        g = [(o.merge_i, o.merge_j, o.into) for o in got]
        w = [(o["merge_i"], o["merge_j"], o["into"]) for o in want]
        k = next((i for i in range(min(len(g), len(w))) if g[i] != w[i]), None)
        print(n, seed, "OK" if g == w else "DIFF at %s: got %s want %s (prev %s)" % (k, g[k] if k is not None else None, w[k] if k is not None else None, w[max(0, (k or 0) - 2):k]))

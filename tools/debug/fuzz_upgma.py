"""Randomised sweep of percentile + UPGMA (apd_clustering) against the literal CPU oracle (run on the GPU box).
usage: python tools/debug/fuzz_upgma.py [n_cases] [seed]"""
import sys, time, numpy as np
sys.path.insert(0, ".")
from audio_pattern_discovery_amd import _lib
from audio_pattern_discovery_amd.clustering import AgglomerativeClustering
from oracle import binding as oracle

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = _lib.Context(0)
fails = 0
t0 = time.time()
for case in range(n_cases):
    n = int(rng.choice([1, 2, 3, 5, 8, 17, 33, 64, 65, 100, 150, 260]))
    kind = int(rng.integers(0, 6))
    if kind == 0:
        d = rng.random((n, n)).astype(np.float32) * 10
    elif kind == 1:
        d = rng.integers(0, 4, (n, n)).astype(np.float32)                   # heavy ties, zeros off the diagonal
    elif kind == 2:
        a = rng.random((n, n)).astype(np.float32); d = (a + a.T)            # symmetric
    elif kind == 3:
        d = rng.random((n, n)).astype(np.float32); d[rng.random((n, n)) < 0.2] = np.inf
    elif kind == 4:
        pts = rng.standard_normal((n, 3)).astype(np.float32)
        d = np.sqrt(((pts[:, None] - pts[None]) ** 2).sum(-1)).astype(np.float32)
    else:
        d = np.abs(rng.standard_normal((n, n))).astype(np.float32) * np.float32(1e-3) + rng.integers(0, 2, (n, n)).astype(np.float32)
    if rng.random() < 0.8:
        np.fill_diagonal(d, 0.0)
    perc = float(rng.choice([0.0, 0.01, 0.05, 0.3, 0.5, 0.9, 0.99]))
    try:
        want_ops, want_roots, want_thr = oracle.clustering(d, n, perc)
    except IndexError:
        continue
    ops, roots, thr = AgglomerativeClustering.clustering(d, n, perc, ctx, return_threshold=True)
    ok = (thr == want_thr or (np.isnan(thr) and np.isnan(want_thr)))
    ok = ok and [(o.merge_i, o.merge_j, o.into, o.operation.name) for o in ops] == [(o["merge_i"], o["merge_j"], o["into"], o["operation"]) for o in want_ops]
    ok = ok and sorted(roots) == want_roots
    if ok:
        for g, w in zip(ops, want_ops):
            if g.distance != w["distance"] and not (np.isnan(g.distance) and np.isnan(w["distance"])):
                ok = False
    if ok:
        ok = AgglomerativeClustering.cluster_sets(ops, roots, n) == oracle.cluster_sets(want_ops, want_roots, n)
    if not ok:
        fails += 1
        print("FAIL case", case, dict(n=n, kind=kind, perc=perc), "thr", thr, want_thr, "n_ops", len(ops), len(want_ops), flush=True)
        for t, (g, w) in enumerate(zip(ops, want_ops)):
            if (g.merge_i, g.merge_j, g.into) != (w["merge_i"], w["merge_j"], w["into"]) or g.distance != w["distance"]:
                print("   first diff at op", t, (g.merge_i, g.merge_j, g.into, g.distance), (w["merge_i"], w["merge_j"], w["into"], w["distance"]))
                break
    if case % 40 == 39:
        print("case", case + 1, "fails", fails, "%.0fs" % (time.time() - t0), flush=True)
print("done: cases", n_cases, "fails", fails)
sys.exit(1 if fails else 0)

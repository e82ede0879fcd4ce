"""Randomised sweep of apd_clustering at sizes where linkage chains are cut into speculative segments (N = 200 .. 1600),
against the cached-linkage CPU oracle (tests/test_oracle.py proves that one equal to the literal loop).  Value ranges are
chosen to hit the exact-sum machinery's corners: subnormal and huge magnitudes (sums that start subnormal, sums that overflow
to +INF), many exact zeros, negative entries, NaN / +INF entries, wide dynamic range inside one chain (binade jumps).
usage (GPU box): python tools/debug/fuzz_upgma_large.py [n_cases] [seed]"""
import sys, time, numpy as np
sys.path.insert(0, ".")
from audio_pattern_discovery_amd import _lib
from audio_pattern_discovery_amd.clustering import AgglomerativeClustering
from oracle import binding as oracle

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ctx = _lib.Context(0)
fails = 0
t0 = time.time()
for case in range(n_cases):
    n = int(rng.choice([200, 333, 512, 700, 1024, 1600]))
    k = max(int(rng.choice([2, 3, 5, 16])), 1)
    pts = rng.standard_normal((k, 6)) * 5
    x = pts[rng.integers(0, k, n)] + rng.standard_normal((n, 6)) * rng.choice([0.01, 0.3, 1.5])
    d = np.sqrt(((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)) * (1.0 + 0.1 * rng.random((n, n)))
    kind = int(rng.integers(0, 8))
    if kind == 0:
        d = d * 1e-41                                                    # subnormal distances: sums crawl out of the subnormal range
    elif kind == 1:
        d = d * 1e36                                                     # sums overflow to +INF midway
    elif kind == 2:
        d = d * np.exp(rng.standard_normal((n, n)) * 6)                  # 10 decades inside one chain: binade jumps
    elif kind == 3:
        d[rng.random((n, n)) < 0.3] = 0.0                                # many exact zeros
    elif kind == 4:
        d = d - 0.3 * d.mean()                                           # negative entries
    elif kind == 5:
        d[rng.random((n, n)) < 0.01] = np.nan
        d[rng.random((n, n)) < 0.01] = np.inf
    elif kind == 6:
        d = np.rint(d * 3)                                               # integers: exact ties and exact half-way roundings
    d = d.astype(np.float32)
    np.fill_diagonal(d, 0.0)
    perc = float(rng.choice([0.05, 0.3, 0.7, 0.95]))
    try:
        want_ops, want_roots, want_thr = oracle.clustering(d, n, perc, fast=True)
    except IndexError:
        continue
    ops, roots, thr = AgglomerativeClustering.clustering(d, n, perc, ctx, return_threshold=True)
    ok = (thr == want_thr or (np.isnan(thr) and np.isnan(want_thr))) and len(ops) == len(want_ops) and sorted(roots) == want_roots
    if ok:
        g = np.array([(o.merge_i, o.merge_j, o.into) for o in ops], np.int64).reshape(-1, 3)
        w = np.array([(o["merge_i"], o["merge_j"], o["into"]) for o in want_ops], np.int64).reshape(-1, 3)
        gd = np.array([o.distance for o in ops], np.float32).view(np.uint32)
        wd = np.array([o["distance"] for o in want_ops], np.float32).view(np.uint32)
        nan = np.isnan(gd.view(np.float32)) & np.isnan(wd.view(np.float32))
        ok = np.array_equal(g, w) and np.array_equal(gd[~nan], wd[~nan])
    if not ok:
        fails += 1
        print("FAIL case", case, dict(n=n, kind=kind, perc=perc, k=k), "thr", thr, want_thr, "n_ops", len(ops), len(want_ops), flush=True)
        for t, (a, b) in enumerate(zip(ops, want_ops)):
            if (a.merge_i, a.merge_j, a.into) != (b["merge_i"], b["merge_j"], b["into"]) or np.float32(a.distance).view(np.uint32) != np.float32(b["distance"]).view(np.uint32):
                print("   first diff at op", t, (a.merge_i, a.merge_j, a.into, a.distance), (b["merge_i"], b["merge_j"], b["into"], b["distance"]))
                break
    if case % 10 == 9:
        print("case", case + 1, "fails", fails, "%.0fs" % (time.time() - t0), flush=True)
print("done: cases", n_cases, "fails", fails)
sys.exit(1 if fails else 0)

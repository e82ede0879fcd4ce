"""Edge-case probes of the C ABI through the Python mirror (run on the GPU box)."""
import sys, numpy as np
sys.path.insert(0, ".")
from audio_pattern_discovery_amd import _lib, synth
from audio_pattern_discovery_amd.alignments import AlignmentWorkers, NDSequence
from audio_pattern_discovery_amd.discovery import Discovery
from audio_pattern_discovery_amd.clustering import AgglomerativeClustering
from oracle import binding as oracle

ctx = _lib.Context(0)
rng = np.random.default_rng(0)

def run(arrs, pct=1.0, pens=(1.0, 1.0, 1.0)):
    n = len(arrs)
    got = AlignmentWorkers.new([NDSequence(a) for a in arrs], ctx).align_all(
        Discovery(warping_band_percentage=pct, insertion_penalty=pens[0], deletion_penalty=pens[1], match_penalty=pens[2])).reshape(n, n)
    frames = np.concatenate(arrs) if n else np.zeros((0, 1), np.float32)
    offsets = np.concatenate([[0], np.cumsum([len(a) for a in arrs])]).astype(np.uint64)
    want = oracle.align_all(frames, offsets, pct, *pens, workers=4) if n else np.zeros((0, 0), np.float32)
    return got, want

def same(got, want, tol=1e-4):
    fin = np.isfinite(want)
    if not np.array_equal(fin, np.isfinite(got)): return False
    nz = fin & (want != 0)
    return bool(np.all(got[fin & (want == 0)] == 0)) and (not nz.any() or float((np.abs(got[nz] - want[nz]) / np.abs(want[nz])).max()) <= tol)

print("n_seq=1:", run([rng.standard_normal((40, 13)).astype(np.float32)])[0].tolist())
g, w = run([rng.standard_normal((n, 13)).astype(np.float32) for n in [1, 1, 2, 2, 3, 1, 2, 3, 3, 1] * 30]); print("300 tiny sequences:", same(g, w))
a = rng.standard_normal((200, 13)).astype(np.float32)
g, w = run([a.copy() for _ in range(20)]); print("20 identical sequences all zero:", bool(np.all(g == 0)), same(g, w))
g, w = run([rng.standard_normal((n, 300)).astype(np.float32) for n in (30, 41, 52, 17)], pct=0.3); print("dim 300 (generic):", same(g, w))
g, w = run([rng.standard_normal((n, 13)).astype(np.float32) for n in (60, 70, 80)], pens=(0.0, 1.0, 1.0)); print("zero penalty (generic):", same(g, w))
g, w = run([rng.standard_normal((n, 13)).astype(np.float32) for n in (60, 70, 80)], pens=(-0.5, 1.0, 1.0)); print("negative penalty (generic):", same(g, w))
bad = [rng.standard_normal((n, 13)).astype(np.float32) for n in (60, 70, 300, 90)]
bad[1][10, 3] = np.nan; bad[2][5, 0] = np.inf
g, _ = run(bad); print("NaN / INF features: returned", g.shape, "finite entries", int(np.isfinite(g).sum()))
# clustering edge: all-equal matrix, 2 points, all-INF off-diagonal
for name, d in (("all equal", np.ones((12, 12), np.float32) - np.eye(12, dtype=np.float32)),
                ("all INF", np.where(np.eye(9) > 0, 0, np.inf).astype(np.float32)),
                ("2 points", np.array([[0, 3], [4, 0]], np.float32))):
    n = d.shape[0]
    ops, roots, thr = AgglomerativeClustering.clustering(d, n, 0.5, ctx, return_threshold=True)
    wops, wroots, wthr = oracle.clustering(d, n, 0.5)
    print("clustering", name, [(o.merge_i, o.merge_j, o.into) for o in ops] == [(o["merge_i"], o["merge_j"], o["into"]) for o in wops], sorted(roots) == wroots)
print("done")

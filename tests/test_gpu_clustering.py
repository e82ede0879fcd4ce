"""GPU parity tests of the percentile threshold and the UPGMA clustering (reference src/numerics.rs:125-133,
src/clustering.rs:81-210) through the C ABI, against the literal CPU oracle.

Tolerance: the merge SEQUENCE (merge_i, merge_j, into, Merge kind) must be identical; linkage distances within
1e-5 relative (the device keeps running cluster sums, the reference re-adds raw distances: f32 association order
differs).  Exact ties resolve to the lowest (id_i, id_j), as the oracle does."""
import glob
import os

import numpy as np
import pytest

from audio_pattern_discovery_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx(apd):
    c = apd.Context(0)
    yield c
    c.close()


def check_ops(got, want, n):
    assert [(o.merge_i, o.merge_j, o.into, o.operation.name) for o in got] == \
           [(o["merge_i"], o["merge_j"], o["into"], o["operation"]) for o in want]
    for g, w in zip(got, want):
        if np.isfinite(w["distance"]):
            assert abs(g.distance - w["distance"]) <= 1e-5 * max(abs(w["distance"]), 1e-30)
        else:
            assert g.distance == w["distance"]


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "*.npz"))))
def test_golden_clustering(ctx, oracle, path):
    from audio_pattern_discovery_amd.clustering import AgglomerativeClustering
    g = np.load(path)
    if "dist" not in g:
        pytest.skip("companion fixture")
    n = len(g["offsets"]) - 1
    perc = float(g["params"][4])
    ops, roots, thr = AgglomerativeClustering.clustering(g["dist"], n, perc, ctx, return_threshold=True)
    assert thr == float(g["threshold"])
    assert [[o.merge_i, o.merge_j, o.into] for o in ops] == g["op_ij"].tolist()
    assert [int(o.operation) for o in ops] == g["op_kind"].tolist()
    assert sorted(roots) == g["roots"].tolist()
    sets = AgglomerativeClustering.cluster_sets(ops, roots, n)
    sizes = g["set_sizes"].tolist()
    assert [len(s) for s in sets] == sizes and [m for s in sets for m in s] == g["set_members"].tolist()


@pytest.mark.parametrize("n,perc,seed", [(9, 0.05, 1), (24, 0.3, 2), (40, 0.9, 3), (64, 0.05, 4), (33, 0.999, 5), (2, 0.5, 6), (1, 0.0, 7)])
def test_random_matrices_match_literal_oracle(ctx, oracle, n, perc, seed):
    from audio_pattern_discovery_amd.clustering import AgglomerativeClustering
    frames, offsets = synth.make_sequences(n, 20, 4, seed=seed, jitter=6)
    d = oracle.align_all(frames, offsets, 1.0, workers=8)
    want_ops, want_roots, want_thr = oracle.clustering(d, n, perc)
    ops, roots, thr = AgglomerativeClustering.clustering(d, n, perc, ctx, return_threshold=True)
    assert thr == want_thr
    check_ops(ops, want_ops, n)
    assert sorted(roots) == want_roots
    assert AgglomerativeClustering.cluster_sets(ops, roots, n) == oracle.cluster_sets(want_ops, want_roots, n)


def test_directed_matrix_and_ties(ctx, oracle):
    """Asymmetric distances (the band is asymmetric) and exact ties: lowest (id_i, id_j) wins, both orders count."""
    from audio_pattern_discovery_amd.clustering import AgglomerativeClustering
    rng = np.random.default_rng(0)
    d = rng.integers(1, 6, size=(20, 20)).astype(np.float32)            # many exact ties, d[i][j] != d[j][i]
    np.fill_diagonal(d, 0.0)
    for perc in (0.2, 0.6, 0.95):
        want_ops, want_roots, want_thr = oracle.clustering(d, 20, perc)
        ops, roots, thr = AgglomerativeClustering.clustering(d, 20, perc, ctx, return_threshold=True)
        assert thr == want_thr
        check_ops(ops, want_ops, 20)
        assert sorted(roots) == want_roots


def test_infinite_distances_and_degenerate_merge(ctx, oracle):
    """All linkages +INF: the reference merges its initial (0, 0) (clustering.rs:179) once and stops."""
    from audio_pattern_discovery_amd.clustering import AgglomerativeClustering
    d = np.full((5, 5), np.inf, dtype=np.float32)
    np.fill_diagonal(d, 0.0)
    want_ops, want_roots, want_thr = oracle.clustering(d, 5, 0.5)
    ops, roots, thr = AgglomerativeClustering.clustering(d, 5, 0.5, ctx, return_threshold=True)
    assert thr == want_thr == np.inf
    check_ops(ops, want_ops, 5)
    assert sorted(roots) == want_roots
    # a finite block plus INF rows
    d[:3, :3] = np.array([[0, 1, 4], [2, 0, 3], [5, 6, 0]], np.float32)
    want_ops, want_roots, _ = oracle.clustering(d, 5, 0.9)
    ops, roots = AgglomerativeClustering.clustering(d, 5, 0.9, ctx)
    check_ops(ops, want_ops, 5)
    assert sorted(roots) == want_roots


def test_percentile_select_matches_sort(ctx, oracle, apd):
    from audio_pattern_discovery_amd.clustering import percentile
    rng = np.random.default_rng(1)
    x = rng.standard_normal(100003).astype(np.float32) * 100
    x[::97] = np.nan
    x[5::1013] = np.inf
    x[7::1013] = -np.inf
    x[11::501] = 0.0
    x[12::501] = -0.0
    for perc in (0.0, 0.05, 0.5, 0.73, 0.98):
        assert percentile(x, perc, ctx) == oracle.percentile(x, perc)
    with pytest.raises(apd.ApdError) as e:
        percentile(x, 0.999, ctx)                                   # index beyond the non-NaN count: the reference panics
    assert e.value.status == apd.APD_ERR_INDEX
    with pytest.raises(apd.ApdError):
        percentile(np.arange(4, dtype=np.float32), 1.0, ctx)
    # the index is computed in f32 from the UNFILTERED length (numerics.rs:126)
    y = np.arange(4096 * 4096 // 64, dtype=np.float32)
    assert percentile(y, 0.05, ctx) == oracle.percentile(y, 0.05)


def test_end_to_end_align_then_cluster(ctx, oracle):
    """main.rs:187-203: align_all -> clustering -> cluster_sets, GPU end to end, vs the oracle end to end."""
    from audio_pattern_discovery_amd.alignments import AlignmentWorkers, NDSequence
    from audio_pattern_discovery_amd.clustering import AgglomerativeClustering
    from audio_pattern_discovery_amd.discovery import Discovery
    frames, offsets = synth.make_sequences(48, 64, 13, seed=21, copies=0.5)
    cfg = Discovery(warping_band_percentage=0.0625, clustering_percentile=0.05)
    w = AlignmentWorkers.new([NDSequence(s) for s in synth.split(frames, offsets)], ctx)
    w.align_all(cfg)
    ops, roots = AgglomerativeClustering.clustering(w.result, 48, cfg.clustering_percentile, ctx)
    d = oracle.align_all(frames, offsets, 0.0625, workers=8)
    want_ops, want_roots, _ = oracle.clustering(d, 48, 0.05)
    check_ops(ops, want_ops, 48)
    assert sorted(roots) == want_roots


@pytest.mark.parametrize("n,kind,perc", [(300, "points", 0.05), (300, "ties", 0.6), (700, "nan", 0.5), (1100, "points", 0.05),
                                         (1100, "inf", 0.9), (1500, "ties", 0.4), (2048, "points", 0.05), (4096, "points", 0.05),
                                         (4500, "uniform", 0.02), (4500, "ties", 0.5), (1500, "big", 0.9), (3000, "big", 0.3),
                                         (1200, "neg", 0.7), (2400, "inf", 0.95), (2000, "ties", 0.97), (1800, "nan", 0.9)])
def test_large_matrices_match_fast_oracle(ctx, oracle, n, kind, perc):
    """Device UPGMA at sizes the literal O(n^4) oracle cannot reach (row-minimum loops beyond one workgroup's width, the
    arg-min over more than 1024 rows, grid-stride rows beyond 4096, exact-order chains of 10^5..10^6 terms that are cut into
    speculatively evaluated segments, with binade changes, +INF, NaN and negative entries inside them), against the cached-linkage
    CPU oracle that tests/test_oracle.py proves equal to the literal one.  Merge sequence (merge_i, merge_j, into, kind),
    roots and threshold identical; linkages bit for bit (the tolerance asked is 1e-5)."""
    from audio_pattern_discovery_amd.clustering import AgglomerativeClustering
    d = synth.make_distance_matrix(n, kind, seed=n + len(kind))
    want_ops, want_roots, want_thr = oracle.clustering(d, n, perc, fast=True)
    ops, roots, thr = AgglomerativeClustering.clustering(d, n, perc, ctx, return_threshold=True)
    assert thr == want_thr
    assert len(ops) == len(want_ops)
    got = np.array([(o.merge_i, o.merge_j, o.into, int(o.operation)) for o in ops], dtype=np.int64).reshape(-1, 4)
    names = {"Sequence2Sequence": 0, "Sequence2Cluster": 1, "Cluster2Sequence": 2, "Cluster2Cluster": 3}
    want = np.array([(o["merge_i"], o["merge_j"], o["into"], names[o["operation"]]) for o in want_ops], dtype=np.int64).reshape(-1, 4)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, "first differing merge %d: got %s want %s" % (bad[0], got[bad[0]], want[bad[0]])
    gd = np.array([o.distance for o in ops], dtype=np.float32)
    wd = np.array([o["distance"] for o in want_ops], dtype=np.float32)
    assert np.array_equal(gd.view(np.uint32), wd.view(np.uint32)), "linkage bits differ"
    assert sorted(roots) == want_roots
    if kind == "points":
        assert len(ops) > n // 2                                     # the run really went deep into the dendrogram


@pytest.mark.parametrize("kind,perc", [("points", 0.05), ("ties", 0.3)])
def test_cfg5_size_matrix_matches_fast_oracle(ctx, oracle, kind, perc):
    """N = 16384 (BASELINE cfg 5's matrix: 1 GiB, 2.6 GB of UPGMA state on the device): the same comparison as above at the size
    the survey's row f1 names.  The O(n^3) oracle stays at tens of seconds because these dendrograms are balanced (many small
    clusters: a merge re-sums |Ci| x n terms)."""
    import os
    from audio_pattern_discovery_amd.clustering import AgglomerativeClustering
    n = 16384
    try:
        avail = os.sysconf("SC_AVPHYS_PAGES") * os.sysconf("SC_PAGE_SIZE")
    except (ValueError, OSError):
        avail = 0
    if avail < 6 * (1 << 30):
        pytest.skip("needs ~6 GB of free host memory for the matrix and the oracle's copies")
    d = synth.make_distance_matrix(n, kind, seed=n + len(kind))
    want_ops, want_roots, want_thr = oracle.clustering(d, n, perc, fast=True)
    ops, roots, thr = AgglomerativeClustering.clustering(d, n, perc, ctx, return_threshold=True)
    assert thr == want_thr
    assert len(ops) == len(want_ops) and len(ops) > 1000
    got = np.array([(o.merge_i, o.merge_j, o.into, int(o.operation)) for o in ops], dtype=np.int64).reshape(-1, 4)
    names = {"Sequence2Sequence": 0, "Sequence2Cluster": 1, "Cluster2Sequence": 2, "Cluster2Cluster": 3}
    want = np.array([(o["merge_i"], o["merge_j"], o["into"], names[o["operation"]]) for o in want_ops], dtype=np.int64).reshape(-1, 4)
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, "first differing merge %d: got %s want %s" % (bad[0], got[bad[0]], want[bad[0]])
    gd = np.array([o.distance for o in ops], dtype=np.float32)
    wd = np.array([o["distance"] for o in want_ops], dtype=np.float32)
    assert np.array_equal(gd.view(np.uint32), wd.view(np.uint32)), "linkage bits differ"
    assert sorted(roots) == want_roots


@pytest.mark.parametrize("kind", ["subnormal", "overflow", "decades", "zeros", "negative", "nan_inf", "integers"])
def test_exact_sums_survive_extreme_value_ranges(ctx, oracle, kind):
    """The parallel evaluation of linkage()'s sequential f32 sum (integer maps per binade, speculative segments) at its corners:
    sums that start subnormal, sums that overflow to +INF midway, ten decades inside one chain (binade jumps), many exact zeros,
    negative terms, NaN / +INF terms, integers (exact ties and exact half-way roundings) -- N = 700, three tight clusters so that
    chains run to 10^4..10^5 terms; everything bit for bit against the CPU oracle (tools/debug/fuzz_upgma_large.py is the
    open-ended version)."""
    from audio_pattern_discovery_amd.clustering import AgglomerativeClustering
    n = 700
    rng = np.random.default_rng(len(kind) * 7 + 1)
    pts = rng.standard_normal((3, 6)) * 5
    x = pts[rng.integers(0, 3, n)] + rng.standard_normal((n, 6)) * 0.3
    d = np.sqrt(((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)) * (1.0 + 0.1 * rng.random((n, n)))
    if kind == "subnormal":
        d = d * 1e-41
    elif kind == "overflow":
        d = d * 1e36
    elif kind == "decades":
        d = d * np.exp(rng.standard_normal((n, n)) * 6)
    elif kind == "zeros":
        d[rng.random((n, n)) < 0.3] = 0.0
    elif kind == "negative":
        d = d - 0.3 * d.mean()
    elif kind == "nan_inf":
        d[rng.random((n, n)) < 0.01] = np.nan
        d[rng.random((n, n)) < 0.01] = np.inf
    else:
        d = np.rint(d * 3)
    d = d.astype(np.float32)
    np.fill_diagonal(d, 0.0)
    want_ops, want_roots, want_thr = oracle.clustering(d, n, 0.9, fast=True)
    ops, roots, thr = AgglomerativeClustering.clustering(d, n, 0.9, ctx, return_threshold=True)
    assert thr == want_thr or (np.isnan(thr) and np.isnan(want_thr))
    assert [(o.merge_i, o.merge_j, o.into) for o in ops] == [(o["merge_i"], o["merge_j"], o["into"]) for o in want_ops]
    gd = np.array([o.distance for o in ops], np.float32)
    wd = np.array([o["distance"] for o in want_ops], np.float32)
    both_nan = np.isnan(gd) & np.isnan(wd)
    assert np.array_equal(gd[~both_nan].view(np.uint32), wd[~both_nan].view(np.uint32)), "linkage bits differ"
    assert sorted(roots) == want_roots
    assert len(ops) > 100



@pytest.mark.parametrize("n,kind,perc", [(64, "ties", 0.6), (300, "points", 0.05), (700, "nan", 0.5), (1100, "inf", 0.9), (1500, "big", 0.9),
                                         (2048, "points", 0.05), (4096, "points", 0.05), (3000, "big", 0.3)])
def test_batches_with_and_without_segment_launches_agree_bitwise(ctx, n, kind, perc, monkeypatch):
    """The merge loop replays one of two captured batches: [select, chain, segment] x 64, or -- while the batch before made no
    segment -- [select, chain] x 64, whose chain launch walks a long chain whole if one turns up after all (exact, slow, and the host
    returns to three launches).  The default policy is held to the oracle's bits by the other tests of this file; here the same
    matrices run with the policy pinned to "always three" (APD_UPGMA_TWO_LAUNCH=0) and "always two" (=2: EVERY long chain, up to 10^5
    terms in the "big" matrices, is walked whole by one wavefront) and with the default: op records (ids, kinds, linkage bits), roots and
    threshold must be identical (clustering.rs:153-187 has one answer)."""
    from audio_pattern_discovery_amd.clustering import AgglomerativeClustering
    d = synth.make_distance_matrix(n, kind, seed=n + len(kind))
    results = []
    for policy in ("0", "2", None):
        if policy is None:
            monkeypatch.delenv("APD_UPGMA_TWO_LAUNCH", raising=False)
        else:
            monkeypatch.setenv("APD_UPGMA_TWO_LAUNCH", policy)
        ops, roots, thr = AgglomerativeClustering.clustering(d, n, perc, ctx, return_threshold=True)
        rec = np.array([(o.merge_i, o.merge_j, o.into, int(o.operation), int(np.float32(o.distance).view(np.uint32))) for o in ops],
                       dtype=np.int64).reshape(-1, 5)
        results.append((rec, sorted(roots), np.float32(thr).view(np.uint32)))
    base = results[0]
    assert len(base[0]) > 0
    for other in results[1:]:
        assert other[0].shape == base[0].shape
        bad = np.nonzero((other[0] != base[0]).any(axis=1))[0]
        assert bad.size == 0, "first differing merge %d: %s vs %s" % (bad[0], other[0][bad[0]], base[0][bad[0]])
        assert other[1] == base[1] and other[2] == base[2]

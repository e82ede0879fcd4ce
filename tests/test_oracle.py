"""CPU tests of the oracle itself: hand-derived known answers, an independent
numpy re-derivation, and the committed golden vectors.  No GPU, no product code.

PARITY UNPINNED: the reference has no tests/fixtures and cannot be built here; these checks
pin the oracle to the semantics read off the cited source lines, not to reference outputs.
"""
import glob
import os

import numpy as np
import pytest

from audio_pattern_discovery_amd import synth
from oracle import np_reference as npr

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def col(v):
    return np.array(v, dtype=np.float32).reshape(-1, 1)


# ---- hand-derived known answers (alignments.rs:116-180 read by hand, D=1, penalties 1)

def test_kat_score_reads_cell_n_minus_1_m_minus_1(oracle):
    # x=[0,1,5], y=[0,9]: D[1][1]=0, D[2][1]=0+|1-0|=1 (insert); score reads (2,1) -> 1/5.
    # A textbook DTW (cell (3,2)) would give 1.0.
    for hm in (False, True):
        assert oracle.dtw_pair(col([0, 1, 5]), col([0, 9]), 10, hashmap=hm) == pytest.approx(0.2, abs=1e-7)
    assert float(npr.dtw_pair(col([0, 1, 5]), col([0, 9]), 10)) == pytest.approx(0.2, abs=1e-7)


def test_kat_length_one_cases(oracle):
    x3, x1 = col([0, 1, 5]), col([2])
    assert oracle.dtw_pair(x1, x3, 10) == np.inf       # cell (0, m-1) absent
    assert oracle.dtw_pair(x3, x1, 10) == np.inf       # cell (n-1, 0) absent
    assert oracle.dtw_pair(x1, col([7]), 10) == 0.0    # cell (0,0) = 0
    assert oracle.dtw_pair(np.zeros((0, 1), np.float32), np.zeros((0, 1), np.float32), 3) == np.inf


def test_kat_identical_sequences_score_zero(oracle):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((17, 13)).astype(np.float32)
    assert oracle.dtw_pair(x, x, 0) == 0.0
    assert oracle.dtw_pair(x, x, 17) == 0.0


def test_kat_tie_goes_to_match_even_if_not_minimal(oracle):
    # alignments.rs:153-159: Dl == I < M must take the MATCH branch.
    # x=[0,0,9], y=[0,0,9]; put the tie at cell (2,2): D[1][1]=0, D[1][2]=0, D[2][1]=0, so
    # M=I=Dl=0 everywhere on the plateau -> trivially match; use asymmetric penalties to see it:
    # cell (2,2) distance 0; then (3,3)... score reads (2,2) = 0.  With x=[0,1,.], y=[0,1,.]:
    x, y = col([0, 3, 5]), col([1, 1, 5])
    # D[1][1]=1 (M: 0+1); D[1][2]: Dl=D[1][1]=1, M=D[0][1]=inf, I=D[0][2]=inf -> delete: 1+1=2
    # D[2][1]: I=D[1][1]=1 -> insert: 1+2=3 ; D[2][2]: M=1, I=2, Dl=3 -> match: 1+2=3 ; /6 = 0.5
    assert oracle.dtw_pair(x, y, 10) == pytest.approx(0.5, abs=1e-7)
    # now force Dl == I < M at (2,2): x=[0,2,.], y=[0,2,.] with M raised by a costly (1,1)
    x, y = col([4, 0, 0]), col([0, 0, 0])
    # D[1][1]=4; D[1][2]=Dl 4 + 4 = 8; D[2][1]= I 4 + 0 = 4; D[2][2]: M=4, I=8, Dl=4 -> tie Dl==M -> match: 4
    assert oracle.dtw_pair(x, y, 10) == pytest.approx(4.0 / 6.0, abs=1e-7)


def test_kat_delete_insert_tie_below_match_takes_match(oracle):
    """THE quirk of alignments.rs:153-159: `Dl == I < M` fails both strict tests and lands in the MATCH branch although MATCH is
    the largest predecessor.  Derived by hand (D = 1, unit penalties, full band), x = [0,1,0,0], y = [1,0,1,0]:
        d = |x_i - y_j|:   row 1: 1 0 1 0     row 2: 0 1 0 1     row 3: 1 0 1 0
        D[1][1] = 0+1 = 1   D[1][2] = Dl 1+0 = 1   D[1][3] = Dl 1+1 = 2
        D[2][1] = I 1+0 = 1 D[2][2] = (1,1,1 tie) M 1+1 = 2   D[2][3] = M=1,I=2,Dl=2 -> M 1+0 = 1
        D[3][1] = I 1+1 = 2 D[3][2] = M=1,I=2,Dl=2 -> M 1+0 = 1
        D[3][3]: M = D[2][2] = 2, I = D[2][3] = 1, Dl = D[3][2] = 1  ->  Dl == I < M  ->  MATCH: 2 + 1 = 3
    score = D[3][3] / (4+4) = 0.375.  A DTW that takes the minimum predecessor gives (1 + 1) / 8 = 0.25.
    The swapped pair has the transposed costs and the same tie at (3,3): 0.375 too."""
    x, y = col([0, 1, 0, 0]), col([1, 0, 1, 0])
    for band in (10, 4, 0):                         # band 0 -> w = 2: cell (1,3) is not visited, nothing on the path changes
        for hm in (False, True):
            assert oracle.dtw_pair(x, y, band, hashmap=hm) == 0.375
            assert oracle.dtw_pair(y, x, band, hashmap=hm) == 0.375
        assert float(npr.dtw_pair(x, y, band)) == 0.375
        assert float(npr.dtw_pair(y, x, band)) == 0.375
    # the same values in component 0 of 13-dimensional frames (what the band kernels are instantiated for)
    x13, y13 = np.zeros((4, 13), np.float32), np.zeros((4, 13), np.float32)
    x13[:, 0], y13[:, 0] = x[:, 0], y[:, 0]
    assert oracle.dtw_pair(x13, y13, 10) == 0.375


def test_kat_band_is_asymmetric_and_widened(oracle):
    # w = max(band, |n-m|) + 2 ; j in [max(i-w,1), min(i+w, m+1))  -> j-i in [-w, w-1]
    assert oracle.dtw_cells(5, 5, 0) == sum(min(i + 2, 6) - max(i - 2, 1) for i in range(1, 6))
    assert oracle.dtw_cells(256, 256, 256) == 256 * 256
    assert oracle.dtw_cells(512, 512, 32) == 33660          # BASELINE.md table, cfg 2
    assert oracle.dtw_cells(1024, 1024, 64) == 130812       # cfg 3
    assert oracle.dtw_cells(2048, 2048, 128) == 515580      # cfg 5


def test_kat_warping_band_truncates_in_f32(oracle):
    assert oracle.warping_band(0.0625, 512) == 32
    assert oracle.warping_band(0.0625, 1055) == 65            # 65.9375 -> 65
    assert oracle.warping_band(1.0, 256) == 256
    assert oracle.warping_band(-1.0, 10) == 0                 # Rust `as usize` saturates
    assert oracle.warping_band(float("nan"), 10) == 0
    assert oracle.warping_band(0.1, 20) == npr.warping_band(0.1, 20)


def test_kat_percentile_index_is_f32_of_unfiltered_len(oracle):
    x = np.arange(100, dtype=np.float32)[::-1].copy()
    assert oracle.percentile(x, 0.05) == 5.0
    x[3] = np.nan                                           # NaN dropped, index from the full length
    assert oracle.percentile(x, 0.05) == 5.0
    assert oracle.percentile(np.array([np.inf, 1.0, 2.0], np.float32), 0.9) == np.inf
    with pytest.raises(IndexError):
        oracle.percentile(np.arange(4, dtype=np.float32), 1.0)


def test_kat_diagonal_zero_and_ordered_pairs(oracle):
    frames, offsets = synth.make_sequences(5, 12, 3, seed=3)
    d = oracle.align_all(frames, offsets, 0.0625, workers=2)
    assert np.all(np.diag(d) == 0.0)
    assert np.all(d[~np.eye(5, dtype=bool)] > 0)


# ---- independent re-derivation

@pytest.mark.parametrize("pct,pens,integer", [(1.0, (1, 1, 1), False), (0.0625, (1, 1, 1), True),
                                              (0.25, (0.7, 1.3, 0.9), True), (0.1, (0.0, 1.0, 0.5), False)])
def test_oracle_matches_numpy_rederivation_bitwise(oracle, pct, pens, integer):
    frames, offsets = synth.make_sequences(6, 20, 5, seed=int(pct * 100) + integer, integer=integer, jitter=7)
    a = oracle.align_all(frames, offsets, pct, *pens, workers=3)
    b = oracle.align_all(frames, offsets, pct, *pens, workers=5, hashmap=True)
    c = npr.align_all(synth.split(frames, offsets), pct, *pens)
    assert np.array_equal(a, b)
    assert np.array_equal(a, c)


def test_oracle_clustering_matches_numpy_rederivation(oracle):
    frames, offsets = synth.make_sequences(9, 16, 4, seed=5)
    d = oracle.align_all(frames, offsets, 1.0, workers=4)
    for perc in (0.05, 0.3, 0.9):
        ops, roots, thr = oracle.clustering(d, 9, perc)
        ops2, roots2, thr2 = npr.clustering(d, 9, perc)
        assert ops == ops2 and roots == roots2 and thr == thr2
    # ids and the Merge enum (clustering.rs:192-201)
    ops, roots, _ = oracle.clustering(d, 9, 0.9)
    for t, o in enumerate(ops):
        assert o["into"] == 9 + t
        kinds = {(True, True): "Sequence2Sequence", (False, False): "Cluster2Cluster",
                 (False, True): "Cluster2Sequence", (True, False): "Sequence2Cluster"}
        assert o["operation"] == kinds[(o["merge_i"] < 9, o["merge_j"] < 9)]
    sets = oracle.cluster_sets(ops, roots, 9)
    assert sorted(m for s in sets for m in s) == sorted(set(m for s in sets for m in s))


def test_clustering_emits_first_op_at_or_above_threshold(oracle):
    # clustering.rs:104-107: the op whose distance >= threshold is still pushed, then the loop ends
    d = np.array([[0, 1, 5], [1, 0, 5], [5, 5, 0]], dtype=np.float32)
    ops, roots, thr = oracle.clustering(d, 3, 0.5)            # sorted: 0,0,0,1,1,5,5,5,5 -> idx 4 -> 1
    assert thr == 1.0 and len(ops) == 1 and ops[0]["distance"] == 1.0
    assert (ops[0]["merge_i"], ops[0]["merge_j"], ops[0]["into"]) == (0, 1, 3)
    assert roots == [2, 3]
    assert oracle.cluster_sets(ops, roots, 3) == [[0, 1]]     # singleton root 2 omitted (clustering.rs:71)


# ---- golden vectors (oracle regression pins; they also travel to the GPU box)

@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "*.npz"))))
def test_oracle_reproduces_golden(oracle, path):
    g = np.load(path)
    if "dist" not in g:
        assert np.array_equal(oracle.encode(g["x"], g["w"], g["b"]), g["enc"])
        np.testing.assert_allclose(oracle.cepstrum(g["audio"], 256, 128, 18), g["ceps13"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(oracle.cepstrum(g["audio"], 256, 128, 32), g["ceps26"], rtol=0, atol=1e-6)
        return
    pct, ins, dele, mat, perc = [float(v) for v in g["params"]]
    n = len(g["offsets"]) - 1
    d = oracle.align_all(g["frames"], g["offsets"], pct, ins, dele, mat, workers=2)
    assert np.array_equal(d, g["dist"])
    ops, roots, thr = oracle.clustering(d, n, perc)
    assert [[o["merge_i"], o["merge_j"], o["into"]] for o in ops] == g["op_ij"].tolist()
    assert roots == g["roots"].tolist() and thr == float(g["threshold"])


def test_companion_oracles_definitions(oracle):
    import scipy.fft as sf
    rng = np.random.default_rng(1)
    x = rng.standard_normal((5, 13)).astype(np.float32)
    w = ((rng.random((13, 8)) - 0.5) / 8).astype(np.float32)
    b = ((rng.random(8) - 0.5) / 8).astype(np.float32)
    z = 255.0 / (1.0 + np.exp(-(x.astype(np.float64) @ w + b)))
    mu = z.mean(axis=1, keepdims=True)
    sd = np.maximum(z.std(axis=1, keepdims=True), 1.0)
    np.testing.assert_allclose(oracle.encode(x, w, b), (z - mu) / sd, rtol=2e-4, atol=2e-4)
    audio = synth.make_audio(256 + 128 * 3 + 1, seed=2)
    c = oracle.cepstrum(audio, 256, 128, 18)
    assert c.shape == (4, 13)                                 # ceil((n-256)/128) frames, 17-4 bins
    ham = (0.54 + 0.46 * np.cos(2 * np.pi * np.arange(256) / 256))
    tri = np.array([0, 1, 2, 3, 4, 5, 6, 6, 5, 4, 3, 2, 1, 0]) / 14.0
    fr = audio[128:384].astype(np.float64) * ham
    mag = np.abs(sf.fft(fr))[:128]
    conv = np.array([tri @ mag[p - 14:p] for p in range(14, 128, 7)])
    ce = 0.5 * sf.dct(np.log(conv + 1e-6), type=1)[4:]
    np.testing.assert_allclose(c[1], ce - ce.mean(), rtol=0, atol=2e-4)


# ---- the fast clustering oracle equals the literal one (it is what checks the device UPGMA at N in the thousands)

def _random_distance_matrix(rng, n, kind):
    if kind == "points":                                   # clustered points, directed noise (the band makes d[i][j] != d[j][i])
        centres = rng.standard_normal((max(n // 6, 1), 3)) * 4
        pts = centres[rng.integers(0, len(centres), n)] + rng.standard_normal((n, 3)) * rng.choice([0.05, 0.5, 2.0])
        d = np.sqrt(((pts[:, None, :] - pts[None, :, :]) ** 2).sum(-1))
        d = d * (1.0 + 0.05 * rng.random((n, n)))
    elif kind == "ties":                                   # small integers: exact ties everywhere, asymmetric
        d = rng.integers(1, rng.integers(3, 9), size=(n, n)).astype(np.float64)
    elif kind == "inf":                                    # blocks that cannot reach each other (length-1 sequences give +INF)
        d = rng.random((n, n)) * 10
        cut = rng.integers(1, n) if n > 1 else 0
        d[:cut, cut:] = np.inf
        d[cut:, :cut] = np.inf
        if rng.random() < 0.3:
            d[rng.integers(0, n), :] = np.inf
    elif kind == "nan":
        d = rng.random((n, n)) * 10
        d[rng.random((n, n)) < 0.05] = np.nan
    else:                                                  # plain uniform
        d = rng.random((n, n))
    d = d.astype(np.float32)
    np.fill_diagonal(d, 0.0)
    return d


def test_fast_clustering_oracle_equals_literal_oracle(oracle):
    rng = np.random.default_rng(20261004)
    kinds = ["points", "ties", "inf", "nan", "uniform"]
    cases = 0
    for rep in range(220):
        n = int(rng.integers(1, 72)) if rep < 212 else 256
        kind = kinds[rep % len(kinds)] if rep < 212 else kinds[rep % 3]
        d = _random_distance_matrix(rng, n, kind)
        perc = float(rng.choice([0.0, 0.05, 0.3, 0.6, 0.9, 0.97]))
        try:
            want = oracle.clustering(d, n, perc)
        except IndexError:
            with pytest.raises(IndexError):
                oracle.clustering(d, n, perc, fast=True)
            continue
        got = oracle.clustering(d, n, perc, fast=True)
        assert len(got[0]) == len(want[0])
        for g, w in zip(got[0], want[0]):                  # every field, the linkage bit for bit (NaN never appears in an op)
            assert g == w, (rep, n, kind, perc, g, w)
        assert got[1] == want[1] and (got[2] == want[2] or (np.isnan(got[2]) and np.isnan(want[2])))
        cases += 1
    assert cases >= 200

"""GPU parity tests: the HIP alignment path through the C ABI vs the CPU oracle.

Tolerance (BASELINE.json north_star): every finite distance within 1e-4 relative of the oracle,
identical 0.0 diagonal and identical +INF pattern.  The kernels use an fma chain for the frame
distance and v_sqrt_f32 (<= 1 ulp), so results are not bitwise equal to the two-rounding CPU code.
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from audio_pattern_discovery_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 1e-4


@pytest.fixture(scope="module")
def ctx(apd):
    c = apd.Context(0)
    yield c
    c.close()


def assert_parity(got, want, rtol=RTOL):
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    assert got.shape == want.shape
    fin = np.isfinite(want)
    assert np.array_equal(fin, np.isfinite(got)), "INF/NaN pattern differs"
    assert np.array_equal(np.isposinf(want), np.isposinf(got))
    zero = fin & (want == 0)
    assert np.all(got[zero] == 0), "exact zeros (diagonal, identical sequences) must stay 0"
    nz = fin & ~zero
    if nz.any():
        rel = np.abs(got[nz] - want[nz]) / np.abs(want[nz])
        assert rel.max() <= rtol, "max rel err %.3e" % rel.max()


def gpu_align_all(ctx, frames, offsets, dim, pct, ins=1.0, dele=1.0, mat=1.0, variant=0):
    """variant 0: production kernel, hybrid distances; 1: generic kernel; 2: production kernel, difference-form distances."""
    from audio_pattern_discovery_amd.alignments import AlignmentWorkers, NDSequence
    from audio_pattern_discovery_amd.discovery import Discovery
    ctx.set_distance_mode("exact" if variant == 2 else "hybrid")
    variant = 0 if variant == 2 else variant
    ctx.set_variant(variant)
    seqs = [NDSequence(s.reshape(-1, dim)) for s in synth.split(frames, offsets)]
    w = AlignmentWorkers.new(seqs, ctx)
    n = len(seqs)
    out = w.align_all(Discovery(warping_band_percentage=pct, insertion_penalty=ins, deletion_penalty=dele,
                                match_penalty=mat)).reshape(n, n).copy()
    ctx.set_variant(0)
    return out


def test_selftest_cross_lane_primitives(ctx):
    ctx.selftest()


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "*.npz"))))
def test_golden_vectors(ctx, path, variant):
    g = np.load(path)
    if "dist" not in g:
        pytest.skip("companion fixture")
    pct, ins, dele, mat, _ = [float(v) for v in g["params"]]
    dim = g["frames"].shape[1]
    got = gpu_align_all(ctx, g["frames"], g["offsets"], dim, pct, ins, dele, mat, variant)
    assert_parity(got, g["dist"])


@pytest.mark.parametrize("variant", [0, 1, 2])
@pytest.mark.parametrize("dim,pct,n_seq,length,jitter,integer,pens", [
    (13, 0.0625, 40, 96, 3, False, (1, 1, 1)),        # band binds, w = band + 2
    (13, 0.0625, 24, 200, 40, False, (1, 1, 1)),      # |n-m| > band: widening (alignments.rs:173)
    (13, 1.0, 20, 64, 8, False, (1, 1, 1)),           # full DTW (shipped config)
    (13, 0.0625, 30, 80, 5, True, (1, 1, 1)),         # integer features: exact ties in the select
    (13, 0.25, 30, 80, 5, True, (0.5, 1.5, 0.75)),    # ties + non-unit penalties
    (8, 0.0625, 33, 120, 3, False, (1, 1, 1)),        # autoencoder latents (cfg 4)
    (10, 0.1, 17, 70, 9, False, (0.9, 1.1, 1.0)),     # shipped auto_encoder = 10
    (26, 0.0625, 18, 90, 4, False, (1, 1, 1)),        # shipped ceps_filter = 32 -> 26 bins
    (5, 0.2, 19, 50, 10, True, (1, 1, 1)),            # generic dims
    (1, 0.0, 21, 30, 6, True, (1, 1, 1)),             # band 0 -> w = |n-m| + 2
    (3, 1.0, 10, 300, 100, False, (1, 1, 1)),         # wide full band (C = 9 / generic)
])
def test_random_batches_match_oracle(ctx, oracle, variant, dim, pct, n_seq, length, jitter, integer, pens):
    frames, offsets = synth.make_sequences(n_seq, length, dim, seed=dim * 1000 + n_seq, integer=integer, jitter=jitter)
    want = oracle.align_all(frames, offsets, pct, *pens, workers=8)
    got = gpu_align_all(ctx, frames, offsets, dim, pct, *pens, variant=variant)
    assert_parity(got, want)
    assert np.all(np.diag(got) == 0.0)                # alignments.rs:51


@pytest.mark.parametrize("variant", [0, 2])
@pytest.mark.parametrize("dim", [2, 7, 9, 11, 12, 14, 16, 17, 20, 21, 25, 27, 40])
def test_any_frame_dimension(ctx, oracle, variant, dim):
    """Dimensions without their own instantiation are zero-padded to the next one when made resident (bit-neutral for
    the distances); beyond 26 the generic kernel runs.  Banded with ties, and full DTW on ragged lengths."""
    frames, offsets = synth.make_sequences(20, 70, dim, seed=4000 + dim, integer=(dim % 2 == 0), jitter=6)
    want = oracle.align_all(frames, offsets, 0.1, 0.8, 1.2, 1.0, workers=8)
    assert_parity(gpu_align_all(ctx, frames, offsets, dim, 0.1, 0.8, 1.2, 1.0, variant=variant), want)
    frames, offsets = synth.make_sequences(18, 110, dim, seed=5000 + dim, jitter=50)
    want = oracle.align_all(frames, offsets, 1.0, workers=8)
    assert_parity(gpu_align_all(ctx, frames, offsets, dim, 1.0, variant=variant), want)


def test_matrix_is_directed_when_band_binds(ctx, oracle):
    # the band j-i in [-w, w-1] is asymmetric, so d(i,j) != d(j,i) in general: both triangles are computed
    frames, offsets = synth.make_sequences(16, 150, 13, seed=99, jitter=12)
    want = oracle.align_all(frames, offsets, 0.02, workers=8)
    got = gpu_align_all(ctx, frames, offsets, 13, 0.02)
    assert_parity(got, want)
    assert np.array_equal(want != want.T, got != got.T) or np.abs(got - got.T).max() > 0


def test_edge_lengths(ctx, oracle):
    # lengths 1 and 2 (absent cell -> INF, n=m=1 -> 0.0), alignments.rs:116-125
    rng = np.random.default_rng(5)
    lens = [1, 1, 2, 2, 3, 5, 17]
    seqs = [rng.standard_normal((ln, 13)).astype(np.float32) for ln in lens]
    frames = np.concatenate(seqs)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    want = oracle.align_all(frames, offsets, 1.0, workers=2)
    got = gpu_align_all(ctx, frames, offsets, 13, 1.0)
    assert_parity(got, want)
    assert got[0, 1] == 0.0 and np.isinf(got[0, 2]) and np.isinf(got[2, 0])


def test_empty_inputs(ctx, apd):
    from audio_pattern_discovery_amd.alignments import AlignmentWorkers, NDSequence
    from audio_pattern_discovery_amd.discovery import Discovery
    w = AlignmentWorkers.new([], ctx)
    assert w.align_all(Discovery()).size == 0
    # a zero-length sequence underflows usize in the reference (alignments.rs:120): refused, not guessed
    seqs = [NDSequence(np.zeros((0, 13), np.float32)), NDSequence(np.ones((4, 13), np.float32))]
    with pytest.raises(apd.ApdError) as e:
        AlignmentWorkers.new(seqs, ctx).align_all(Discovery())
    assert e.value.status == apd.APD_ERR_EMPTY_SEQUENCE


def test_single_pair_api_matches_oracle(ctx, oracle):
    from audio_pattern_discovery_amd.alignments import Alignment, AlignmentParams
    rng = np.random.default_rng(3)
    for n, m, band in [(40, 47, 5), (64, 64, 64), (3, 2, 10), (1, 1, 0), (1, 5, 3), (90, 30, 4)]:
        x = rng.standard_normal((n, 13)).astype(np.float32)
        y = rng.standard_normal((m, 13)).astype(np.float32)
        a = Alignment(ctx)
        a.construct_alignment(x, y, AlignmentParams(band, 0.8, 1.2, 1.0))
        want = oracle.dtw_pair(x, y, band, 0.8, 1.2, 1.0)
        assert_parity(np.array([a.score()]), np.array([want]))
    # the same shape again and again with new values (a host looping over fixed windows): the context refills its pair batch
    a = Alignment(ctx)
    for k in range(6):
        x = rng.standard_normal((70, 13)).astype(np.float32) * (k + 1)
        y = rng.standard_normal((64, 13)).astype(np.float32)
        for band in (5, 9):
            a.construct_alignment(x, y, AlignmentParams(band, 1.0, 1.0, 1.0))
            assert_parity(np.array([a.score()]), np.array([oracle.dtw_pair(x, y, band)]))
    assert Alignment(ctx).score() == float("inf")       # Alignment::new().score(), alignments.rs:117-118
    # long sequences, unit penalties: an explicit band that never binds (full-matrix kernel, several column passes, either
    # orientation), one that binds beyond a wavefront (wide kernel), one within a wavefront (systolic)
    for n, m, band in [(2600, 1900, 2600), (1900, 2600, 4000), (2000, 2100, 450), (1500, 1450, 120)]:
        x = np.cumsum(rng.standard_normal((n, 10)), axis=0).astype(np.float32) * 0.3
        y = np.cumsum(rng.standard_normal((m, 10)), axis=0).astype(np.float32) * 0.3
        a = Alignment(ctx)
        a.construct_alignment(x, y, AlignmentParams.default(band))
        assert_parity(np.array([a.score()]), np.array([oracle.dtw_pair(x, y, band)]))


def test_kat_through_the_gpu(ctx):
    from audio_pattern_discovery_amd.alignments import Alignment, AlignmentParams
    a = Alignment(ctx)
    a.construct_alignment(np.array([[0], [1], [5]], np.float32), np.array([[0], [9]], np.float32), AlignmentParams.default(3))
    assert a.score() == pytest.approx(0.2, rel=1e-6)   # SURVEY.md §4 KAT: cell (n-1, m-1), not (n, m)


@pytest.mark.parametrize("dim", [1, 8, 13, 26])
@pytest.mark.parametrize("mode,variant", [("hybrid", 0), ("exact", 0), ("strict", 0), ("hybrid", 1)])
def test_kat_delete_insert_tie_below_match_through_the_gpu(ctx, mode, variant, dim):
    """alignments.rs:153-159, hand-derived in tests/test_oracle.py::test_kat_delete_insert_tie_below_match_takes_match:
    x = [0,1,0,0], y = [1,0,1,0] -> cell (3,3) has Dl == I = 1 < M = 2 and takes MATCH: score 3/8 = 0.375 (a minimum-rule DTW: 0.25).
    Integer features: every distance form is exact, so EVERY mode and kernel family must return exactly 0.375 -- through
    Alignment::construct_alignment (apd_align_pair) and through AlignmentWorkers::align_all, with a binding band (band kernels)
    and a full one (strip kernels)."""
    from audio_pattern_discovery_amd.alignments import Alignment, AlignmentParams, AlignmentWorkers, NDSequence
    from audio_pattern_discovery_amd.discovery import Discovery
    x, y = np.zeros((4, dim), np.float32), np.zeros((4, dim), np.float32)
    x[:, 0], y[:, 0] = [0, 1, 0, 0], [1, 0, 1, 0]
    ctx.set_distance_mode(mode)
    ctx.set_variant(variant)
    try:
        for band in (0, 10):
            a = Alignment(ctx)
            a.construct_alignment(x, y, AlignmentParams.default(band))
            assert a.score() == 0.375
            a.construct_alignment(y, x, AlignmentParams.default(band))
            assert a.score() == 0.375
        seqs = [NDSequence(x), NDSequence(y)] * 9                  # 18 sequences: more than one tile, every ordered pair
        for pct in (0.0, 1.0):
            got = AlignmentWorkers.new(seqs, ctx).align_all(Discovery(warping_band_percentage=pct)).reshape(18, 18)
            for i in range(18):
                for j in range(18):
                    assert got[i, j] == (0.0 if (i - j) % 2 == 0 else 0.375), (pct, i, j, got[i, j])
    finally:
        ctx.set_distance_mode("hybrid")
        ctx.set_variant(0)


def test_sharded_tiles_equal_single_launch(ctx, apd, oracle):
    """world=3 slabs computed one after the other on one GPU, concatenated as an all-gather would,
    then unpacked, must equal the single-launch matrix (the multi-GPU data path minus RCCL)."""
    from audio_pattern_discovery_amd.alignments import Batch
    from audio_pattern_discovery_amd.discovery import Discovery
    frames, offsets = synth.make_sequences(70, 60, 13, seed=4)
    cfg = Discovery(warping_band_percentage=0.0625).align_config()
    L = apd.lib()
    batch = Batch(ctx, frames, offsets, 13)
    n, world = 70, 3
    slab = int(L.apd_slab_floats(n, world))
    gathered = ctx.alloc(4 * world * slab)
    gathered.fill(0)
    for r in range(world):
        apd.check(L.apd_align_tiles_async(ctx.handle, batch.handle, C.byref(cfg), r, world, gathered.at(4 * r * slab)), ctx.handle)
    out = ctx.alloc(4 * n * n)
    apd.check(L.apd_unpack_tiles_async(ctx.handle, batch.handle, world, gathered.at(), out.at()), ctx.handle)
    ctx.synchronize()
    assert_parity(out.to_numpy(np.float32).reshape(n, n), oracle.align_all(frames, offsets, 0.0625, workers=8))


def test_full_size_properties_cfg2(ctx, oracle):
    """BASELINE cfg 2 shape (1024 x len~512, band 32) is too big for the oracle in full: check
    size-independent properties + a random sample of entries against the oracle."""
    frames, offsets = synth.make_sequences(1024, 512, 13, seed=2)
    got = gpu_align_all(ctx, frames, offsets, 13, 0.0625)
    assert np.all(np.diag(got) == 0.0)
    off = ~np.eye(1024, dtype=bool)
    assert np.all(np.isfinite(got[off])) and np.all(got[off] > 0)
    rng = np.random.default_rng(0)
    pi = rng.integers(0, 1024, 600).astype(np.uint32)
    pj = rng.integers(0, 1024, 600).astype(np.uint32)
    keep = pi != pj
    want, _ = oracle.align_sample(frames, offsets, pi[keep], pj[keep], 0.0625, workers=8)
    assert_parity(got[pi[keep], pj[keep]], want)
    # identical sequences score exactly 0: duplicate one sequence and re-run a small batch
    seqs = synth.split(frames, offsets)[:8]
    seqs.append(seqs[3].copy())
    f2 = np.concatenate(seqs)
    o2 = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    g2 = gpu_align_all(ctx, f2, o2, 13, 0.0625)
    assert g2[3, 8] == 0.0 and g2[8, 3] == 0.0


def test_full_size_properties_cfg3(ctx, oracle):
    """BASELINE cfg 3 shape (4096 x len~1024, band 64: 16.8 M ordered pairs) through the device entry points:
    size-independent properties + sampled entries against the oracle + a checksum that must not depend on how
    the pair tiles are sharded (world 1 vs world 4 slabs)."""
    from audio_pattern_discovery_amd.alignments import Batch
    from audio_pattern_discovery_amd.discovery import Discovery
    from audio_pattern_discovery_amd import _lib
    n = 4096
    frames, offsets = synth.make_sequences(n, 1024, 13, seed=0xA9D3)
    frames[int(offsets[77]):int(offsets[78])] = 0.0                        # a constant sequence ...
    frames[int(offsets[78]):int(offsets[78]) + 5] = 0.0                     # ... and a neighbour sharing 5 leading frames
    cfg = Discovery(warping_band_percentage=0.0625).align_config()
    L = _lib.lib()
    batch = Batch(ctx, frames, offsets, 13)
    out = ctx.alloc(4 * n * n)
    _lib.check(L.apd_align_all_device_async(ctx.handle, batch.handle, C.byref(cfg), out.at()), ctx.handle)
    ctx.synchronize()
    got = out.to_numpy(np.float32).reshape(n, n)
    assert np.all(np.diag(got) == 0.0)
    off = ~np.eye(n, dtype=bool)
    assert np.all(np.isfinite(got[off])) and np.all(got[off] >= 0)
    rng = np.random.default_rng(1)
    pi = np.concatenate([rng.integers(0, n, 300), [77, 78, 0, n - 1]]).astype(np.uint32)
    pj = np.concatenate([rng.integers(0, n, 300), [78, 77, n - 1, 0]]).astype(np.uint32)
    keep = pi != pj
    want, _ = oracle.align_sample(frames, offsets, pi[keep], pj[keep], 0.0625, workers=8)
    assert_parity(got[pi[keep], pj[keep]], want)
    # sharding invariance: 4 slabs computed separately, unpacked, give the identical matrix (bitwise)
    world = 4
    slab = int(L.apd_slab_floats(n, world))
    gathered = ctx.alloc(4 * world * slab)
    gathered.fill(0)
    for r in range(world):
        _lib.check(L.apd_align_tiles_async(ctx.handle, batch.handle, C.byref(cfg), r, world, gathered.at(4 * r * slab)), ctx.handle)
    out2 = ctx.alloc(4 * n * n)
    _lib.check(L.apd_unpack_tiles_async(ctx.handle, batch.handle, world, gathered.at(), out2.at()), ctx.handle)
    ctx.synchronize()
    assert np.array_equal(got.view(np.uint32), out2.to_numpy(np.uint32).reshape(n, n))


@pytest.mark.parametrize("scale", [0.0, 1e-4, 1e-3, 1e-2, 0.1, 0.5, 2.0])
def test_near_duplicates_hybrid_distance_accuracy(ctx, oracle, scale):
    """The norm-expansion distance loses digits where frames nearly coincide; those cells are recomputed in the
    difference form.  Sequences that are copies of each other up to noise of every magnitude must stay within
    tolerance, and exact copies must score exactly 0."""
    rng = np.random.default_rng(17)
    base = synth.make_sequences(6, 120, 13, seed=5, copies=0.0)
    frames, offsets = base
    seqs = synth.split(frames, offsets)
    noisy = [s + scale * rng.standard_normal(s.shape).astype(np.float32) for s in seqs]
    offs = [s + np.float32(50.0) for s in seqs[:2]]                      # large common offset: cancellation everywhere
    allseq = seqs + noisy + offs + [o + scale * rng.standard_normal(o.shape).astype(np.float32) for o in offs]
    f2 = np.concatenate(allseq).astype(np.float32)
    o2 = np.concatenate([[0], np.cumsum([len(s) for s in allseq])]).astype(np.uint64)
    want = oracle.align_all(f2, o2, 0.0625, workers=8)
    got = gpu_align_all(ctx, f2, o2, 13, 0.0625, variant=0)
    assert_parity(got, want)
    if scale == 0.0:
        assert got[0, 6] == 0.0 and got[6, 0] == 0.0 and got[12, 14] == 0.0


@pytest.mark.parametrize("pct", [0.0625, 1.0])
def test_mixed_lengths_use_several_kernel_geometries(ctx, oracle, pct):
    """Lengths from 3 to ~700 in one batch: tiles are grouped by the band their widest pair needs, so this run goes
    through several systolic geometries AND the generic kernel in one align_all; every entry must still match."""
    rng = np.random.default_rng(23)
    lens = np.concatenate([rng.integers(3, 40, 20), rng.integers(60, 90, 20), rng.integers(250, 300, 12), rng.integers(600, 700, 6)])
    seqs = [np.cumsum(rng.standard_normal((int(n), 13)), axis=0).astype(np.float32) * 0.3 for n in lens]
    frames = np.concatenate(seqs)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    want = oracle.align_all(frames, offsets, pct, workers=8)
    got = gpu_align_all(ctx, frames, offsets, 13, pct)
    assert_parity(got, want)


@pytest.mark.parametrize("variant", [10205, 10207, 10209, 10405, 10407, 10409, 10805, 10807, 10809])
@pytest.mark.parametrize("distance", ["hybrid", "exact"])
def test_wide_kernel_geometries(ctx, oracle, variant, distance):
    """dtw_fused_wide (NW waves per pair, LDS mailboxes at the wave seams) in every instantiated geometry, full DTW and
    a binding band, on lengths that put the result cell, the D[0][0] injection and the band edges in different waves."""
    rng = np.random.default_rng(variant)
    lens = [2, 3, 70, 130, 260, 333, 520, 611] if variant >= 10400 else [2, 3, 70, 130, 200, 260, 290, 310]
    seqs = [np.cumsum(rng.standard_normal((n, 13)), axis=0).astype(np.float32) * 0.4 for n in lens]
    seqs.append(seqs[5].copy())                                          # exact duplicate: 0.0
    frames = np.concatenate(seqs)
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    from audio_pattern_discovery_amd.alignments import AlignmentWorkers, NDSequence
    from audio_pattern_discovery_amd.discovery import Discovery
    for pct in (1.0, 0.3):
        want = oracle.align_all(frames, offsets, pct, workers=8)
        ctx.set_distance_mode(distance)
        ctx.set_variant(variant)
        w = AlignmentWorkers.new([NDSequence(s) for s in seqs], ctx)
        got = w.align_all(Discovery(warping_band_percentage=pct)).reshape(len(seqs), len(seqs)).copy()
        ctx.set_variant(0)
        ctx.set_distance_mode("hybrid")
        assert_parity(got, want)
        dup = len(seqs) - 1
    assert got[5, dup] == 0.0 and got[dup, 5] == 0.0


@pytest.mark.parametrize("variant", [0, 20103, 20105, 20107, 20109, 20111, 20113, 20203, 20207, 20213, 20403, 20405, 20409, 20413])
@pytest.mark.parametrize("distance", ["hybrid", "exact"])
def test_full_matrix_kernel(ctx, oracle, variant, distance):
    """dtw_full_matrix (column strips in passes, ONE DP for both ordered pairs): valid when the band never binds and the penalties
    are equal -- then score(a,b) == score(b,a) in the reference itself, which the oracle confirms here bit for bit."""
    rng = np.random.default_rng(variant + 7)
    # strips of 64 * CW columns: one pass for the short lengths, up to three for the long ones (the boundary column of a
    # pass goes through LDS), lengths on either side of a pass boundary
    wcols = (64 // max((variant - 20000) // 100, 1)) * (variant % 100)   # columns per pass: G * CW
    lens = [2, 3, 50, 64, 65, 130, 190, 191] if variant == 0 else [2, 3, 70, 193, 194, 333, 520, 571, wcols + 1, wcols + 2, 2 * wcols + 1]
    lens = lens + [1, 17, 17, 18, 40, 40, 41, 75, 100]                   # 26 sequences: tiles whose waves hold several different pairs
    seqs = [np.cumsum(rng.standard_normal((n, 13)), axis=0).astype(np.float32) * 0.4 for n in lens]
    seqs.append(seqs[5].copy())
    frames = np.concatenate(seqs)
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    want = oracle.align_all(frames, offsets, 1.0, 0.7, 0.7, 0.7, workers=8)
    assert np.array_equal(want, want.T)                                  # the symmetry the kernel relies on
    from audio_pattern_discovery_amd.alignments import AlignmentWorkers, NDSequence
    from audio_pattern_discovery_amd.discovery import Discovery
    ctx.set_distance_mode(distance)
    ctx.set_variant(variant)
    w = AlignmentWorkers.new([NDSequence(s) for s in seqs], ctx)
    got = w.align_all(Discovery(warping_band_percentage=1.0, insertion_penalty=0.7, deletion_penalty=0.7,
                                match_penalty=0.7)).reshape(len(seqs), len(seqs)).copy()
    ctx.set_variant(0)
    ctx.set_distance_mode("hybrid")
    assert_parity(got, want)
    dup = len(seqs) - 1
    assert got[5, dup] == 0.0 and got[dup, 5] == 0.0


@pytest.mark.parametrize("dim,lo,hi,pct", [(10, 150, 900, 1.0), (13, 500, 2400, 1.0), (13, 300, 1200, 0.5), (26, 200, 700, 1.0)])
def test_multi_wave_kernels_are_deterministic(ctx, oracle, dim, lo, hi, pct):
    """Ragged corpora on the multi-wave kernels (full-matrix NW up to 8; band-form wide kernel for pct 0.5): the waves of
    a workgroup hand DP values across their seams through LDS mailboxes, so besides parity the result must be bitwise
    the same run after run (a mailbox overwritten before a slower wave has read it shows up here)."""
    from audio_pattern_discovery_amd.alignments import AlignmentWorkers, NDSequence
    from audio_pattern_discovery_amd.discovery import Discovery
    n = 40
    frames, offsets = synth.make_sequences(n, (lo + hi) // 2, dim, seed=31 * dim + hi, jitter=(hi - lo) // 2)
    seqs = [NDSequence(s) for s in synth.split(frames, offsets)]
    cfg = Discovery(warping_band_percentage=pct)
    runs = []
    for _ in range(4):
        runs.append(AlignmentWorkers.new(seqs, ctx).align_all(cfg).reshape(n, n).copy())
    for r in runs[1:]:
        assert np.array_equal(runs[0], r)
    assert_parity(runs[0], oracle.align_all(frames, offsets, pct, workers=16))


def test_launches_are_cut_below_the_work_item_limit(ctx, oracle):
    """11700 short sequences on a 64-lanes-per-pair kernel: 268 278 tiles x 16 384 work-items exceed 2^32, beyond which the
    runtime cuts a grid short without an error; launch_align issues runs of tiles instead.  (cfg 5 -- 16384 recordings,
    band 128 -- is the BASELINE configuration that crosses this limit.)"""
    from audio_pattern_discovery_amd import _lib
    n, dim = 11700, 13
    rng = np.random.default_rng(3)
    lens = rng.integers(50, 55, n)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    frames = rng.standard_normal((int(offsets[-1]), dim)).astype(np.float32)
    L = _lib.lib()
    d_frames = ctx.upload(frames)
    batch = C.c_void_p()
    _lib.check(L.apd_batch_create(ctx.handle, d_frames.at(), offsets.ctypes.data_as(C.POINTER(C.c_uint64)), n, dim, 1,
                                  C.byref(batch)), ctx.handle)
    cfg = _lib.AlignConfig(1.0, 1.0, 1.0, 1.0)
    d_out = ctx.alloc(4 * n * n)
    _lib.check(L.apd_align_all_device_async(ctx.handle, batch, C.byref(cfg), d_out.at()), ctx.handle)
    ctx.synchronize()
    full = d_out.to_numpy(np.float32).reshape(n, n)
    order = np.argsort(-lens, kind="stable")                     # tiles follow the length order: sample its head, middle and tail
    pos_i = np.concatenate([rng.integers(0, n, 200), [0, 1, n - 1, n - 2, n // 2]])
    pos_j = np.concatenate([rng.integers(0, n, 200), [n - 1, n - 3, n - 5, 0, n - 1]])
    keep = pos_i != pos_j
    pi, pj = order[pos_i[keep]].astype(np.uint32), order[pos_j[keep]].astype(np.uint32)
    want, _ = oracle.align_sample(frames, offsets, pi, pj, 1.0, workers=8)
    got = full[pi, pj]
    L.apd_batch_destroy(batch)
    assert_parity(got, want)
    assert float(np.abs(np.diag(full)).max()) == 0.0
    assert int((full == 0).sum()) == n                           # nothing left unwritten: only the diagonal is zero


@pytest.mark.parametrize("dim", [13, 26])
@pytest.mark.parametrize("key", [809, 1602, 1603, 1605, 1607, 1609, 3205, 3207, 3209, 6403, 6405, 6407, 6409])
def test_every_systolic_geometry(ctx, oracle, key, dim):
    """Each (lanes per pair G, offsets per lane C) of dtw_fused_systolic, forced through the tuning variant, on a band that
    fills its lanes (2w+1 just below G*C) and on a narrow one (idle upper lanes); unit penalties in both distance forms,
    and unequal penalties (bit-identical to the oracle).  G = 32 moves data across lanes with two DPP writes per move."""
    g, c = key // 100, key % 100
    if c > (9 if dim <= 13 else 7 if dim <= 16 else 5):
        pytest.skip("C = 9 is instantiated for D <= 13, C = 7 for D <= 16")
    from audio_pattern_discovery_amd.alignments import AlignmentWorkers, NDSequence
    from audio_pattern_discovery_amd.discovery import Discovery
    length = 300
    for band in ((g * c - 1) // 2 - 2, 3):
        pct = (band + 0.5) / length
        frames, offsets = synth.make_sequences(20, length, dim, seed=key + dim + band, jitter=0)
        seqs = [NDSequence(s) for s in synth.split(frames, offsets)]
        for pens, mode in (((1.0, 1.0, 1.0), "hybrid"), ((1.0, 1.0, 1.0), "exact"), ((0.6, 1.3, 1.0), "hybrid")):
            want = oracle.align_all(frames, offsets, pct, *pens, workers=8)
            ctx.set_distance_mode(mode)
            ctx.set_variant(key)
            got = AlignmentWorkers.new(seqs, ctx).align_all(Discovery(warping_band_percentage=pct, insertion_penalty=pens[0],
                                                                      deletion_penalty=pens[1], match_penalty=pens[2])).reshape(20, 20)
            ctx.set_variant(0)
            ctx.set_distance_mode("hybrid")
            if pens == (1.0, 1.0, 1.0):
                assert_parity(got, want)
            else:
                assert np.array_equal(got, want)


@pytest.mark.parametrize("variant", [0, 30105, 30109, 30405, 30409])
@pytest.mark.parametrize("distance", ["hybrid", "exact"])
def test_banded_column_strips(ctx, oracle, variant, distance):
    """Column strips with a binding band (dtw_full_matrix<..., BANDED>): two DPs, band edges masked by +INF distances.
    Ragged lengths make w = max(band, |n - m|) + 2 wide for unequal pairs and narrow for equal ones; forced variants put
    every tile on the strips, variant 0 lets the dispatcher mix them with the band-form kernels."""
    from audio_pattern_discovery_amd.alignments import AlignmentWorkers, NDSequence
    from audio_pattern_discovery_amd.discovery import Discovery
    rng = np.random.default_rng(variant + 3)
    lens = [2, 3, 60, 61, 75, 130, 200, 333, 334, 520, 700, 64 * 9 + 2, 90, 90, 91, 150, 410, 55]
    seqs = [np.cumsum(rng.standard_normal((n, 13)), axis=0).astype(np.float32) * 0.4 for n in lens]
    seqs.append(seqs[7].copy())
    frames = np.concatenate(seqs)
    offsets = np.concatenate([[0], np.cumsum([len(s) for s in seqs])]).astype(np.uint64)
    for pct, pen in ((0.05, 1.0), (0.3, 0.7), (0.9, 1.0)):
        want = oracle.align_all(frames, offsets, pct, pen, pen, pen, workers=8)
        ctx.set_distance_mode(distance)
        ctx.set_variant(variant)
        runs = [AlignmentWorkers.new([NDSequence(s) for s in seqs], ctx).align_all(
            Discovery(warping_band_percentage=pct, insertion_penalty=pen, deletion_penalty=pen, match_penalty=pen)
        ).reshape(len(seqs), len(seqs)).copy() for _ in range(2)]
        ctx.set_variant(0)
        ctx.set_distance_mode("hybrid")
        assert np.array_equal(runs[0], runs[1])
        assert_parity(runs[0], want)
        assert runs[0][7, len(seqs) - 1] == 0.0


def test_one_batch_many_configurations(ctx, oracle):
    """The same resident batch aligned under different Discovery settings: the cached tile plans are keyed by everything
    the kernel choice depends on (a plan made for equal penalties puts full-band tiles on the one-DP kernel, which is
    wrong for unequal ones)."""
    from audio_pattern_discovery_amd.alignments import AlignmentWorkers, NDSequence
    from audio_pattern_discovery_amd.discovery import Discovery
    frames, offsets = synth.make_sequences(30, 120, 13, seed=77, jitter=50)
    w = AlignmentWorkers.new([NDSequence(s) for s in synth.split(frames, offsets)], ctx)
    for pct, pens in [(1.0, (1.0, 1.0, 1.0)), (1.0, (0.5, 1.5, 1.0)), (0.1, (1.0, 1.0, 1.0)), (0.1, (2.0, 1.0, 0.5)), (1.0, (0.7, 0.7, 0.7)),
                      (1.0, (1.0, 1.0, 1.0))]:
        want = oracle.align_all(frames, offsets, pct, *pens, workers=8)
        got = w.align_all(Discovery(warping_band_percentage=pct, insertion_penalty=pens[0], deletion_penalty=pens[1],
                                    match_penalty=pens[2])).reshape(30, 30).copy()
        assert_parity(got, want)


def test_very_long_sequences_and_the_length_limit(ctx, oracle):
    """A 33 000-frame recording against short slices under a full band: column strips with a 132 KB boundary column in
    LDS.  Beyond ~38 000 frames no kernel holds such a band (the strips keep one float per row in LDS, the generic kernel
    its band offsets): the call reports an error instead of computing garbage, and the context stays usable.  Then the
    destruction order the garbage collector may choose: context first, its batches afterwards."""
    from audio_pattern_discovery_amd import _lib
    from audio_pattern_discovery_amd.alignments import AlignmentWorkers, NDSequence
    from audio_pattern_discovery_amd.discovery import Discovery
    rng = np.random.default_rng(0)
    arrs = [rng.standard_normal((n, 8)).astype(np.float32) for n in (33000, 120, 70)]
    frames = np.concatenate(arrs)
    offsets = np.concatenate([[0], np.cumsum([len(a) for a in arrs])]).astype(np.uint64)
    got = AlignmentWorkers.new([NDSequence(a) for a in arrs], ctx).align_all(Discovery(warping_band_percentage=1.0)).reshape(3, 3)
    assert_parity(got, oracle.align_all(frames, offsets, 1.0, workers=8))
    too_long = [NDSequence(rng.standard_normal((n, 8)).astype(np.float32)) for n in (41000, 120)]
    w = AlignmentWorkers.new(too_long, ctx)
    with pytest.raises(Exception) as e:
        w.align_all(Discovery(warping_band_percentage=1.0))
    assert "band" in str(e.value).lower() or "wide" in str(e.value).lower()
    ok = AlignmentWorkers.new([NDSequence(rng.standard_normal((n, 8)).astype(np.float32)) for n in (50, 60)], ctx)
    assert np.isfinite(ok.align_all(Discovery(warping_band_percentage=1.0))).all()
    # context destroyed before its batch: apd_destroy releases the batch's device memory, apd_batch_destroy the rest
    c2 = _lib.Context(0)
    w2 = AlignmentWorkers.new([NDSequence(a) for a in arrs[1:]], c2)
    c2.close()
    del w2


# ---- non-finite features, poison, fault injection ---------------------------------------------------------------

def _assert_same_bits_or_nan(got, want):
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    nan = np.isnan(want)
    assert np.array_equal(nan, np.isnan(got)), "NaN pattern differs"
    assert np.array_equal(got[~nan].view(np.uint32), want[~nan].view(np.uint32)), "bits differ"


@pytest.mark.parametrize("pct,pens", [(0.0625, (1.0, 1.0, 1.0)), (1.0, (1.0, 1.0, 1.0)), (0.25, (0.7, 1.3, 0.9))])
def test_nonfinite_features_follow_the_reference_select(ctx, oracle, apd, pct, pens):
    """NaN compares false, so a node whose predecessors or distance are NaN takes the MATCH branch and the NaN propagates
    (alignments.rs:153-159); an infinite feature gives d = +INF or (INF - INF) NaN.  The fast kernels' fminf select would drop
    NaNs: the repack kernel flags such a batch and every pair goes through the literal kernel, whose results equal the
    oracle's bit for bit (NaN payloads aside)."""
    frames, offsets = synth.make_sequences(20, 40, 13, seed=31, jitter=9)
    frames = frames.copy()
    o = offsets.astype(np.int64)
    frames[o[3] + 5, 2] = np.nan                   # one NaN component in the middle of sequence 3
    frames[o[7] + 0, :] = np.nan                   # a whole NaN first frame
    frames[o[9] + 17, 4] = np.inf                  # +INF component
    frames[o[12] + 30, 0] = -np.inf                # -INF component
    frames[o[12] + 31, 0] = np.inf
    frames[o[15 + 1] - 1, 1] = np.nan              # NaN in the LAST frame of sequence 15: never read by score()
    want = oracle.align_all(frames, offsets, pct, *pens, workers=8)
    assert np.isnan(want).any() and np.isfinite(want).any()
    got = gpu_align_all(ctx, frames, offsets, 13, pct, *pens)
    _assert_same_bits_or_nan(got, want)
    # the flag is per batch: a finite batch on the same context goes back to the fast kernels, unflagged
    from audio_pattern_discovery_amd.alignments import Batch
    nf = C.c_int(-1)
    b = Batch(ctx, frames, offsets, 13)
    apd.check(apd.lib().apd_batch_nonfinite(ctx.handle, b.handle, C.byref(nf)), ctx.handle)
    assert nf.value == 1
    clean, off2 = synth.make_sequences(20, 40, 13, seed=31, jitter=9)
    b2 = Batch(ctx, clean, off2, 13)
    apd.check(apd.lib().apd_batch_nonfinite(ctx.handle, b2.handle, C.byref(nf)), ctx.handle)
    assert nf.value == 0


def test_a_shortened_launch_is_reported_not_zero_filled(ctx, oracle, apd):
    """Every slab is poisoned with NaN before the alignment launches and the unpack writes every matrix entry: tiles that a
    launch failed to process (here: fault injection drops the last 2 tiles of every kernel class) surface as NaN in the
    matrix and as APD_ERR_INCOMPLETE -- the reference would leave silent 0.0 rows (alignments.rs:64-66)."""
    from audio_pattern_discovery_amd.alignments import AlignmentWorkers, NDSequence
    from audio_pattern_discovery_amd.discovery import Discovery
    n = 80
    frames, offsets = synth.make_sequences(n, 50, 13, seed=77)
    want = oracle.align_all(frames, offsets, 0.0625, workers=8)
    w = AlignmentWorkers.new([NDSequence(s) for s in synth.split(frames, offsets)], ctx)
    ctx.set_fault_injection(2)
    try:
        with pytest.raises(apd.ApdError) as e:
            w.align_all(Discovery(warping_band_percentage=0.0625))
    finally:
        ctx.set_fault_injection(0)
    assert e.value.status == apd.APD_ERR_INCOMPLETE
    got = w.result.reshape(n, n).copy()
    off = ~np.eye(n, dtype=bool)
    assert np.isnan(got).any() and not np.any(got[off] == 0.0)            # poison, not zeros
    ok = ~np.isnan(got)
    assert_parity(got[ok], want[ok])                                      # what was written is right
    assert np.all(np.diag(got) == 0.0)
    # the asynchronous entry points report through apd_synchronize, once
    from audio_pattern_discovery_amd.alignments import Batch
    b = Batch(ctx, frames, offsets, 13)
    cfg = Discovery(warping_band_percentage=0.0625).align_config()
    out = ctx.alloc(4 * n * n)
    out.fill(0)
    ctx.set_fault_injection(1)
    apd.check(apd.lib().apd_align_all_device_async(ctx.handle, b.handle, C.byref(cfg), out.at()), ctx.handle)
    ctx.set_fault_injection(0)
    with pytest.raises(apd.ApdError) as e2:
        ctx.synchronize()
    assert e2.value.status == apd.APD_ERR_INCOMPLETE
    ctx.synchronize()                                                     # reported once, then clear
    # and a healthy run afterwards is complete and right
    assert_parity(w.align_all(Discovery(warping_band_percentage=0.0625)).reshape(n, n), want)


def test_batch_refill_keeps_plans_and_takes_new_values(ctx, oracle, apd):
    from audio_pattern_discovery_amd.alignments import Batch
    from audio_pattern_discovery_amd.discovery import Discovery
    n = 40
    f1, offsets = synth.make_sequences(n, 70, 13, seed=5)
    f2 = (f1[::-1] * np.float32(1.5)).copy()                              # other values, same lengths
    cfg = Discovery(warping_band_percentage=0.0625).align_config()
    L = apd.lib()
    b = Batch(ctx, f1, offsets, 13)
    out = ctx.alloc(4 * n * n)
    apd.check(L.apd_align_all_device_async(ctx.handle, b.handle, C.byref(cfg), out.at()), ctx.handle)
    ctx.synchronize()
    assert_parity(out.to_numpy(np.float32).reshape(n, n), oracle.align_all(f1, offsets, 0.0625, workers=8))
    d_f2 = ctx.upload(f2)
    apd.check(L.apd_batch_refill(ctx.handle, b.handle, d_f2.at(), 1), ctx.handle)
    apd.check(L.apd_align_all_device_async(ctx.handle, b.handle, C.byref(cfg), out.at()), ctx.handle)
    ctx.synchronize()
    assert_parity(out.to_numpy(np.float32).reshape(n, n), oracle.align_all(f2, offsets, 0.0625, workers=8))
    f3 = f1.copy()
    f3[11, 3] = np.nan                                                    # a refill re-evaluates the non-finite flag
    apd.check(L.apd_batch_refill(ctx.handle, b.handle, f3.ctypes.data_as(C.POINTER(C.c_float)), 0), ctx.handle)
    apd.check(L.apd_align_all_device_async(ctx.handle, b.handle, C.byref(cfg), out.at()), ctx.handle)
    ctx.synchronize()
    _assert_same_bits_or_nan(out.to_numpy(np.float32).reshape(n, n), oracle.align_all(f3, offsets, 0.0625, workers=8))


# ---- the multi-GPU entry points on the one GPU a test box has ---------------------------------------------------------

def test_align_all_multi_one_device_equals_align_all(ctx, oracle, apd):
    """apd_align_all_multi with n_devices = 1: ncclCommInitAll, the in-place all-gather of one slab, unpack -- the same bits as
    apd_align_all, and RCCL reports one rank."""
    from audio_pattern_discovery_amd import sharding
    from audio_pattern_discovery_amd.discovery import Discovery
    frames, offsets = synth.make_sequences(70, 60, 13, seed=4)
    cfg = Discovery(warping_band_percentage=0.0625).align_config()
    single = gpu_align_all(ctx, frames, offsets, 13, 0.0625)
    multi, seen = sharding.align_all_multi([0], frames, offsets, 13, cfg)
    assert seen == 1
    assert np.array_equal(multi.view(np.uint32), single.view(np.uint32))
    assert_parity(multi, oracle.align_all(frames, offsets, 0.0625, workers=8))
    with pytest.raises(apd.ApdError):
        sharding.align_all_multi([0, 0], frames, offsets, 13, cfg)         # two ranks on one device


def test_sharded_async_through_a_library_owned_communicator(ctx, oracle, apd):
    """One process per GPU form with world = 1: unique id -> apd_comm_create -> apd_align_all_sharded_async (tiles, ncclAllGather
    on the context's stream, unpack) equals apd_align_all_device_async bit for bit; apd_all_gather_async moves data."""
    from audio_pattern_discovery_amd import sharding
    from audio_pattern_discovery_amd.alignments import Batch
    from audio_pattern_discovery_amd.discovery import Discovery
    n = 70
    frames, offsets = synth.make_sequences(n, 60, 13, seed=4)
    cfg = Discovery(warping_band_percentage=0.0625).align_config()
    comm = sharding.Comm(ctx, sharding.Comm.unique_id(), 0, 1)
    assert comm.count() == 1 and comm.rank() == 0
    b = Batch(ctx, frames, offsets, 13)
    a, c = ctx.alloc(4 * n * n), ctx.alloc(4 * n * n)
    comm.align_all_sharded_async(b.handle, cfg, a.ptr)
    apd.check(apd.lib().apd_align_all_device_async(ctx.handle, b.handle, C.byref(cfg), c.at()), ctx.handle)
    ctx.synchronize()
    assert np.array_equal(a.to_numpy(np.uint32), c.to_numpy(np.uint32))
    assert_parity(a.to_numpy(np.float32).reshape(n, n), oracle.align_all(frames, offsets, 0.0625, workers=8))
    src = ctx.upload(np.arange(1000, dtype=np.float32))
    dst = ctx.alloc(4000)
    dst.fill(0)
    apd.check(apd.lib().apd_all_gather_async(ctx.handle, comm.handle, src.at(), dst.at(), 1000), ctx.handle)
    ctx.synchronize()
    assert np.array_equal(dst.to_numpy(np.float32), np.arange(1000, dtype=np.float32))
    comm.close()
    # destruction order is the caller's: a context may go before its communicator (and its batches)
    ctx2 = apd.Context(0)
    comm2 = sharding.Comm(ctx2, sharding.Comm.unique_id(), 0, 1)
    ctx2.close()
    with pytest.raises(apd.ApdError):
        comm2.count()                                                    # orphaned: refused
    comm2.close()


def test_async_alignment_only_enqueues(ctx, apd):
    """The _async entry points return while the kernels run: the choice between the fast kernels and the literal, NaN-faithful
    one is made on the device (the repack kernel's flag is read by the kernels, not by the host), so a refill + alignment is
    two enqueues.  Asserted on the stream itself (apd_stream_busy = hipStreamQuery), not on wall-clock ratios: when the calls
    return, the stream still has the work in it -- 4096 x len~512, a ~0.2 s launch."""
    from audio_pattern_discovery_amd.alignments import Batch
    n = 4096
    frames, offsets = synth.make_sequences(n, 512, 13, seed=99)
    cfg = apd.AlignConfig(0.0625, 1.0, 1.0, 1.0)
    L = apd.lib()
    d_frames = ctx.upload(frames)
    b = Batch(ctx, d_frames.ptr, offsets, 13, on_device=True)
    out = ctx.alloc(4 * n * n)
    apd.check(L.apd_align_all_device_async(ctx.handle, b.handle, C.byref(cfg), out.at()), ctx.handle)     # plans, code objects
    ctx.synchronize()
    assert not ctx.stream_busy()
    first = out.to_numpy(np.uint32)
    apd.check(L.apd_batch_refill(ctx.handle, b.handle, d_frames.at(), 1), ctx.handle)
    apd.check(L.apd_align_all_device_async(ctx.handle, b.handle, C.byref(cfg), out.at()), ctx.handle)
    assert ctx.stream_busy(), "the alignment had finished when apd_align_all_device_async returned: it waited somewhere"
    ctx.synchronize()
    assert not ctx.stream_busy()
    assert np.array_equal(out.to_numpy(np.uint32), first)

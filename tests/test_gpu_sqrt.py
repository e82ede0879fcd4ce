"""Strict mode's square root, exhaustively: `sqrt_rn_finite` (csrc/dtw_common.h: v_sqrt_f32 + an exact two-sided fix-up) must
return the bits of the correctly rounded square root -- Rust's `f32::sqrt`, /root/reference/src/numerics.rs:119 -- for EVERY f32
in its domain 2^-96 <= x < +INF.  The device compares it with the compiler's correctly rounded `sqrtf`; this file also anchors that
comparison partner against numpy's (IEEE) sqrt on the host for a sample, so the chain is fix-up == device sqrtf == IEEE."""
import numpy as np
import pytest

from audio_pattern_discovery_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return _lib.Context(0)


def test_sqrt_fixup_is_correctly_rounded_for_every_f32_in_its_domain(ctx):
    total_bad, seen = 0, 0
    hist = [0] * 5
    for first in range(0, 1 << 32, 1 << 30):                      # all 2^32 patterns, four launches
        bad, first_bad, h = ctx.selftest_sqrt(first, 1 << 30)
        assert bad == 0, "sqrt_rn_finite differs from sqrtf at bits 0x%08x (%d patterns in this quarter)" % (first_bad, bad)
        total_bad += bad
        hist = [a + b for a, b in zip(hist, h)]
    seen = sum(hist)
    # the domain: exponent fields 31 (2^-96) .. 254, every mantissa, sign +
    assert seen == (254 - 31 + 1) << 23
    # what the fix-up rests on: the bare instruction is never more than one ulp off
    assert hist[0] == 0 and hist[4] == 0, hist
    print("v_sqrt_f32 ulp offsets over the domain: -1: %d  0: %d  +1: %d" % (hist[1], hist[2], hist[3]))


def test_strict_distance_bits_with_tiny_zero_and_huge_differences(ctx):
    """Inputs outside the fix-up's domain take the general sequence: identical frames (d2 = 0), differences so small that
    d2 < 2^-96, subnormal squares -- scores stay bit-identical to the CPU oracle in strict mode."""
    from audio_pattern_discovery_amd.alignments import AlignmentWorkers, NDSequence
    from audio_pattern_discovery_amd.discovery import Discovery
    from oracle import binding as oracle

    rng = np.random.default_rng(5)
    n_seq, length, dim = 20, 90, 13
    base = rng.standard_normal((n_seq, length, dim)).astype(np.float32)
    base[1] = base[0]                                              # identical sequences: zeros everywhere on the diagonal path
    base[2] = base[0] * np.float32(1e-30)                          # squares underflow: d2 subnormal or 0
    base[3] = base[2] + np.float32(1e-38)
    base[4, ::3] = base[0, ::3]                                    # some identical frames
    base[5] = base[0] * np.float32(3e-20)                          # d2 around 1e-39 .. 1e-37: below 2^-96 = 1.3e-29
    base[6] = base[0] * np.float32(2e-15)                          # d2 around 2^-96
    base[7] = base[0] * np.float32(1e17)                           # large but below the 2^60 routing bound
    frames = base.reshape(-1, dim)
    offsets = np.arange(n_seq + 1, dtype=np.uint64) * length
    ctx.set_distance_mode(2)
    try:
        for pct in (0.1, 1.0):
            workers = AlignmentWorkers.new([NDSequence(base[s]) for s in range(n_seq)], ctx)
            got = workers.align_all(Discovery(warping_band_percentage=pct)).reshape(n_seq, n_seq)
            want = oracle.align_all(frames, offsets, pct, workers=4).reshape(n_seq, n_seq)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), \
                "strict mode differs from the oracle at %s" % (np.argwhere(got.view(np.uint32) != want.view(np.uint32))[:5],)
    finally:
        ctx.set_distance_mode(1)


def test_features_beyond_2_pow_60_take_the_literal_kernel(ctx):
    """|feature| >= 2^60 raises the batch flag (squared distances could overflow): the literal kernel computes +INF where the CPU does."""
    from audio_pattern_discovery_amd.alignments import AlignmentWorkers, NDSequence
    from audio_pattern_discovery_amd.discovery import Discovery
    from oracle import binding as oracle

    rng = np.random.default_rng(6)
    n_seq, length, dim = 6, 40, 13
    base = rng.standard_normal((n_seq, length, dim)).astype(np.float32)
    base[0] *= np.float32(3e19)                                    # squares overflow to +INF on the CPU as well
    base[1, 5, 2] = np.float32(2.0 ** 60)
    frames = base.reshape(-1, dim)
    offsets = np.arange(n_seq + 1, dtype=np.uint64) * length
    for mode in (1, 2):
        ctx.set_distance_mode(mode)
        try:
            workers = AlignmentWorkers.new([NDSequence(base[s]) for s in range(n_seq)], ctx)
            got = workers.align_all(Discovery(warping_band_percentage=0.2)).reshape(n_seq, n_seq)
        finally:
            ctx.set_distance_mode(1)
        want = oracle.align_all(frames, offsets, 0.2, workers=2).reshape(n_seq, n_seq)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))

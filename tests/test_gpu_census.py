"""Full-matrix parity census: the default distance form (hybrid norm expansion, fminf select) against strict mode over EVERY
entry of the BASELINE shapes.

Strict mode (apd_set_distance_mode 2) computes operation for operation what reference src/numerics.rs:114-120 and
src/alignments.rs:129-160 compute and is bit-identical to the CPU oracle (tests/test_gpu_fuzz.py; re-anchored here on >= 300
sampled entries per shape, compared BITWISE).  So default-vs-strict on the GPU is default-vs-reference-arithmetic over all
N(N-1) ordered pairs, at seconds of cost, where the sampled tests see 300-600 entries.  What could differ: the one known
deviation of the default mode -- a coincidental exact DELETE/INSERT tie of two rounded f32 sums, which the reference resolves to
MATCH (alignments.rs:153-159) and the fast distance forms do not see.

Two claims per shape, asserted separately:
  * STRICT MODE == ORACLE, BITWISE: >= 300 sampled entries per shape carry exactly the CPU oracle's bits (0 differing) -- the
    tolerance-compliant mode, whose speed bench.py reports as secondary.cfg3_strict.
  * DEFAULT MODE, KNOWN DEVIATION, BOUNDED: identical +INF / zero pattern; every entry within 1e-4 of the strict matrix EXCEPT at
    most kKnownDeviationCount entries per corpus, each within kKnownDeviationWorst -- the documented bound of the one known
    deviation (profiles/r03/census_sweep_cfg{2,3,4}.txt: 1-6 entries in 1e8 over 60-100 corpora per shape, worst 2.2e-3).  A count
    or a magnitude beyond the bound fails; nothing is pinned to one pair of one corpus."""
import ctypes as C

import numpy as np
import pytest

from audio_pattern_discovery_amd import synth

pytestmark = pytest.mark.gpu
u64p, f32p = C.POINTER(C.c_uint64), C.POINTER(C.c_float)
kKnownDeviationCount = 8          # entries beyond 1e-4 tolerated per 1.7e7-entry corpus (expected: 0 - 2; sweep mean 1 per cfg4 corpus)
kKnownDeviationWorst = 3e-3       # ... and none of them further off than this (worst seen in 2e9 swept entries: 2.2e-3)


@pytest.fixture(scope="module")
def ctx(apd):
    c = apd.Context(0)
    yield c
    c.close()


def both_modes(apd, ctx, d_frames, offsets, dim, pct):
    """(default-mode matrix, strict-mode matrix) of one resident batch."""
    L = apd.lib()
    n = len(offsets) - 1
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    batch = C.c_void_p()
    apd.check(L.apd_batch_create(ctx.handle, d_frames.at(), off.ctypes.data_as(u64p), n, dim, 1, C.byref(batch)), ctx.handle)
    cfg = apd.AlignConfig(pct, 1.0, 1.0, 1.0)
    d_out = ctx.alloc(4 * n * n)
    out = []
    for mode in ("hybrid", "strict"):
        ctx.set_distance_mode(mode)
        apd.check(L.apd_align_all_device_async(ctx.handle, batch, C.byref(cfg), d_out.at()), ctx.handle)
        ctx.synchronize()
        out.append(d_out.to_numpy(np.float32).reshape(n, n))
    ctx.set_distance_mode("hybrid")
    L.apd_batch_destroy(batch)
    return out


def census(default, strict):
    n = default.shape[0]
    assert np.array_equal(np.isposinf(default), np.isposinf(strict)) and not np.isnan(default).any() and not np.isnan(strict).any()
    assert not np.isneginf(default).any() and not np.isneginf(strict).any()
    assert np.array_equal(default == 0, strict == 0)
    assert np.all(np.diag(default) == 0)
    m = np.isfinite(strict) & (strict != 0)
    rel = np.abs(default[m] - strict[m]) / np.abs(strict[m])
    over = int((rel > 1e-4).sum())
    worst = None
    if over:
        flat = int(np.flatnonzero(m)[int(np.argmax(rel))])
        worst = (flat // n, flat % n)
    return float(rel.max()), over, worst, int(m.sum())


def assert_known_deviation_bound(name, max_rel, over, worst):
    """The default mode's documented bound: a handful of entries beyond 1e-4 per corpus at most, none beyond 3e-3."""
    assert over <= kKnownDeviationCount, "%s: %d entries beyond 1e-4 (bound %d), worst pair %s (%.3e)" % (name, over, kKnownDeviationCount, worst, max_rel)
    assert max_rel <= (kKnownDeviationWorst if over else 1e-4), "%s: worst entry %.3e at %s" % (name, max_rel, worst)


def anchor_strict(oracle, strict, frames, offsets, pct, k, seed):
    """STRICT MODE == ORACLE, BITWISE: k sampled entries of the strict matrix, 0 differing from the CPU oracle's bits."""
    n = len(offsets) - 1
    rng = np.random.default_rng(seed)
    pi = rng.integers(0, n, k).astype(np.uint32)
    pj = (pi + 1 + rng.integers(0, n - 1, k)).astype(np.uint32) % n
    want, _ = oracle.align_sample(frames, offsets, pi, pj, pct, workers=8)
    differing = int((strict[pi, pj].view(np.uint32) != want.view(np.uint32)).sum())
    assert differing == 0, "strict mode: %d of %d sampled entries differ bitwise from the oracle" % (differing, k)


@pytest.mark.parametrize("name,n,length,seed", [("cfg2", 1024, 512, 0xA9D2), ("cfg3", 4096, 1024, 0xA9D3)])
def test_strict_bitwise_and_default_known_deviation_bound_mfcc_shapes(ctx, apd, oracle, name, n, length, seed):
    frames, offsets = synth.make_sequences(n, length, 13, seed=seed)
    default, strict = both_modes(apd, ctx, ctx.upload(frames), offsets, 13, 0.0625)
    anchor_strict(oracle, strict, frames, offsets, 0.0625, 320, seed)
    max_rel, over, worst, counted = census(default, strict)
    assert counted == n * (n - 1)                                       # every ordered pair is finite and non-zero here
    assert_known_deviation_bound(name, max_rel, over, worst)           # these corpora: 0 entries, 4e-7 (cfg2) and 1e-5 (cfg3)


def test_strict_bitwise_and_default_known_deviation_bound_cfg4_through_the_encoder(ctx, apd, oracle):
    n = 4096
    frames, offsets = synth.make_sequences(n, 1024, 13, seed=0xA9D4)
    rng = np.random.default_rng(0xE1C)
    w = ((rng.random((13, 8)) - 0.5) / 8).astype(np.float32)
    b = ((rng.random(8) - 0.5) / 8).astype(np.float32)
    total = int(offsets[-1])
    d_src, d_lat = ctx.upload(frames), ctx.alloc(4 * total * 8)
    apd.check(apd.lib().apd_encode(ctx.handle, d_src.at(), total, 13, w.ctypes.data_as(f32p), b.ctypes.data_as(f32p), 8, 1, d_lat.at()), ctx.handle)
    default, strict = both_modes(apd, ctx, d_lat, offsets, 8, 0.0625)
    lat = d_lat.to_numpy(np.float32).reshape(total, 8)                   # the device's own embeddings: isolates the alignment
    anchor_strict(oracle, strict, lat, offsets, 0.0625, 320, 4)
    max_rel, over, worst, counted = census(default, strict)
    assert counted == n * (n - 1)
    # This corpus holds ONE unordered pair (both ordered entries, 3.0e-4) of the known deviation: somewhere on that pair's optimal
    # path the reference's DELETE and INSERT predecessors are bit-equal f32 sums and alignments.rs:153-159 then takes the larger
    # MATCH predecessor; the fast distance forms differ in the last bit, see no tie and keep the smaller one.  Strict mode
    # reproduces it (the anchor above is bitwise).  Asserted: the documented bound, not this pair.
    assert_known_deviation_bound("cfg4", max_rel, over, worst)


def test_strict_bitwise_and_default_known_deviation_bound_cfg5_shape_from_device_cepstra(ctx, apd, oracle):
    n, n_samp = 256, 256 + 128 * 2048
    rng = np.random.default_rng(0xC5)
    base = [synth.make_audio(n_samp, seed=500 + k) for k in range(16)]
    audio = []
    for k in range(n):
        amp = 150 * (k // 16)
        noise = rng.integers(-amp, amp + 1, n_samp, dtype=np.int32) if amp else 0
        audio.append(np.clip(base[k % 16].astype(np.int32) + noise, -32768, 32767).astype(np.int16))
    s_off = (np.arange(n + 1, dtype=np.uint64) * n_samp)
    d_audio, d_ceps = ctx.upload(np.concatenate(audio)), ctx.alloc(4 * n * 2048 * 13)
    f_off, nb = np.zeros(n + 1, dtype=np.uint64), C.c_uint32(0)
    apd.check(apd.lib().apd_cepstrum_batch(ctx.handle, d_audio.at(), s_off.ctypes.data_as(u64p), n, 256, 128, 18, 1, d_ceps.at(),
                                           f_off.ctypes.data_as(u64p), C.byref(nb)), ctx.handle)
    default, strict = both_modes(apd, ctx, d_ceps, f_off, 13, 0.0625)
    ceps = d_ceps.to_numpy(np.float32).reshape(n * 2048, 13)
    anchor_strict(oracle, strict, ceps, f_off, 0.0625, 300, 5)
    max_rel, over, worst, counted = census(default, strict)
    assert counted == n * (n - 1)                                       # 16 distinct recordings and noisy takes of them: no exact repeat
    assert_known_deviation_bound("cfg5 shape", max_rel, over, worst)


def test_strict_mode_is_bitwise_the_oracle_on_every_entry_of_cfg2(ctx, apd, oracle):
    """Not a sample: all 1 047 552 ordered pairs of the cfg 2 corpus, strict mode against the CPU oracle, bit for bit (the oracle
    needs ~15 s for them on 16 threads).  The same comparison over all 16.8 M pairs of cfg 3 takes the oracle 13 minutes:
    tools/strict_full_check.py, result in profiles/r04/strict_full_matrix_cfg3.txt."""
    n = 1024
    frames, offsets = synth.make_sequences(n, 512, 13, seed=0xA9D2)
    _, strict = both_modes(apd, ctx, ctx.upload(frames), offsets, 13, 0.0625)
    want = oracle.align_all(frames, offsets, 0.0625, workers=16)
    differing = int((strict.view(np.uint32) != want.view(np.uint32)).sum())
    assert differing == 0, "%d of %d entries differ bitwise" % (differing, n * n)

"""BASELINE.json configurations 4 and 5 at their real per-pair shape, through the C ABI with everything resident in HBM
(the cfg 2 / cfg 3 twins live in test_gpu_dtw.py):

* cfg 5: recordings of 262 400 i16 samples -> apd_cepstrum_batch (dft_win 256, step 128, ceps_filter 18: 13 bins x 2048
  frames, spectrogram.rs:31-80) -> apd_batch_create(frames_on_device) -> apd_align_all_device_async at
  warping_band_percentage 0.0625 (band 128, w = 130), 256 recordings instead of 16 384 (the per-pair work is the same).
* cfg 4: 4096 sequences x len~1024 x 13 -> apd_encode (13 -> 8, neural.rs:55-71) -> alignment, at full size.

The oracle cannot run either in full: sampled entries + size-independent properties (zero diagonal, no +INF, no NaN,
exact zero for an exact copy).  Tolerances: DTW 1e-4 relative on identical input features (north_star); cepstrum
3e-4 absolute (f32 FFT vs the oracle's f64 DFT; rustfft/rustdct parity is unpinned, see oracle/apd_oracle.h)."""
import ctypes as C

import numpy as np
import pytest

from audio_pattern_discovery_amd import synth

pytestmark = pytest.mark.gpu
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)


@pytest.fixture(scope="module")
def ctx(apd):
    c = apd.Context(0)
    yield c
    c.close()


def rel_err(got, want):
    nz = want != 0
    return float(np.max(np.abs(got[nz] - want[nz]) / np.abs(want[nz]))) if nz.any() else 0.0


def test_cfg5_shape_audio_to_matrix_on_device(ctx, oracle, apd):
    L = apd.lib()
    n, n_samp = 256, 256 + 128 * 2048                                   # 262 400 samples -> 2048 frames
    rng = np.random.default_rng(0xC5)
    base = [synth.make_audio(n_samp, seed=500 + k) for k in range(16)]
    audio = []
    for k in range(n):                                                   # 16 families of progressively noisier takes
        amp = 150 * (k // 16)
        noise = rng.integers(-amp, amp + 1, n_samp, dtype=np.int32) if amp else 0
        audio.append(np.clip(base[k % 16].astype(np.int32) + noise, -32768, 32767).astype(np.int16))
    audio[200] = audio[40].copy()                                        # an exact repeat: both ordered scores exactly 0
    s_off = (np.arange(n + 1, dtype=np.uint64) * n_samp)
    d_audio = ctx.upload(np.concatenate(audio))
    f_off = np.zeros(n + 1, dtype=np.uint64)
    nb = C.c_uint32(0)
    d_ceps = ctx.alloc(4 * n * 2048 * 13)
    apd.check(L.apd_cepstrum_batch(ctx.handle, d_audio.at(), s_off.ctypes.data_as(u64p), n, 256, 128, 18, 1,
                                   d_ceps.at(), f_off.ctypes.data_as(u64p), C.byref(nb)), ctx.handle)
    assert nb.value == 13 and f_off.tolist() == [2048 * k for k in range(n + 1)]
    batch = C.c_void_p()
    apd.check(L.apd_batch_create(ctx.handle, d_ceps.at(), f_off.ctypes.data_as(u64p), n, 13, 1, C.byref(batch)), ctx.handle)
    cfg = apd.AlignConfig(0.0625, 1.0, 1.0, 1.0)
    d_out = ctx.alloc(4 * n * n)
    d_out.fill(0xFF)                                                     # NaN
    apd.check(L.apd_align_all_device_async(ctx.handle, batch, C.byref(cfg), d_out.at()), ctx.handle)
    ctx.synchronize()
    got = d_out.to_numpy(np.float32).reshape(n, n)
    ceps = d_ceps.to_numpy(np.float32).reshape(n, 2048, 13)
    L.apd_batch_destroy(batch)
    # the UPGMA leg on the RESIDENT matrix (cepstrum -> DTW -> UPGMA never leaves HBM), against the O(n^3) oracle restatement of
    # clustering.rs:81-209 on the same matrix: ops, kinds, roots, threshold and linkage BITS
    ops = (apd.ClusterOp * n)()
    roots = np.zeros(n, dtype=np.uint32)
    n_ops, n_roots, thr = C.c_uint32(0), C.c_uint32(0), C.c_float(0)
    for perc in (0.05, 0.5):
        apd.check(L.apd_clustering(ctx.handle, d_out.at(), 1, n, perc, ops, C.byref(n_ops), roots.ctypes.data_as(C.POINTER(C.c_uint32)),
                                   C.byref(n_roots), C.byref(thr)), ctx.handle)
        want_ops, want_roots, want_thr = oracle.clustering(got, n, perc, fast=True)
        assert n_ops.value == len(want_ops) and n_ops.value >= 1
        names = ["Sequence2Sequence", "Sequence2Cluster", "Cluster2Sequence", "Cluster2Cluster"]
        assert [(ops[k].merge_i, ops[k].merge_j, ops[k].into, names[ops[k].operation]) for k in range(n_ops.value)] == \
               [(o["merge_i"], o["merge_j"], o["into"], o["operation"]) for o in want_ops]
        assert np.array_equal(np.array([ops[k].distance for k in range(n_ops.value)], np.float32).view(np.uint32),
                              np.array([o["distance"] for o in want_ops], np.float32).view(np.uint32))
        assert roots[:n_roots.value].tolist() == sorted(want_roots)
        assert np.float32(thr.value).view(np.uint32) == np.float32(want_thr).view(np.uint32)
    assert (ops[0].merge_i, ops[0].merge_j) in ((40, 200), (200, 40))    # the exact repeat merges first (distance 0)
    # properties
    assert np.all(np.diag(got) == 0.0)
    off = ~np.eye(n, dtype=bool)
    assert np.all(np.isfinite(got[off])) and np.all(got[off] >= 0)
    assert got[40, 200] == 0.0 and got[200, 40] == 0.0
    assert np.count_nonzero(got[off] == 0.0) == 2
    # features of a subset of recordings against the oracle's cepstrum
    subset = sorted({int(v) for v in rng.integers(0, n, 24)} | {40, 200, 0, n - 1})
    for s in subset[:12]:
        want_c = oracle.cepstrum(audio[s], 256, 128, 18)
        assert want_c.shape == (2048, 13)
        np.testing.assert_allclose(ceps[s], want_c, rtol=0, atol=3e-4)
    # >= 300 sampled ordered pairs: the oracle's DTW on the device-made features (isolates the alignment: 1e-4) ...
    pi = rng.choice(subset, 340).astype(np.uint32)
    pj = rng.choice(subset, 340).astype(np.uint32)
    keep = pi != pj
    pi, pj = pi[keep], pj[keep]
    assert len(pi) >= 300
    remap = {s: k for k, s in enumerate(subset)}
    sub_frames = np.concatenate([ceps[s] for s in subset])
    sub_off = (np.arange(len(subset) + 1, dtype=np.uint64) * 2048)
    li = np.array([remap[int(v)] for v in pi], np.uint32)
    lj = np.array([remap[int(v)] for v in pj], np.uint32)
    want, cells = oracle.align_sample(sub_frames, sub_off, li, lj, 0.0625, workers=8)
    assert cells == len(pi) * 515580                                    # BASELINE cfg 5 cells per pair (w = 130)
    assert np.array_equal(want == 0, got[pi, pj] == 0) and np.all(np.isfinite(want))
    assert rel_err(got[pi, pj], want) <= 1e-4
    # ... and end to end against the oracle's own cepstra of a few recordings (cepstrum parity unpinned: 1e-3)
    few = subset[:6]
    o_frames = np.concatenate([oracle.cepstrum(audio[s], 256, 128, 18) for s in few])
    o_off = (np.arange(len(few) + 1, dtype=np.uint64) * 2048)
    want_e2e = oracle.align_all(o_frames, o_off, 0.0625, workers=8)
    sel = got[np.ix_(few, few)]
    m = ~np.eye(len(few), dtype=bool)
    assert rel_err(sel[m], want_e2e[m]) <= 1e-3


def test_cfg4_full_size_through_the_encoder(ctx, oracle, apd):
    L = apd.lib()
    n = 4096
    frames, offsets = synth.make_sequences(n, 1024, 13, seed=0xA9D4)
    rng = np.random.default_rng(0xE1C)
    w = ((rng.random((13, 8)) - 0.5) / 8).astype(np.float32)             # Mat::seeded scale (numerics.rs:178-186)
    b = ((rng.random(8) - 0.5) / 8).astype(np.float32)
    total = int(offsets[-1])
    d_src = ctx.upload(frames)
    d_lat = ctx.alloc(4 * total * 8)
    apd.check(L.apd_encode(ctx.handle, d_src.at(), total, 13, w.ctypes.data_as(f32p), b.ctypes.data_as(f32p), 8, 1, d_lat.at()), ctx.handle)
    off_c = np.ascontiguousarray(offsets, dtype=np.uint64)
    batch = C.c_void_p()
    apd.check(L.apd_batch_create(ctx.handle, d_lat.at(), off_c.ctypes.data_as(u64p), n, 8, 1, C.byref(batch)), ctx.handle)
    cfg = apd.AlignConfig(0.0625, 1.0, 1.0, 1.0)
    d_out = ctx.alloc(4 * n * n)
    d_out.fill(0xFF)                                                     # NaN
    apd.check(L.apd_align_all_device_async(ctx.handle, batch, C.byref(cfg), d_out.at()), ctx.handle)
    ctx.synchronize()
    got = d_out.to_numpy(np.float32).reshape(n, n)
    lat = d_lat.to_numpy(np.float32).reshape(total, 8)
    L.apd_batch_destroy(batch)
    assert np.all(np.diag(got) == 0.0)
    off = ~np.eye(n, dtype=bool)
    assert np.all(np.isfinite(got[off])) and np.all(got[off] > 0)
    # the encoder output of the whole corpus against the oracle (expf differs from libm by <= 2 ulp: 1e-5)
    want_lat = oracle.encode(frames, w, b)
    np.testing.assert_allclose(lat, want_lat, rtol=1e-5, atol=1e-5)
    pi = np.concatenate([rng.integers(0, n, 320), [0, n - 1]]).astype(np.uint32)
    pj = np.concatenate([rng.integers(0, n, 320), [n - 1, 0]]).astype(np.uint32)
    keep = pi != pj
    want, _ = oracle.align_sample(lat, offsets, pi[keep], pj[keep], 0.0625, workers=8)      # same features: isolates the alignment
    assert rel_err(got[pi[keep], pj[keep]], want) <= 1e-4
    want2, _ = oracle.align_sample(want_lat, offsets, pi[keep][:64], pj[keep][:64], 0.0625, workers=8)   # end to end
    assert rel_err(got[pi[keep][:64], pj[keep][:64]], want2) <= 1e-4

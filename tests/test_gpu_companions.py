"""GPU parity tests of the feature companions against the CPU oracle.

Encoder (neural.rs:55-71): same arithmetic order; expf differs from libm by <= 2 ulp -> 1e-5 relative.
Cepstrum (spectrogram.rs:31-80): the reference's FFT/DCT come from un-vendored, un-pinned crates (rustfft 3.0.0,
rustdct *), so this is PARITY UNPINNED against the reference; the oracle evaluates the mathematical definitions in
f64, the kernel an f32 radix-2 FFT: agreement to 2e-4 absolute on values of magnitude ~1..10."""
import os

import numpy as np
import pytest

from audio_pattern_discovery_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "companions.npz")


@pytest.fixture(scope="module")
def ctx(apd):
    c = apd.Context(0)
    yield c
    c.close()


def test_encoder_golden_and_random(ctx, oracle):
    from audio_pattern_discovery_amd.neural import AutoEncoder
    g = np.load(GOLD)
    nn = AutoEncoder(g["w"], g["b"])
    assert nn.n_latent() == 8
    np.testing.assert_allclose(nn.predict_frames(g["x"], ctx), g["enc"], rtol=1e-5, atol=1e-5)
    rng = np.random.default_rng(3)
    for d_in, latent, t in [(13, 8, 5000), (26, 10, 777), (13, 10, 1), (4, 3, 64)]:
        x = (rng.standard_normal((t, d_in)) * 3).astype(np.float32)
        w = ((rng.random((d_in, latent)) - 0.5) / latent).astype(np.float32)      # Mat::seeded scale (numerics.rs:182)
        b = ((rng.random(latent) - 0.5) / latent).astype(np.float32)
        got = AutoEncoder(w, b).predict_frames(x, ctx)
        np.testing.assert_allclose(got, oracle.encode(x, w, b), rtol=1e-5, atol=1e-5)
    # sigma is floored at 1.0 (neural.rs:62): constant latent -> zeros, not NaN
    z = AutoEncoder(np.zeros((13, 8), np.float32), np.zeros(8, np.float32)).predict_frames(np.ones((3, 13), np.float32), ctx)
    assert np.all(z == 0.0)


def test_encoded_sequence_feeds_alignment(ctx, oracle):
    """main.rs:150-161 then 187-195: cepstrum-like frames -> encoded(&nn) -> align_all, all on the GPU."""
    from audio_pattern_discovery_amd.alignments import AlignmentWorkers, NDSequence
    from audio_pattern_discovery_amd.discovery import Discovery
    from audio_pattern_discovery_amd.neural import AutoEncoder
    rng = np.random.default_rng(9)
    w = ((rng.random((13, 8)) - 0.5) / 8).astype(np.float32)
    b = ((rng.random(8) - 0.5) / 8).astype(np.float32)
    frames, offsets = synth.make_sequences(12, 50, 13, seed=9)
    nn = AutoEncoder(w, b)
    seqs = [NDSequence(s).encoded(nn, ctx) for s in synth.split(frames, offsets)]
    assert seqs[0].n_bins == 8
    got = AlignmentWorkers.new(seqs, ctx).align_all(Discovery(warping_band_percentage=0.0625)).reshape(12, 12)
    enc = oracle.encode(frames, w, b)
    want = oracle.align_all(enc, offsets, 0.0625, workers=4)
    np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-6)


def test_cepstrum_golden_and_shapes(ctx, oracle):
    from audio_pattern_discovery_amd.alignments import NDSequence
    g = np.load(GOLD)
    s13 = NDSequence.new(256, 128, 18, g["audio"], ctx)            # ceps_filter 18 -> 13 bins
    s26 = NDSequence.new(256, 128, 32, g["audio"], ctx)            # shipped config -> 26 bins
    assert (s13.len(), s13.n_bins) == g["ceps13"].shape and (s26.len(), s26.n_bins) == g["ceps26"].shape
    np.testing.assert_allclose(s13.frames, g["ceps13"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(s26.frames, g["ceps26"], rtol=0, atol=2e-4)
    # frame count: i in (fft_size..n).step_by(step) (spectrogram.rs:51); n <= fft_size -> no frame
    assert NDSequence.new(256, 128, 18, g["audio"][:256], ctx).len() == 0
    assert NDSequence.new(256, 128, 18, g["audio"][:257], ctx).len() == 1
    assert NDSequence.new(256, 128, 18, g["audio"][:385], ctx).len() == 2


@pytest.mark.parametrize("fft,step,filt", [(256, 128, 18), (256, 64, 32), (512, 256, 32), (128, 32, 8), (1024, 512, 64),
                                           (300, 128, 18), (250, 100, 20), (441, 220, 21), (1000, 500, 50), (96, 33, 12)])
def test_cepstrum_random_audio(ctx, oracle, fft, step, filt):
    from audio_pattern_discovery_amd.alignments import NDSequence
    audio = synth.make_audio(fft + step * 37 + 3, seed=fft + filt)
    got = NDSequence.new(fft, step, filt, audio, ctx)
    want = oracle.cepstrum(audio, fft, step, filt)
    assert got.frames.shape == want.shape
    np.testing.assert_allclose(got.frames, want, rtol=0, atol=3e-4)
    assert np.abs(got.frames.mean(axis=1)).max() < 1e-4            # mean-centred bins (spectrogram.rs:74-75)


def test_cepstrum_window_limits(ctx, apd):
    """Any window length up to 4096 (rustfft plans any length, spectrogram.rs:44-48): powers of two by FFT, the others by the
    defining sum.  Beyond the limit, and filter banks that leave fewer than 5 outputs, are refused."""
    from audio_pattern_discovery_amd.alignments import NDSequence
    assert NDSequence.new(300, 128, 18, synth.make_audio(2000, seed=1), ctx).n_bins > 0
    with pytest.raises(apd.ApdError) as e:
        NDSequence.new(8192, 128, 18, synth.make_audio(20000, seed=1), ctx)
    assert e.value.status == apd.APD_ERR_UNSUPPORTED
    with pytest.raises(apd.ApdError) as e:
        NDSequence.new(256, 128, 2, synth.make_audio(2000, seed=1), ctx)        # L = 128: no filter output
    assert e.value.status == apd.APD_ERR_INVALID_ARG


def test_encoder_shapes_and_alignment(ctx, oracle, apd):
    """The staged encoder (64 frames per wavefront through LDS): frame counts around the chunk size, even and odd dimensions
    (row padding), an output that does not start on a 16-byte boundary, and the un-staged fallback for wide frames."""
    import ctypes as C
    from audio_pattern_discovery_amd.neural import AutoEncoder
    rng = np.random.default_rng(11)
    for d_in, latent, t in [(13, 8, 63), (13, 8, 64), (13, 8, 65), (26, 10, 1000), (12, 7, 129), (16, 16, 257), (3, 2, 1), (200, 40, 300)]:
        x = (rng.standard_normal((t, d_in)) * 2).astype(np.float32)
        w = ((rng.random((d_in, latent)) - 0.5) / latent).astype(np.float32)
        b = ((rng.random(latent) - 0.5) / latent).astype(np.float32)
        np.testing.assert_allclose(AutoEncoder(w, b).predict_frames(x, ctx), oracle.encode(x, w, b), rtol=1e-5, atol=1e-5)
    # device pointers one float off a 16-byte boundary
    d_in, latent, t = 13, 8, 500
    x = (rng.standard_normal((t, d_in)) * 2).astype(np.float32)
    w = ((rng.random((d_in, latent)) - 0.5) / latent).astype(np.float32)
    b = ((rng.random(latent) - 0.5) / latent).astype(np.float32)
    d_x = ctx.alloc(4 * (t * d_in + 1))
    d_x.copy_from(x.ravel(), byte_offset=4)
    d_z = ctx.alloc(4 * (t * latent + 3))
    d_z.fill(0)
    f32p = C.POINTER(C.c_float)
    apd.check(apd.lib().apd_encode(ctx.handle, d_x.at(4), t, d_in, w.ctypes.data_as(f32p), b.ctypes.data_as(f32p), latent, 1,
                                   d_z.at(12)), ctx.handle)
    ctx.synchronize()
    np.testing.assert_allclose(d_z.to_numpy(np.float32)[3:].reshape(t, latent), oracle.encode(x, w, b), rtol=1e-5, atol=1e-5)


def test_audio_to_clusters_on_device(ctx, oracle, apd):
    """cfg 5's data path at toy size, everything resident in HBM: i16 audio of many recordings -> apd_cepstrum_batch ->
    apd_encode -> apd_batch_create(frames_on_device) -> align_all, against the oracle pipeline (main.rs:150-161 then
    187-195).  Cepstrum parity is unpinned (see module docstring): DTW costs agree to 1e-3."""
    import ctypes as C
    L = apd.lib()
    rng = np.random.default_rng(5)
    lens = rng.integers(256 + 128 * 20, 256 + 128 * 40, size=12)
    audio = [synth.make_audio(int(n), seed=100 + i) for i, n in enumerate(lens)]
    audio[7] = audio[2].copy()                                             # a repeated recording: distance exactly 0
    samples = np.concatenate(audio)
    s_off = np.concatenate([[0], np.cumsum([len(a) for a in audio])]).astype(np.uint64)
    w = ((rng.random((13, 8)) - 0.5) / 8).astype(np.float32)
    b = ((rng.random(8) - 0.5) / 8).astype(np.float32)
    n = len(audio)
    d_samples = ctx.upload(samples)
    f_off = np.zeros(n + 1, dtype=np.uint64)
    nb = C.c_uint32(0)
    u64p = C.POINTER(C.c_uint64)
    apd.check(L.apd_cepstrum_batch(ctx.handle, d_samples.at(), s_off.ctypes.data_as(u64p), n, 256, 128, 18, 1,
                                   None, f_off.ctypes.data_as(u64p), C.byref(nb)), ctx.handle)
    assert nb.value == 13
    total = int(f_off[-1])
    d_ceps = ctx.alloc(4 * total * 13)
    apd.check(L.apd_cepstrum_batch(ctx.handle, d_samples.at(), s_off.ctypes.data_as(u64p), n, 256, 128, 18, 1,
                                   d_ceps.at(), f_off.ctypes.data_as(u64p), C.byref(nb)), ctx.handle)
    d_lat = ctx.alloc(4 * total * 8)
    f32p = C.POINTER(C.c_float)
    apd.check(L.apd_encode(ctx.handle, d_ceps.at(), total, 13, w.ctypes.data_as(f32p), b.ctypes.data_as(f32p), 8, 1, d_lat.at()), ctx.handle)
    batch = C.c_void_p()
    apd.check(L.apd_batch_create(ctx.handle, d_lat.at(), f_off.ctypes.data_as(u64p), n, 8, 1, C.byref(batch)), ctx.handle)
    cfg = apd.AlignConfig(0.0625, 1, 1, 1)
    got = np.empty((n, n), dtype=np.float32)
    apd.check(L.apd_align_all(ctx.handle, batch, C.byref(cfg), got.ctypes.data_as(f32p)), ctx.handle)
    L.apd_batch_destroy(batch)
    # oracle pipeline
    feats = [oracle.encode(oracle.cepstrum(a, 256, 128, 18), w, b) for a in audio]
    assert [len(f) for f in feats] == np.diff(f_off).astype(int).tolist()
    want = oracle.align_all(np.concatenate(feats), f_off, 0.0625, workers=4)
    np.testing.assert_allclose(d_ceps.to_numpy(np.float32).reshape(total, 13), np.concatenate([oracle.cepstrum(a, 256, 128, 18) for a in audio]),
                               rtol=0, atol=3e-4)
    assert got[2, 7] == 0.0 and got[7, 2] == 0.0
    np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-5)


def test_interesting_ranges_vat(ctx, oracle, apd):
    """NDSequence::interesting_ranges (spectrogram.rs:174-216): GPU variance + percentile select vs the oracle."""
    from audio_pattern_discovery_amd.alignments import NDSequence
    rng = np.random.default_rng(4)
    f = rng.standard_normal((3000, 13)).astype(np.float32)
    for a, b, g in [(100, 400, 5.0), (900, 1000, 7.0), (1500, 1510, 9.0), (2000, 2600, 3.0)]:
        f[a:b] *= g
    seq = NDSequence(f)
    for moving, perc, min_len in [(15, 0.5, 150), (15, 0.95, 5), (1, 0.3, 0), (40, 0.8, 60)]:
        got = [(s.start, s.stop) for s in seq.interesting_ranges(moving, perc, min_len, ctx)]
        assert got == oracle.interesting_ranges(f, moving, perc, min_len)
    sl = seq.interesting_ranges(15, 0.5, 150, ctx)[0]
    assert sl.extract().len() == sl.len() and sl.extract().n_bins == 13
    with pytest.raises(apd.ApdError) as e:
        seq.interesting_ranges(15, 1.0, 10, ctx)                          # percentile index == len: the reference panics
    assert e.value.status == apd.APD_ERR_INDEX


def test_async_feature_stage_only_enqueues(apd, oracle):
    """apd_encoder / apd_cepstrum_plan: weights, tables and offsets uploaded once; apd_encode_async / apd_cepstrum_batch_async only
    enqueue (the stream is still busy when they return: no allocation, no table building, no synchronisation inside), and their
    output is bit-identical to the one-call forms apd_encode / apd_cepstrum_batch."""
    import ctypes as C
    L = apd.lib()
    ctx = apd.Context(0)
    f32p, u64p = C.POINTER(C.c_float), C.POINTER(C.c_uint64)
    rng = np.random.default_rng(8)
    # encoder: 8 M frames of 13 -> 8 (a ~0.3 ms kernel), enqueued 40 times
    t, d_in, latent = 8_000_000, 13, 8
    x = rng.standard_normal((t, d_in)).astype(np.float32)
    w = ((rng.random((d_in, latent)) - 0.5) / latent).astype(np.float32)
    b = ((rng.random(latent) - 0.5) / latent).astype(np.float32)
    d_x, d_ref, d_out = ctx.upload(x), ctx.alloc(4 * t * latent), ctx.alloc(4 * t * latent)
    apd.check(L.apd_encode(ctx.handle, d_x.at(), t, d_in, w.ctypes.data_as(f32p), b.ctypes.data_as(f32p), latent, 1, d_ref.at()), ctx.handle)
    enc = C.c_void_p()
    apd.check(L.apd_encoder_create(ctx.handle, w.ctypes.data_as(f32p), b.ctypes.data_as(f32p), d_in, latent, C.byref(enc)), ctx.handle)
    apd.check(L.apd_encode_async(ctx.handle, enc, d_x.at(), t, d_out.at()), ctx.handle)               # code object loaded
    ctx.synchronize()
    for _ in range(40):
        apd.check(L.apd_encode_async(ctx.handle, enc, d_x.at(), t, d_out.at()), ctx.handle)
    assert ctx.stream_busy(), "40 encoder launches had all finished when the last call returned"
    ctx.synchronize()
    assert np.array_equal(d_out.to_numpy(np.uint32), d_ref.to_numpy(np.uint32))
    # cepstrum plan: 64 recordings x 262400 samples, enqueued 20 times
    n, n_samp = 64, 262400
    audio = rng.integers(-20000, 20000, n * n_samp, dtype=np.int16)
    s_off = np.arange(n + 1, dtype=np.uint64) * n_samp
    d_audio = ctx.upload(audio)
    f_off, f_off2, nb, nb2 = np.zeros(n + 1, np.uint64), np.zeros(n + 1, np.uint64), C.c_uint32(0), C.c_uint32(0)
    apd.check(L.apd_cepstrum_batch(ctx.handle, d_audio.at(), s_off.ctypes.data_as(u64p), n, 256, 128, 18, 1, None, f_off.ctypes.data_as(u64p), C.byref(nb)), ctx.handle)
    total = int(f_off[-1])
    d_cref, d_c = ctx.alloc(4 * total * nb.value), ctx.alloc(4 * total * nb.value)
    apd.check(L.apd_cepstrum_batch(ctx.handle, d_audio.at(), s_off.ctypes.data_as(u64p), n, 256, 128, 18, 1, d_cref.at(), f_off.ctypes.data_as(u64p), C.byref(nb)), ctx.handle)
    plan = C.c_void_p()
    apd.check(L.apd_cepstrum_plan_create(ctx.handle, s_off.ctypes.data_as(u64p), n, 256, 128, 18, f_off2.ctypes.data_as(u64p), C.byref(nb2), C.byref(plan)), ctx.handle)
    assert np.array_equal(f_off, f_off2) and nb.value == nb2.value == 13
    apd.check(L.apd_cepstrum_batch_async(ctx.handle, plan, d_audio.at(), d_c.at()), ctx.handle)
    ctx.synchronize()
    for _ in range(20):
        apd.check(L.apd_cepstrum_batch_async(ctx.handle, plan, d_audio.at(), d_c.at()), ctx.handle)
    assert ctx.stream_busy(), "20 cepstrum launches had all finished when the last call returned"
    ctx.synchronize()
    assert np.array_equal(d_c.to_numpy(np.uint32), d_cref.to_numpy(np.uint32))
    # the objects outlive nothing: destroyed before or after their context, either order is legal
    apd.check(L.apd_encoder_destroy(enc))
    ctx.close()
    apd.check(L.apd_cepstrum_plan_destroy(plan))                   # orphaned by apd_destroy: only the host part is left to free

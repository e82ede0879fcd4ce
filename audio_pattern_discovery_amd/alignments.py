"""Host mirror of src/alignments.rs (+ the NDSequence container of src/spectrogram.rs:13-24,99-101,
152-154) over the C ABI of include/apd.h.  Names, argument meaning and error behaviour follow the
Rust items; the compute is the HIP library, never Python."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib


class NDSequence:
    """spectrogram.rs:13-24: flat row-major frames `[T][n_bins]` plus n_bins."""

    def __init__(self, frames, n_bins=None, audio_id=0):
        a = np.ascontiguousarray(frames, dtype=np.float32)
        if a.ndim == 1:
            if not n_bins:
                raise ValueError("n_bins required for flat frames")
            a = a.reshape(-1, n_bins)
        self.frames = a
        self.n_bins = int(a.shape[1])
        self.audio_id = audio_id

    @staticmethod
    def new(fft_size, fft_step, filter_size, raw_audio, ctx=None):
        """NDSequence::new (spectrogram.rs:31-94), cepstrum branch: `raw_audio` is the i16 sample vector
        (AudioData.data, audio.rs:10-14).  The plotting-only z-scored spectrogram is not produced."""
        ctx = ctx or _lib.default_context()
        s = np.ascontiguousarray(getattr(raw_audio, "data", raw_audio), dtype=np.int16)
        nf, nb = C.c_uint64(0), C.c_uint32(0)
        L = _lib.lib()
        _lib.check(L.apd_cepstrum(ctx.handle, C.c_void_p(s.ctypes.data), s.size, fft_size, fft_step, filter_size, 0, None,
                                  C.byref(nf), C.byref(nb)), ctx.handle)
        out = np.empty((nf.value, nb.value), dtype=np.float32)
        if nf.value:
            _lib.check(L.apd_cepstrum(ctx.handle, C.c_void_p(s.ctypes.data), s.size, fft_size, fft_step, filter_size, 0,
                                      C.c_void_p(out.ctypes.data), C.byref(nf), C.byref(nb)), ctx.handle)
        return NDSequence(out, nb.value, getattr(raw_audio, "id", 0))

    def encoded(self, nn, ctx=None):        # spectrogram.rs:103-121
        return NDSequence(nn.predict_frames(self.frames, ctx), nn.n_latent(), self.audio_id)

    def interesting_ranges(self, moving_average, perc, min_len, ctx=None):
        """spectrogram.rs:192-216 -> list of Slice(start, stop)."""
        ctx = ctx or _lib.default_context()
        t = self.len()
        ranges = np.zeros(2 * max(t, 1), dtype=np.uint64)
        n = C.c_uint64(0)
        _lib.check(_lib.lib().apd_interesting_ranges(ctx.handle, C.c_void_p(self.frames.ctypes.data), t, self.n_bins,
                                                     int(moving_average), float(perc), int(min_len), 0,
                                                     ranges.ctypes.data_as(C.POINTER(C.c_uint64)), t, C.byref(n)), ctx.handle)
        return [Slice(int(ranges[2 * i]), int(ranges[2 * i + 1]), self) for i in range(n.value)]

    def vec(self, t):                       # spectrogram.rs:99-101
        return self.frames[t]

    def len(self):                          # spectrogram.rs:152-154
        return int(self.frames.shape[0])

    __len__ = len


@dataclass
class Slice:
    """spectrogram.rs:222-243: a frame range of a sequence."""
    start: int
    stop: int
    sequence: "NDSequence" = None

    def len(self):
        return self.stop - self.start

    def extract(self):                      # spectrogram.rs:245-262
        return NDSequence(self.sequence.frames[self.start:self.stop], self.sequence.n_bins, self.sequence.audio_id)


@dataclass
class AlignmentParams:
    """alignments.rs:77-83"""
    warping_band: int
    insertion_penalty: float = 1.0
    deletion_penalty: float = 1.0
    match_penalty: float = 1.0

    @staticmethod
    def default(length):                    # alignments.rs:86-93
        return AlignmentParams(int(length), 1.0, 1.0, 1.0)


class Alignment:
    """alignments.rs:99-181.  `sparse` (the DP table) is never read by any caller of the reference
    (only score() is), so it is not materialised; construct_alignment runs the HIP kernel."""

    def __init__(self, ctx=None):           # Alignment::new, :107-111
        self.n = 0
        self.m = 0
        self._score = float("inf")
        self._ctx = ctx

    def construct_alignment(self, x, y, params):     # :165-180
        ctx = self._ctx or _lib.default_context()
        xs = np.ascontiguousarray(x.frames if isinstance(x, NDSequence) else x, dtype=np.float32)
        ys = np.ascontiguousarray(y.frames if isinstance(y, NDSequence) else y, dtype=np.float32)
        dim = xs.shape[1] if xs.ndim == 2 and xs.shape[1] else ys.shape[1]
        self.n, self.m = int(xs.shape[0]), int(ys.shape[0])
        p = _lib.AlignmentParamsC(int(params.warping_band), params.insertion_penalty,
                                  params.deletion_penalty, params.match_penalty)
        out = C.c_float(0)
        _lib.check(_lib.lib().apd_align_pair(ctx.handle, xs.ctypes.data_as(C.POINTER(C.c_float)), self.n,
                                             ys.ctypes.data_as(C.POINTER(C.c_float)), self.m, int(dim),
                                             C.byref(p), C.byref(out)), ctx.handle)
        self._score = float(out.value)

    def score(self):                        # :116-125
        if self.n == 0 and self.m == 0:
            return float("inf")
        return self._score


class Batch:
    """apd_batch: Arc<Vec<NDSequence>> resident in HBM."""

    def __init__(self, ctx, frames, offsets, dim, on_device=False):
        self.ctx = ctx
        self.offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.n_seq = len(self.offsets) - 1
        self.dim = int(dim)
        self.handle = C.c_void_p()
        if on_device:
            ptr = C.c_void_p(int(frames))
        else:
            self._host = np.ascontiguousarray(frames, dtype=np.float32)
            ptr = C.c_void_p(self._host.ctypes.data)
        _lib.check(_lib.lib().apd_batch_create(ctx.handle, ptr, self.offsets.ctypes.data_as(C.POINTER(C.c_uint64)),
                                               self.n_seq, self.dim, int(bool(on_device)), C.byref(self.handle)),
                   ctx.handle)

    def close(self):
        if getattr(self, "handle", None):
            _lib.lib().apd_batch_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class AlignmentWorkers:
    """alignments.rs:11-68.  `new(data)` takes the sequences, `align_all(params)` blocks until
    `result` (n*n, row-major, diagonal 0.0) is filled.  `alignment_workers` is accepted and ignored:
    the GPU grid replaces the OS threads of :35-41."""

    def __init__(self, data, ctx=None, devices=None):     # AlignmentWorkers::new, :17-26
        """`devices`: HIP device ordinals -- the workers of :33-41 become these GPUs, driven through the library's persistent
        multi-device handle (made here once; every align_all then pays kernels + one all-gather + unpack)."""
        self._multi = None
        if devices is not None:
            from . import sharding
            self._multi = sharding.Multi(devices)
            ctx = self._multi.contexts[0]
        self.ctx = ctx or _lib.default_context()
        self.data = list(data)
        n = len(self.data)
        self.result = np.zeros(n * n, dtype=np.float32)
        dims = {s.n_bins for s in self.data}
        if len(dims) > 1:
            raise ValueError("all sequences must share n_bins")
        self._dim = dims.pop() if dims else 1
        lens = [s.len() for s in self.data]
        offsets = np.zeros(n + 1, dtype=np.uint64)
        offsets[1:] = np.cumsum(lens)
        frames = (np.concatenate([s.frames for s in self.data], axis=0) if n
                  else np.zeros((0, self._dim), np.float32))
        self._batch = (self._multi.batch(offsets, self._dim, frames=frames) if self._multi is not None
                       else Batch(self.ctx, frames, offsets, self._dim))

    @staticmethod
    def new(data, ctx=None, devices=None):
        return AlignmentWorkers(data, ctx, devices)

    def align_all(self, params):            # :31-67, params: Discovery
        cfg = params.align_config()
        if self._multi is not None:
            self.result[:] = self._multi.align_all(self._batch, cfg).ravel()
            return self.result
        _lib.check(_lib.lib().apd_align_all(self.ctx.handle, self._batch.handle, C.byref(cfg),
                                            self.result.ctypes.data_as(C.POINTER(C.c_float))), self.ctx.handle)
        return self.result

    def close(self):
        """Releases the GPU side (the reference's Drop of the Arcs)."""
        if self._batch is not None:
            self._batch.close()
            self._batch = None
        if self._multi is not None:
            self._multi.close()
            self._multi = None


def align_work(offsets, dim, cfg, rank=0, world=1):
    """apd_align_work: (ordered pairs, reference-loop cells, algorithmic bytes) of a rank's share."""
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    p, c, b = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
    _lib.check(_lib.lib().apd_align_work(offsets.ctypes.data_as(C.POINTER(C.c_uint64)), len(offsets) - 1, int(dim),
                                         C.byref(cfg), rank, world, C.byref(p), C.byref(c), C.byref(b)))
    return int(p.value), int(c.value), int(b.value)

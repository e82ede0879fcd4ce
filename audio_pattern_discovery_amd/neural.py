"""Host mirror of the forward half of src/neural.rs: AutoEncoder::{n_latent, predict}.
Training (take_step) and bincode (de)serialisation are outside the accelerated path."""
import ctypes as C

import numpy as np

from . import _lib


class Mat:
    """numerics.rs:171-174: flat row-major matrix + column count."""

    def __init__(self, flat, cols):
        self.flat = np.ascontiguousarray(flat, dtype=np.float32).ravel()
        self.cols = int(cols)

    def rows(self):
        return self.flat.size // self.cols


class AutoEncoder:
    """neural.rs:12-19.  Only the encoder half is used by predict()."""

    def __init__(self, w_encode, b_encode, w_decode=None, b_decode=None):
        self.w_encode = w_encode if isinstance(w_encode, Mat) else Mat(w_encode, np.asarray(w_encode).shape[1])
        self.b_encode = b_encode if isinstance(b_encode, Mat) else Mat(b_encode, np.asarray(b_encode).size)
        self.w_decode, self.b_decode = w_decode, b_decode

    def n_latent(self):                       # neural.rs:22-24
        return self.b_encode.cols

    def predict_frames(self, frames, ctx=None):
        """AutoEncoder::predict applied to every row of `frames` ([t][d_in]) -> [t][latent] (neural.rs:55-71)."""
        ctx = ctx or _lib.default_context()
        x = np.ascontiguousarray(frames, dtype=np.float32)
        d_in, latent = self.w_encode.rows(), self.n_latent()
        if x.ndim != 2 or x.shape[1] != d_in:
            raise AssertionError("self.cols == other.rows()")     # the reference's only assert (numerics.rs:306)
        out = np.empty((x.shape[0], latent), dtype=np.float32)
        f32p = C.POINTER(C.c_float)
        _lib.check(_lib.lib().apd_encode(ctx.handle, C.c_void_p(x.ctypes.data), x.shape[0], d_in,
                                         self.w_encode.flat.ctypes.data_as(f32p), self.b_encode.flat.ctypes.data_as(f32p),
                                         latent, 0, C.c_void_p(out.ctypes.data)), ctx.handle)
        return out

    def predict(self, x, ctx=None):           # neural.rs:55-71, x: Mat 1 x D
        flat = x.flat if isinstance(x, Mat) else np.asarray(x, np.float32)
        return Mat(self.predict_frames(np.asarray(flat, np.float32).reshape(1, -1), ctx).ravel(), self.n_latent())

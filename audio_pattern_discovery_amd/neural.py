"""Host mirror of the forward half of src/neural.rs: AutoEncoder::{n_latent, predict, from_file, save_file}.
Training (take_step) is outside the accelerated path.  from_file / save_file read and write the
reference's `output/encoder/auto_encoder.bin` (bincode 1.x default configuration: little-endian, usize and Vec
lengths as u64, f32 as 4 bytes; struct fields in declaration order -- neural.rs:13-19, numerics.rs:171-174), so the
weights of a real reference run can be fed to apd_encode."""
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

    @staticmethod
    def from_bytes(buf):
        """bincode::deserialize::<AutoEncoder> (neural.rs:30-36) through apd_autoencoder_parse / apd_autoencoder_copy:
        w_encode, w_decode, b_encode, b_decode, each a Mat { flat: Vec<f32>, cols: usize }."""
        buf = bytes(buf)
        view = _lib.AutoEncoderView()
        raw = C.create_string_buffer(buf, len(buf))
        if _lib.lib().apd_autoencoder_parse(raw, len(buf), C.byref(view)) != _lib.APD_OK:
            raise ValueError("not a bincode AutoEncoder (truncated, trailing bytes or inconsistent Mat shapes)")
        mats = []
        for mv in (view.w_encode, view.w_decode, view.b_encode, view.b_decode):
            flat = np.empty(mv.len, dtype=np.float32)
            _lib.check(_lib.lib().apd_autoencoder_copy(raw, C.byref(mv), flat.ctypes.data_as(C.POINTER(C.c_float))))
            mats.append(Mat(flat, mv.cols))
        return AutoEncoder(mats[0], mats[2], mats[1], mats[3])

    @staticmethod
    def from_file(file):                      # neural.rs:30-36
        with open(file, "rb") as fp:
            return AutoEncoder.from_bytes(fp.read())

    def to_bytes(self):                       # bincode::serialize (neural.rs:39-44) through apd_autoencoder_serialize
        mats = []
        for m in (self.w_encode, self.w_decode, self.b_encode, self.b_decode):
            if m is None:
                raise ValueError("decoder half missing: cannot serialise")
            mats.append(m if isinstance(m, Mat) else Mat(m, np.asarray(m).shape[-1]))
        latent, d_in = mats[0].cols, mats[0].rows()
        f32p = C.POINTER(C.c_float)
        ptrs = [m.flat.ctypes.data_as(f32p) for m in mats]
        n = C.c_uint64(0)
        _lib.check(_lib.lib().apd_autoencoder_serialize(None, None, None, None, d_in, latent, None, 0, C.byref(n)))
        out = C.create_string_buffer(n.value)
        if (mats[1].flat.size, mats[2].flat.size, mats[3].flat.size) != (d_in * latent, latent, d_in):
            raise ValueError("AutoEncoder matrices have inconsistent shapes")
        _lib.check(_lib.lib().apd_autoencoder_serialize(ptrs[0], ptrs[1], ptrs[2], ptrs[3], d_in, latent, out, n.value, C.byref(n)))
        return out.raw

    def save_file(self, file):                # neural.rs:39-44
        with open(file, "wb") as fp:
            fp.write(self.to_bytes())

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

"""ctypes binding of libapd_hip.so -- the C ABI declared in include/apd.h.

The library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  There is no CPU
fallback: a missing library raises ImportError-like errors here, a missing GPU makes
``Context()`` raise ``ApdError(APD_ERR_NO_DEVICE)``.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("APD_LIB") or os.path.join(_HERE, "libapd_hip.so")   # APD_LIB: tuning builds only

APD_OK = 0
APD_ERR_INVALID_ARG = -1
APD_ERR_NO_DEVICE = -2
APD_ERR_HIP = -3
APD_ERR_OOM = -4
APD_ERR_EMPTY_SEQUENCE = -5
APD_ERR_BAND_TOO_WIDE = -6
APD_ERR_INDEX = -7
APD_ERR_UNSUPPORTED = -8
APD_ERR_INCOMPLETE = -9
APD_ERR_COMM = -10
APD_COMM_ID_BYTES = 128


class AlignConfig(C.Structure):
    """apd_align_config: the Discovery fields the path reads (discovery.rs:17-20)."""
    _fields_ = [("warping_band_percentage", C.c_float), ("insertion_penalty", C.c_float),
                ("deletion_penalty", C.c_float), ("match_penalty", C.c_float)]


class AlignmentParamsC(C.Structure):
    """apd_alignment_params: AlignmentParams (alignments.rs:77-83)."""
    _fields_ = [("warping_band", C.c_uint64), ("insertion_penalty", C.c_float),
                ("deletion_penalty", C.c_float), ("match_penalty", C.c_float)]


class ClusterOp(C.Structure):
    """apd_cluster_op: ClusteringOperation (clustering.rs:19-25)."""
    _fields_ = [("merge_i", C.c_uint32), ("merge_j", C.c_uint32), ("into", C.c_uint32),
                ("distance", C.c_float), ("operation", C.c_uint32)]


class MatView(C.Structure):
    """apd_mat_view: where a Mat { flat, cols } (numerics.rs:171-174) lies in a bincode image."""
    _fields_ = [("offset", C.c_uint64), ("len", C.c_uint64), ("cols", C.c_uint64)]


class AutoEncoderView(C.Structure):
    """apd_autoencoder_view (neural.rs:13-19)."""
    _fields_ = [("w_encode", MatView), ("w_decode", MatView), ("b_encode", MatView), ("b_decode", MatView)]


class DiscoveryC(C.Structure):
    """apd_discovery (discovery.rs:7-26)."""
    _fields_ = [("dft_win", C.c_uint64), ("dft_step", C.c_uint64), ("ceps_filter", C.c_uint64), ("vat_moving", C.c_uint64),
                ("vat_percentile", C.c_float), ("vat_min_len", C.c_uint64), ("alignment_workers", C.c_uint64),
                ("clustering_percentile", C.c_float), ("warping_band_percentage", C.c_float), ("insertion_penalty", C.c_float),
                ("deletion_penalty", C.c_float), ("match_penalty", C.c_float), ("auto_encoder", C.c_uint64),
                ("learning_rate", C.c_float), ("epochs", C.c_uint64), ("epoch_drop", C.c_float), ("drop", C.c_float)]


class ApdError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = "apd status %d" % status
        try:
            msg = "%s (%s)" % (lib().apd_status_string(status).decode(), status)
        except Exception:
            pass
        super().__init__(msg + (": " + detail if detail else ""))


# every symbol include/apd.h declares: (name, restype, argtypes)
_f32p, _u64p, _u32p, _vp = C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_void_p
SYMBOLS = [
    ("apd_create", C.c_int, [C.c_int, C.POINTER(_vp)]),
    ("apd_destroy", C.c_int, [_vp]),
    ("apd_set_stream", C.c_int, [_vp, _vp]),
    ("apd_synchronize", C.c_int, [_vp]),
    ("apd_status_string", C.c_char_p, [C.c_int]),
    ("apd_last_error", C.c_char_p, [_vp]),
    ("apd_set_timing", C.c_int, [_vp, C.c_int]),
    ("apd_last_kernel_ms", C.c_float, [_vp]),
    ("apd_set_variant", C.c_int, [_vp, C.c_int]),
    ("apd_set_distance_mode", C.c_int, [_vp, C.c_int, C.c_float]),
    ("apd_selftest", C.c_int, [_vp]),
    ("apd_selftest_sqrt", C.c_int, [_vp, C.c_uint32, C.c_uint64, _u64p, _u32p, _u64p]),
    ("apd_set_fault_injection", C.c_int, [_vp, C.c_uint32]),
    ("apd_batch_refill", C.c_int, [_vp, _vp, _vp, C.c_int]),
    ("apd_batch_nonfinite", C.c_int, [_vp, _vp, C.POINTER(C.c_int)]),
    ("apd_comm_unique_id", C.c_int, [_vp]),
    ("apd_comm_create", C.c_int, [_vp, _vp, C.c_uint32, C.c_uint32, C.POINTER(_vp)]),
    ("apd_comm_destroy", C.c_int, [_vp]),
    ("apd_comm_count", C.c_int, [_vp, _u32p]),
    ("apd_comm_rank", C.c_int, [_vp, _u32p]),
    ("apd_align_all_sharded_async", C.c_int, [_vp, _vp, _vp, C.POINTER(AlignConfig), _vp]),
    ("apd_all_gather_async", C.c_int, [_vp, _vp, _vp, _vp, C.c_uint64]),
    ("apd_align_all_multi", C.c_int, [C.POINTER(C.c_int), C.c_uint32, _f32p, _u64p, C.c_uint32, C.c_uint32,
                                      C.POINTER(AlignConfig), _f32p, _u32p]),
    ("apd_dtw_all_pairs", C.c_int, [_f32p, _u64p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, _f32p]),
    ("apd_upgma", C.c_int, [_f32p, C.c_uint32, C.c_float, C.POINTER(ClusterOp), _u32p, _u32p, _u32p]),
    ("apd_multi_create", C.c_int, [C.POINTER(C.c_int), C.c_uint32, C.POINTER(_vp)]),
    ("apd_multi_destroy", C.c_int, [_vp]),
    ("apd_multi_size", C.c_uint32, [_vp]),
    ("apd_multi_ranks_seen", C.c_int, [_vp, _u32p]),
    ("apd_multi_collective", C.c_char_p, [_vp]),
    ("apd_multi_last_error", C.c_char_p, [_vp]),
    ("apd_multi_context", _vp, [_vp, C.c_uint32]),
    ("apd_multi_batch_create", C.c_int, [_vp, _vp, C.POINTER(_vp), _u64p, C.c_uint32, C.c_uint32, C.POINTER(_vp)]),
    ("apd_multi_batch_refill", C.c_int, [_vp, _vp, _vp, C.POINTER(_vp)]),
    ("apd_multi_batch_destroy", C.c_int, [_vp]),
    ("apd_multi_align_all_async", C.c_int, [_vp, _vp, C.POINTER(AlignConfig), _vp]),
    ("apd_multi_synchronize", C.c_int, [_vp]),
    ("apd_multi_result", _vp, [_vp]),
    ("apd_multi_align_all", C.c_int, [_vp, _vp, C.POINTER(AlignConfig), _f32p]),
    ("apd_device_alloc", C.c_int, [_vp, C.c_uint64, C.POINTER(_vp)]),
    ("apd_device_free", C.c_int, [_vp, _vp]),
    ("apd_copy_to_device", C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    ("apd_copy_to_host", C.c_int, [_vp, _vp, _vp, C.c_uint64]),
    ("apd_device_fill", C.c_int, [_vp, _vp, C.c_int, C.c_uint64]),
    ("apd_runtime_info", C.c_uint64, [C.c_char_p, C.c_uint64]),
    ("apd_discovery_alignment_params", C.c_int, [C.POINTER(AlignConfig), C.c_uint64, C.POINTER(AlignmentParamsC)]),
    ("apd_batch_create", C.c_int, [_vp, _vp, _u64p, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(_vp)]),
    ("apd_batch_destroy", C.c_int, [_vp]),
    ("apd_batch_len", C.c_uint32, [_vp]),
    ("apd_align_all", C.c_int, [_vp, _vp, C.POINTER(AlignConfig), _f32p]),
    ("apd_align_all_device_async", C.c_int, [_vp, _vp, C.POINTER(AlignConfig), _vp]),
    ("apd_tile_size", C.c_uint32, []),
    ("apd_num_tiles", C.c_uint64, [C.c_uint32]),
    ("apd_rank_tiles", C.c_uint64, [C.c_uint32, C.c_uint32, C.c_uint32]),
    ("apd_slab_floats", C.c_uint64, [C.c_uint32, C.c_uint32]),
    ("apd_align_tiles_async", C.c_int, [_vp, _vp, C.POINTER(AlignConfig), C.c_uint32, C.c_uint32, _vp]),
    ("apd_unpack_tiles_async", C.c_int, [_vp, _vp, C.c_uint32, _vp, _vp]),
    ("apd_length_order", C.c_int, [_u64p, C.c_uint32, _u32p]),
    ("apd_rank_tile_list", C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, _u32p, C.c_uint64, _u64p]),
    ("apd_unpack_tiles_host", C.c_int, [_u64p, C.c_uint32, C.c_uint32, _f32p, _f32p]),
    ("apd_align_work", C.c_int, [_u64p, C.c_uint32, C.c_uint32, C.POINTER(AlignConfig), C.c_uint32, C.c_uint32,
                                 _u64p, _u64p, _u64p]),
    ("apd_align_pair", C.c_int, [_vp, _f32p, C.c_uint64, _f32p, C.c_uint64, C.c_uint32,
                                 C.POINTER(AlignmentParamsC), _f32p]),
    ("apd_percentile", C.c_int, [_vp, _vp, C.c_uint64, C.c_float, C.c_int, _f32p]),
    ("apd_clustering", C.c_int, [_vp, _vp, C.c_int, C.c_uint32, C.c_float, C.POINTER(ClusterOp), _u32p, _u32p,
                                 _u32p, _f32p]),
    ("apd_cluster_sets", C.c_int, [C.POINTER(ClusterOp), C.c_uint32, _u32p, C.c_uint32, C.c_uint32, _u32p, _u32p,
                                   _u32p]),
    ("apd_encoder_create", C.c_int, [_vp, _f32p, _f32p, C.c_uint32, C.c_uint32, C.POINTER(_vp)]),
    ("apd_encoder_destroy", C.c_int, [_vp]),
    ("apd_encode_async", C.c_int, [_vp, _vp, _vp, C.c_uint64, _vp]),
    ("apd_cepstrum_plan_create", C.c_int, [_vp, _u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _u64p, _u32p, C.POINTER(_vp)]),
    ("apd_cepstrum_plan_destroy", C.c_int, [_vp]),
    ("apd_cepstrum_batch_async", C.c_int, [_vp, _vp, _vp, _vp]),
    ("apd_stream_busy", C.c_int, [_vp, C.POINTER(C.c_int)]),
    ("apd_debug_affinity_probe", C.c_int, [_vp, _vp, C.POINTER(C.c_int)]),
    ("apd_encode", C.c_int, [_vp, _vp, C.c_uint64, C.c_uint32, _f32p, _f32p, C.c_uint32, C.c_int, _vp]),
    ("apd_interesting_ranges", C.c_int, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_float, C.c_uint64, C.c_int, _u64p,
                                         C.c_uint64, _u64p]),
    ("apd_cepstrum_batch", C.c_int, [_vp, _vp, _u64p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _vp, _u64p,
                                     _u32p]),
    ("apd_cepstrum", C.c_int, [_vp, _vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, _vp, _u64p,
                               _u32p]),
    ("apd_autoencoder_parse", C.c_int, [_vp, C.c_uint64, C.POINTER(AutoEncoderView)]),
    ("apd_autoencoder_copy", C.c_int, [_vp, C.POINTER(MatView), _f32p]),
    ("apd_autoencoder_serialize", C.c_int, [_f32p, _f32p, _f32p, _f32p, C.c_uint32, C.c_uint32, _vp, C.c_uint64, _u64p]),
    ("apd_discovery_parse_toml", C.c_int, [C.c_char_p, C.POINTER(DiscoveryC)]),
    ("apd_dendrograms", C.c_int, [C.POINTER(ClusterOp), C.c_uint32, _u32p, C.c_uint32, C.POINTER(C.c_char_p), C.c_uint32, _vp,
                                  C.c_uint64, _u64p, _u32p, _u32p]),
]

_lib = None


def lib():
    """Loads libapd_hip.so; raises OSError loudly if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OSError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status, ctx_handle=None):
    if status != APD_OK:
        detail = ""
        if ctx_handle is not None and status in (APD_ERR_HIP, APD_ERR_INCOMPLETE, APD_ERR_COMM):
            detail = lib().apd_last_error(ctx_handle).decode()
        raise ApdError(status, detail)


def runtime_info():
    """apd_runtime_info: the HIP and RCCL libraries (path, version) the library's calls are bound to in this process."""
    n = int(lib().apd_runtime_info(None, 0))
    buf = C.create_string_buffer(n)
    lib().apd_runtime_info(buf, n)
    return buf.value.decode()


class Context:
    """apd_context: one GPU, one HIP stream."""

    def __init__(self, device=0, stream=None, borrowed=None):
        self._borrowed = borrowed is not None                # a context owned by an apd_multi handle: never destroyed here
        if self._borrowed:
            self.handle = _vp(int(borrowed))
            return
        self.handle = _vp()
        check(lib().apd_create(int(device), C.byref(self.handle)))
        if stream is not None:
            check(lib().apd_set_stream(self.handle, _vp(int(stream))), self.handle)

    def close(self):
        if getattr(self, "handle", None):
            if not self._borrowed:
                lib().apd_destroy(self.handle)
            self.handle = None

    def alloc(self, nbytes):
        """apd_device_alloc: a DeviceBuffer of nbytes in this context's HBM."""
        return DeviceBuffer(self, nbytes)

    def upload(self, array):
        """A DeviceBuffer holding a copy of the (contiguous) numpy array."""
        buf = DeviceBuffer(self, array.nbytes)
        buf.copy_from(array)
        return buf

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        check(lib().apd_synchronize(self.handle), self.handle)

    def set_timing(self, on=True):
        check(lib().apd_set_timing(self.handle, int(bool(on))))

    def last_kernel_ms(self):
        return float(lib().apd_last_kernel_ms(self.handle))

    def set_variant(self, v):
        check(lib().apd_set_variant(self.handle, int(v)))

    def set_distance_mode(self, mode, tau=0.0):
        """0 / "exact": difference form; 1 / "hybrid": norm expansion with exact recomputation below tau;
        2 / "strict": the reference's arithmetic operation for operation in every kernel family (bit-identical scores, ~2x slower);
        0 already is that on the band-form kernels."""
        mode = {"exact": 0, "hybrid": 1, "strict": 2}.get(mode, mode)
        check(lib().apd_set_distance_mode(self.handle, int(mode), float(tau)))

    def selftest(self):
        check(lib().apd_selftest(self.handle), self.handle)

    def selftest_sqrt(self, first_bits, count):
        """TEST HOOK: strict mode's square root against the correctly rounded one over `count` consecutive f32 bit patterns.
        Returns (mismatches, first mismatching pattern, [bare v_sqrt_f32 ulp-offset counts for <=-2, -1, 0, +1, >=+2])."""
        bad, first, hist = C.c_uint64(0), C.c_uint32(0), (C.c_uint64 * 5)()
        check(lib().apd_selftest_sqrt(self.handle, first_bits, count, C.byref(bad), C.byref(first), hist), self.handle)
        return bad.value, first.value, list(hist)

    def stream_busy(self):
        """TEST HOOK: True while work enqueued on the context's stream has not finished."""
        b = C.c_int(0)
        check(lib().apd_stream_busy(self.handle, C.byref(b)), self.handle)
        return bool(b.value)

    def set_fault_injection(self, drop_tiles):
        """TEST HOOK: the next alignment launches skip the last `drop_tiles` tiles of every kernel class."""
        check(lib().apd_set_fault_injection(self.handle, int(drop_tiles)))


class DeviceBuffer:
    """HBM owned through the C ABI (apd_device_alloc / _free / apd_copy_to_*): what the tests and bench.py keep features and
    matrices resident in -- no torch anywhere near the data path."""

    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, int(nbytes)
        self._p = _vp()
        check(lib().apd_device_alloc(ctx.handle, self.nbytes, C.byref(self._p)), ctx.handle)

    @property
    def ptr(self):
        return self._p.value or 0

    def at(self, byte_offset=0):
        return _vp(self.ptr + int(byte_offset))

    def copy_from(self, array, byte_offset=0):
        import numpy as np
        a = np.ascontiguousarray(array)
        if byte_offset + a.nbytes > self.nbytes:
            raise ValueError("copy past the end of the device buffer")
        check(lib().apd_copy_to_device(self.ctx.handle, self.at(byte_offset), _vp(a.ctypes.data), a.nbytes), self.ctx.handle)

    def to_numpy(self, dtype, count=None, byte_offset=0):
        import numpy as np
        dt = np.dtype(dtype)
        count = (self.nbytes - byte_offset) // dt.itemsize if count is None else int(count)
        out = np.empty(count, dtype=dt)
        if byte_offset + out.nbytes > self.nbytes:
            raise ValueError("copy past the end of the device buffer")
        check(lib().apd_copy_to_host(self.ctx.handle, _vp(out.ctypes.data), self.at(byte_offset), out.nbytes), self.ctx.handle)
        return out

    def fill(self, byte_value):
        check(lib().apd_device_fill(self.ctx.handle, self._p, int(byte_value), self.nbytes), self.ctx.handle)

    def free(self):
        if getattr(self, "_p", None) and self._p.value and getattr(self.ctx, "handle", None):
            lib().apd_device_free(self.ctx.handle, self._p)
        self._p = _vp()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx

"""Pair-tile sharding of the all-pairs matrix (host side of include/apd.h's sharded entry points).

The reference splits the N x N matrix into contiguous row blocks, one per OS thread
(alignments.rs:33-37).  Here the unit is a 16 x 16 tile of UNORDERED pairs (a < b): one fused sweep
yields both out[a][b] and out[b][a], so row slabs would split a pair's two results across ranks.
Tiles are cut over the sequences in LENGTH ORDER (length_order: longest first), so that the sequences of a
tile have like lengths and one kernel geometry fits them all.  Tile g (row-major over the upper triangle, diagonal tiles included) belongs to rank g % world; a rank
packs its tiles, in order, into one slab [tiles][2][16][16]; all slabs have the same size so that ONE
all-gather assembles them, and apd_unpack_tiles_* scatters them into the N x N matrix.
"""
import ctypes as C

import numpy as np

from . import _lib


def tile_size():
    return int(_lib.lib().apd_tile_size())


def num_tiles(n_seq):
    return int(_lib.lib().apd_num_tiles(n_seq))


def slab_floats(n_seq, world):
    return int(_lib.lib().apd_slab_floats(n_seq, world))


def length_order(offsets):
    """order[p] = caller's index of the sequence at position p of the tiling order."""
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    order = np.zeros(len(off) - 1, dtype=np.uint32)
    _lib.check(_lib.lib().apd_length_order(off.ctypes.data_as(C.POINTER(C.c_uint64)), len(off) - 1,
                                           order.ctypes.data_as(C.POINTER(C.c_uint32))))
    return order


def rank_tiles(n_seq, rank, world):
    """(n_tiles, 2) array of (tile_a, tile_b) owned by `rank`, in slab order; tile t covers positions
    16 t .. 16 t + 15 of length_order()."""
    n = C.c_uint64(0)
    _lib.check(_lib.lib().apd_rank_tile_list(n_seq, rank, world, None, 0, C.byref(n)))
    out = np.zeros((n.value, 2), dtype=np.uint32)
    if n.value:
        _lib.check(_lib.lib().apd_rank_tile_list(n_seq, rank, world, out.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                 n.value, C.byref(n)))
    return out


def unpack_host(offsets, world, gathered):
    """Scatter `world` gathered slabs (host memory, rank order) into the (n_seq, n_seq) matrix."""
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    n_seq = len(off) - 1
    g = np.ascontiguousarray(gathered, dtype=np.float32).ravel()
    if g.size != world * slab_floats(n_seq, world):
        raise ValueError("gathered must hold world * slab_floats values")
    out = np.empty((n_seq, n_seq), dtype=np.float32)
    _lib.check(_lib.lib().apd_unpack_tiles_host(off.ctypes.data_as(C.POINTER(C.c_uint64)), n_seq, world, g.ctypes.data_as(C.POINTER(C.c_float)),
                                                out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


class Comm:
    """apd_comm: this rank's end of the library-owned RCCL communicator (one process per GPU).  Rank 0 makes the id with
    ``Comm.unique_id()`` and hands its 128 bytes to the other ranks over any host channel."""

    def __init__(self, ctx, id_bytes, rank, world):
        if len(id_bytes) != _lib.APD_COMM_ID_BYTES:
            raise ValueError("communicator ids are %d bytes" % _lib.APD_COMM_ID_BYTES)
        self.ctx = ctx
        self.handle = C.c_void_p()
        buf = C.create_string_buffer(bytes(id_bytes), _lib.APD_COMM_ID_BYTES)
        _lib.check(_lib.lib().apd_comm_create(ctx.handle, buf, int(rank), int(world), C.byref(self.handle)), ctx.handle)

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(_lib.APD_COMM_ID_BYTES)
        _lib.check(_lib.lib().apd_comm_unique_id(buf))
        return buf.raw

    def count(self):
        """ncclCommCount: the ranks RCCL actually sees."""
        n = C.c_uint32(0)
        _lib.check(_lib.lib().apd_comm_count(self.handle, C.byref(n)), self.ctx.handle)
        return int(n.value)

    def rank(self):
        r = C.c_uint32(0)
        _lib.check(_lib.lib().apd_comm_rank(self.handle, C.byref(r)), self.ctx.handle)
        return int(r.value)

    def align_all_sharded_async(self, batch_handle, cfg, d_out_ptr):
        """This rank's pair tiles + ONE ncclAllGather + unpack, on the context's stream; d_out_ptr: device address of n*n floats."""
        _lib.check(_lib.lib().apd_align_all_sharded_async(self.ctx.handle, self.handle, batch_handle, C.byref(cfg), C.c_void_p(int(d_out_ptr))),
                   self.ctx.handle)

    def close(self):
        if getattr(self, "handle", None):
            _lib.lib().apd_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Multi:
    """apd_multi: ONE process driving len(devices) GPUs through a persistent handle -- contexts, the RCCL communicators of
    ncclCommInitAll, one worker thread per device and the gather workspaces are made once (AlignmentWorkers over N GPUs,
    reference alignments.rs:11-67)."""

    def __init__(self, devices):
        self.devices = [int(d) for d in devices]
        self.handle = C.c_void_p()
        devs = (C.c_int * len(self.devices))(*self.devices)
        _lib.check(_lib.lib().apd_multi_create(devs, len(self.devices), C.byref(self.handle)))
        self.contexts = [_lib.Context(borrowed=_lib.lib().apd_multi_context(self.handle, i)) for i in range(len(self.devices))]

    def _check(self, status):
        if status != _lib.APD_OK:
            raise _lib.ApdError(status, _lib.lib().apd_multi_last_error(self.handle).decode())

    def ranks_seen(self):
        n = C.c_uint32(0)
        self._check(_lib.lib().apd_multi_ranks_seen(self.handle, C.byref(n)))
        return int(n.value)

    def collective(self):
        return _lib.lib().apd_multi_collective(self.handle).decode()

    def batch(self, offsets, dim, frames=None, d_frames=None):
        return MultiBatch(self, offsets, dim, frames, d_frames)

    def align_all_async(self, batch, cfg, d_out_ptr=None):
        self._check(_lib.lib().apd_multi_align_all_async(self.handle, batch.handle, C.byref(cfg), C.c_void_p(int(d_out_ptr)) if d_out_ptr else None))

    def align_all(self, batch, cfg):
        out = np.empty((batch.n_seq, batch.n_seq), dtype=np.float32)
        self._check(_lib.lib().apd_multi_align_all(self.handle, batch.handle, C.byref(cfg), out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def result_ptr(self):
        return _lib.lib().apd_multi_result(self.handle) or 0

    def synchronize(self):
        self._check(_lib.lib().apd_multi_synchronize(self.handle))

    def close(self):
        if getattr(self, "handle", None):
            _lib.lib().apd_multi_destroy(self.handle)
            self.handle = None
            for c in self.contexts:
                c.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiBatch:
    """apd_multi_batch: the corpus resident on every device of a Multi."""

    def __init__(self, multi, offsets, dim, frames=None, d_frames=None):
        self.multi = multi
        self.offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.n_seq = len(self.offsets) - 1
        self.handle = C.c_void_p()
        host, dev = self._args(frames, d_frames)
        multi._check(_lib.lib().apd_multi_batch_create(multi.handle, host, dev, self.offsets.ctypes.data_as(C.POINTER(C.c_uint64)), self.n_seq,
                                                       int(dim), C.byref(self.handle)))

    def _args(self, frames, d_frames):
        if d_frames is not None:
            self._dev = (C.c_void_p * len(d_frames))(*[int(p) for p in d_frames])
            return None, self._dev
        self._host = np.ascontiguousarray(frames, dtype=np.float32)
        return C.c_void_p(self._host.ctypes.data), None

    def refill(self, frames=None, d_frames=None):
        host, dev = self._args(frames, d_frames)
        self.multi._check(_lib.lib().apd_multi_batch_refill(self.multi.handle, self.handle, host, dev))

    def close(self):
        if getattr(self, "handle", None) and getattr(self.multi, "handle", None):
            _lib.lib().apd_multi_batch_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def align_all_multi(devices, frames, offsets, dim, cfg):
    """apd_align_all_multi: ONE process, len(devices) GPUs (ncclCommInitAll).  Returns ((n, n) matrix, ranks RCCL saw)."""
    off = np.ascontiguousarray(offsets, dtype=np.uint64)
    fr = np.ascontiguousarray(frames, dtype=np.float32)
    n = len(off) - 1
    out = np.empty((n, n), dtype=np.float32)
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    seen = C.c_uint32(0)
    _lib.check(_lib.lib().apd_align_all_multi(devs, len(devices), fr.ctypes.data_as(C.POINTER(C.c_float)), off.ctypes.data_as(C.POINTER(C.c_uint64)),
                                              n, int(dim), C.byref(cfg), out.ctypes.data_as(C.POINTER(C.c_float)), C.byref(seen)))
    return out, int(seen.value)

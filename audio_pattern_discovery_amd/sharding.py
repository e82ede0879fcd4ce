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

"""Host mirror of src/clustering.rs over the C ABI of include/apd.h."""
import ctypes as C
import enum
from dataclasses import dataclass

import numpy as np

from . import _lib


class Merge(enum.IntEnum):
    """clustering.rs:8-13"""
    Sequence2Sequence = 0
    Sequence2Cluster = 1
    Cluster2Sequence = 2
    Cluster2Cluster = 3


@dataclass
class ClusteringOperation:
    """clustering.rs:19-25 (`distance` is private in the reference; kept readable here for tests)."""
    merge_i: int
    merge_j: int
    into: int
    distance: float
    operation: Merge


class AgglomerativeClustering:
    """clustering.rs:27-210: the two associated functions the pipeline calls (main.rs:196-203)."""

    @staticmethod
    def clustering(distances, n_instances, perc, ctx=None, return_threshold=False):
        """clustering.rs:81-110 -> (Vec<ClusteringOperation>, set of root ids)."""
        ctx = ctx or _lib.default_context()
        d = np.ascontiguousarray(distances, dtype=np.float32).ravel()
        if d.size != n_instances * n_instances:
            raise ValueError("distances must hold n_instances^2 values")
        ops = (_lib.ClusterOp * max(n_instances, 1))()
        roots = np.zeros(max(n_instances, 1), dtype=np.uint32)
        n_ops, n_roots, thr = C.c_uint32(0), C.c_uint32(0), C.c_float(0)
        _lib.check(_lib.lib().apd_clustering(ctx.handle, C.c_void_p(d.ctypes.data), 0, n_instances, float(perc), ops,
                                             C.byref(n_ops), roots.ctypes.data_as(C.POINTER(C.c_uint32)),
                                             C.byref(n_roots), C.byref(thr)), ctx.handle)
        out = [ClusteringOperation(o.merge_i, o.merge_j, o.into, o.distance, Merge(o.operation))
               for o in ops[:n_ops.value]]
        ids = set(int(r) for r in roots[:n_roots.value])
        return (out, ids, float(thr.value)) if return_threshold else (out, ids)

    @staticmethod
    def cluster_sets(operations, cluster_ids, n_instances):
        """clustering.rs:40-76.  Roots are visited in ascending order (the reference iterates a HashSet)."""
        ops = (_lib.ClusterOp * max(len(operations), 1))()
        for t, o in enumerate(operations):
            ops[t] = _lib.ClusterOp(o.merge_i, o.merge_j, o.into, o.distance, int(o.operation))
        roots = np.array(sorted(cluster_ids), dtype=np.uint32)
        members = np.zeros(n_instances + len(operations) + 2, dtype=np.uint32)
        set_off = np.zeros(len(roots) + 2, dtype=np.uint32)
        n_sets = C.c_uint32(0)
        _lib.check(_lib.lib().apd_cluster_sets(ops, len(operations), roots.ctypes.data_as(C.POINTER(C.c_uint32)),
                                               len(roots), n_instances, members.ctypes.data_as(C.POINTER(C.c_uint32)),
                                               set_off.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(n_sets)))
        return [members[set_off[s]:set_off[s + 1]].tolist() for s in range(n_sets.value)]


def percentile(x, perc, ctx=None):
    """numerics.rs:125-133 on the GPU (apd_percentile)."""
    ctx = ctx or _lib.default_context()
    a = np.ascontiguousarray(x, dtype=np.float32).ravel()
    v = C.c_float(0)
    _lib.check(_lib.lib().apd_percentile(ctx.handle, C.c_void_p(a.ctypes.data), a.size, float(perc), 0, C.byref(v)),
               ctx.handle)
    return float(v.value)


def dendrograms(operations, clusters, labels):
    """The bracket strings reporting.rs:135-169 builds for TikZ-qtree (apd_dendrograms), one per root that was merged at
    least once: leaf i is rendered as labels[i] (the reference puts an \\includegraphics reference there, reporting.rs:211-221),
    node k as "[.k [<left> <right> ] ]".  Roots never merged are skipped, as the reference does (reporting.rs:200).
    Lets a run be diffed against the reference's output/ (SURVEY.md 8(f) item 4)."""
    ops = (_lib.ClusterOp * max(len(operations), 1))()
    for t, o in enumerate(operations):
        ops[t] = _lib.ClusterOp(o.merge_i, o.merge_j, o.into, o.distance, int(o.operation))
    roots = np.array(sorted(clusters), dtype=np.uint32)
    lab = (C.c_char_p * max(len(labels), 1))(*[str(v).encode() for v in labels])
    which = np.zeros(len(roots) + 1, dtype=np.uint32)
    n_bytes, n_strings = C.c_uint64(0), C.c_uint32(0)
    u32p = C.POINTER(C.c_uint32)
    L = _lib.lib()
    _lib.check(L.apd_dendrograms(ops, len(operations), roots.ctypes.data_as(u32p), len(roots), lab, len(labels), None, 0,
                                 C.byref(n_bytes), which.ctypes.data_as(u32p), C.byref(n_strings)))
    buf = C.create_string_buffer(n_bytes.value + 1)
    _lib.check(L.apd_dendrograms(ops, len(operations), roots.ctypes.data_as(u32p), len(roots), lab, len(labels), buf, n_bytes.value,
                                 C.byref(n_bytes), which.ctypes.data_as(u32p), C.byref(n_strings)))
    parts = buf.raw[:n_bytes.value].split(b"\0")[:n_strings.value]
    return {int(roots[which[i]]): parts[i].decode() for i in range(n_strings.value)}

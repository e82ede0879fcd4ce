"""The two one-call entry points SURVEY.md section 8(b) spells out (`apd_dtw_all_pairs`, `apd_upgma`): host buffers in, host buffers
out, the library owning contexts, device buffers and communicators for the duration of the call -- what the Rust shim's
`AlignmentWorkers::align_all` (src/alignments.rs:31-67) and `AgglomerativeClustering::clustering` (src/clustering.rs:81-110) would
bind first.  Through the C ABI, against the CPU oracle (DTW: 1e-4 relative, +INF / zero pattern identical; clustering: merge
sequence identical, linkages within 1e-5) and against the context-based entry points (bitwise)."""
import ctypes as C

import numpy as np
import pytest

from audio_pattern_discovery_amd import synth

pytestmark = pytest.mark.gpu


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


@pytest.mark.parametrize("n_seq,length,dim,band", [(24, 40, 13, 0.0625), (17, 30, 8, 1.0), (3, 5, 2, 0.5)])
def test_dtw_all_pairs_and_upgma_in_one_call_each(apd, oracle, n_seq, length, dim, band):
    frames, offsets = synth.make_sequences(n_seq, length, dim, seed=n_seq, jitter=max(length // 5, 1))
    L = apd.lib()
    out = np.full((n_seq, n_seq), -1.0, dtype=np.float32)
    apd.check(L.apd_dtw_all_pairs(_p(frames, C.c_float), _p(offsets, C.c_uint64), n_seq, dim, band, 1.0, 1.0, 1.0, 1, _p(out, C.c_float)))
    want = oracle.align_all(frames, offsets, band, workers=8).reshape(n_seq, n_seq)
    assert np.array_equal(np.isinf(out), np.isinf(want)) and np.array_equal(out == 0.0, want == 0.0)
    fin = np.isfinite(want)
    assert np.max(np.abs(out[fin] - want[fin]) / np.maximum(np.abs(want[fin]), 1e-30), initial=0.0) <= 1e-4
    assert np.all(np.diag(out) == 0.0)

    # the same matrix through the context-based calls: bit for bit
    ctx = apd.Context(0)
    try:
        from audio_pattern_discovery_amd.alignments import Batch
        cfg = apd.AlignConfig(band, 1.0, 1.0, 1.0)
        batch = Batch(ctx, frames, offsets, dim)
        ref = np.zeros((n_seq, n_seq), dtype=np.float32)
        apd.check(L.apd_align_all(ctx.handle, batch.handle, C.byref(cfg), _p(ref, C.c_float)), ctx.handle)
        assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    finally:
        ctx.close()

    ops = (apd.ClusterOp * n_seq)()
    roots = np.zeros(n_seq, dtype=np.uint32)
    n_ops, n_roots = C.c_uint32(0), C.c_uint32(0)
    apd.check(L.apd_upgma(_p(want.astype(np.float32), C.c_float), n_seq, 0.3, ops, C.byref(n_ops), _p(roots, C.c_uint32), C.byref(n_roots)))
    want_ops, want_roots, _ = oracle.clustering(want, n_seq, 0.3)
    got = [(o.merge_i, o.merge_j, o.into) for o in ops[:n_ops.value]]
    assert got == [(o["merge_i"], o["merge_j"], o["into"]) for o in want_ops]
    for g, w in zip(ops[:n_ops.value], want_ops):
        assert g.distance == w["distance"] or abs(g.distance - w["distance"]) <= 1e-5 * abs(w["distance"])
    assert sorted(roots[:n_roots.value].tolist()) == want_roots


def test_one_call_entry_points_refuse_bad_arguments(apd):
    L = apd.lib()
    frames = np.zeros((4, 2), dtype=np.float32)
    offsets = np.array([0, 2, 2, 4], dtype=np.uint64)           # a zero-length sequence: the reference underflows usize (alignments.rs:120)
    out = np.zeros((3, 3), dtype=np.float32)
    rc = L.apd_dtw_all_pairs(_p(frames, C.c_float), _p(offsets, C.c_uint64), 3, 2, 1.0, 1.0, 1.0, 1.0, 1, _p(out, C.c_float))
    assert rc != 0 and b"zero-length" in L.apd_status_string(rc)
    n_ops, n_roots = C.c_uint32(7), C.c_uint32(7)
    assert L.apd_upgma(None, 3, 0.5, None, C.byref(n_ops), None, C.byref(n_roots)) != 0

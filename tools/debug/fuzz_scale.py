"""Hybrid-distance robustness sweep (run on the GPU box): feature scales, common offsets and mixed magnitudes."""
import sys, numpy as np
sys.path.insert(0, ".")
from audio_pattern_discovery_amd import synth, _lib
from audio_pattern_discovery_amd.alignments import AlignmentWorkers, NDSequence
from audio_pattern_discovery_amd.discovery import Discovery
from oracle import binding as oracle

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
ctx = _lib.Context(0)
worst = 0.0
for case in range(120):
    dim = int(rng.choice([10, 13, 20, 26]))
    n_seq, length = int(rng.integers(4, 30)), int(rng.choice([30, 120, 400]))
    pct = float(rng.choice([0.0625, 0.25, 1.0]))
    frames, offsets = synth.make_sequences(n_seq, length, dim, seed=int(rng.integers(1 << 30)), jitter=length // 8, copies=0.5)
    kind = int(rng.integers(0, 5))
    if kind == 0: frames = frames * np.float32(1e3)
    elif kind == 1: frames = frames * np.float32(1e-3)
    elif kind == 2: frames = frames + np.float32(100.0)
    elif kind == 3: frames = frames * (10.0 ** rng.integers(-2, 3, size=(len(frames), 1))).astype(np.float32)   # per-frame magnitudes
    else: frames[rng.random(len(frames)) < 0.1] = 0.0                                                            # all-zero frames
    frames = np.ascontiguousarray(frames, dtype=np.float32)
    want = oracle.align_all(frames, offsets, pct, workers=16)
    seqs = [NDSequence(s) for s in synth.split(frames, offsets)]
    got = AlignmentWorkers.new(seqs, ctx).align_all(Discovery(warping_band_percentage=pct)).reshape(n_seq, n_seq)
    fin = np.isfinite(want)
    assert np.array_equal(fin, np.isfinite(got)), (case, kind)
    zero = fin & (want == 0)
    assert np.all(got[zero] == 0), (case, kind, "zeros")
    nz = fin & ~zero
    rel = float((np.abs(got[nz] - want[nz]) / np.abs(want[nz])).max()) if nz.any() else 0.0
    worst = max(worst, rel)
    if rel > 1e-4:
        print("FAIL", case, dict(dim=dim, n_seq=n_seq, length=length, pct=pct, kind=kind), rel)
print("worst rel", worst)

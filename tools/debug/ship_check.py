"""Debug: full-band ragged batch, GPU vs oracle on every pair of a subset, and run-to-run determinism."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from audio_pattern_discovery_amd import synth, _lib
from audio_pattern_discovery_amd.alignments import AlignmentWorkers, NDSequence
from audio_pattern_discovery_amd.discovery import Discovery
from oracle import binding as oracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 96
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 10
frames, offsets = synth.make_sequences(n, 525, dim, seed=77, jitter=375)
ctx = _lib.Context(0)
seqs = [NDSequence(s) for s in synth.split(frames, offsets)]
cfg = Discovery(warping_band_percentage=1.0)
outs = []
for rep in range(3):
    w = AlignmentWorkers.new(seqs, ctx)
    outs.append(w.align_all(cfg).reshape(n, n).copy())
print("deterministic:", np.array_equal(outs[0], outs[1]), np.array_equal(outs[0], outs[2]))
want = oracle.align_all(frames, offsets, 1.0, workers=16)
rel = np.abs(outs[0] - want) / np.maximum(np.abs(want), 1e-30)
np.fill_diagonal(rel, 0)
print("max rel", rel.max())
bad = np.argwhere(rel > 1e-5)
lens = np.diff(offsets.astype(np.int64))
for a, b in bad[:20]:
    print(a, b, lens[a], lens[b], outs[0][a, b], want[a, b], rel[a, b])
print("n bad", len(bad))

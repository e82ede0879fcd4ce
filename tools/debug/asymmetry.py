import sys, numpy as np
sys.path.insert(0, "/root/repo")
from audio_pattern_discovery_amd import _lib, synth
from audio_pattern_discovery_amd.alignments import AlignmentWorkers, NDSequence
from audio_pattern_discovery_amd.discovery import Discovery
ctx = _lib.Context(0)
for (n, L, pct) in [(1024, 512, 0.0625), (2048, 1024, 0.0625)]:
    frames, offsets = synth.make_sequences(n, L, 13, seed=77)
    w = AlignmentWorkers.new([NDSequence(s) for s in synth.split(frames, offsets)], ctx)
    got = w.align_all(Discovery(warping_band_percentage=pct)).reshape(n, n)
    iu = np.triu_indices(n, 1)
    a, b = got[iu], got.T[iu]
    print(n, L, "pairs", len(a), "asymmetric (bitwise)", int((a.view(np.uint32) != b.view(np.uint32)).sum()), "rel diff > 1e-6:", int((np.abs(a - b) > 1e-6 * np.abs(a)).sum()))

"""GPU box: replay fuzz case (seed, case) of tools/debug/fuzz.py on the default geometry choice, print the tile plan and the bad pairs.
usage: [APD_LIB=...] python tools/debug/replay_pair.py <seed> <case>"""
import os, sys, numpy as np
sys.path.insert(0, ".")
os.environ["APD_DEBUG_PLAN"] = "1"
from audio_pattern_discovery_amd import synth, _lib
from audio_pattern_discovery_amd.alignments import AlignmentWorkers, NDSequence
from audio_pattern_discovery_amd.discovery import Discovery
from oracle import binding as oracle
seed, target = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
for case in range(target + 1):
    dim = int(rng.choice([1, 2, 3, 5, 8, 9, 10, 12, 13, 15, 16, 19, 20, 23, 26, 27, 33]))
    n_seq = int(rng.integers(2, 50))
    length = int(rng.choice([3, 8, 20, 60, 150, 400, 900, 1700, 2600]))
    jitter = int(rng.integers(0, max(length - 1, 1)))
    pct = float(rng.choice([0.0, 0.01, 0.0625, 0.1, 0.25, 0.5, 0.9, 1.0, 1.5]))
    integer = bool(rng.random() < 0.4)
    pk = rng.random()
    pens = (1.0, 1.0, 1.0) if pk < 0.5 else ((0.7, 0.7, 0.7) if pk < 0.65 else tuple(float(v) for v in rng.choice([0.25, 0.5, 0.8, 1.0, 1.2, 2.0], 3)))
    mk = rng.random()
    if length >= 400:
        n_seq = min(n_seq, 14 if length < 1700 else 7)
    sseed = int(rng.integers(1 << 30))
    copies = float(rng.choice([0.0, 0.25, 0.6]))
print(dict(dim=dim, n_seq=n_seq, length=length, jitter=jitter, pct=pct, integer=integer, pens=pens, sseed=sseed, copies=copies))
frames, offsets = synth.make_sequences(n_seq, length, dim, seed=sseed, integer=integer, jitter=jitter, copies=copies)
want = oracle.align_all(frames, offsets, pct, *pens, workers=16)
ctx = _lib.Context(0)
seqs = [NDSequence(s) for s in synth.split(frames, offsets)]
lens = np.diff(offsets.astype(np.int64))
m = np.isfinite(want) & (want != 0)
for md in ("hybrid", "exact", "strict"):
    ctx.set_distance_mode(md)
    got = AlignmentWorkers.new(seqs, ctx).align_all(Discovery(warping_band_percentage=pct, insertion_penalty=pens[0], deletion_penalty=pens[1],
                                                              match_penalty=pens[2])).reshape(n_seq, n_seq)
    bad = np.argwhere(m & (np.abs(got - want) > 1e-4 * np.abs(want)))
    print(md, "bad", [(int(a), int(b), int(lens[a]), int(lens[b]), float(got[a, b]), float(want[a, b])) for a, b in bad[:6]], flush=True)

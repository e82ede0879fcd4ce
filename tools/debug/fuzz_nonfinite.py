"""Randomised sweep of batches with NaN / +-INF features against the CPU oracle (run on the GPU box): such a batch must take
the literal kernel -- chosen ON THE DEVICE since round 3 (the fast kernels return on the repack kernel's flag, the persistent
fallback does the work) -- and every score must equal the oracle bit for bit, NaN pattern included (alignments.rs:153-159: a NaN
compares false and takes MATCH).  Interleaved with finite batches on the same context, whose results must stay within 1e-4.
usage: python tools/debug/fuzz_nonfinite.py [n_cases] [seed]"""
import sys, time, numpy as np
sys.path.insert(0, ".")
from audio_pattern_discovery_amd import synth, _lib
from audio_pattern_discovery_amd.alignments import AlignmentWorkers, NDSequence
from audio_pattern_discovery_amd.discovery import Discovery
from oracle import binding as oracle

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = _lib.Context(0)
fails, t0 = 0, time.time()
for case in range(n_cases):
    dim = int(rng.choice([1, 3, 8, 10, 13, 16, 20, 26, 29]))
    n_seq = int(rng.integers(2, 40))
    length = int(rng.choice([3, 12, 60, 150, 400, 900]))
    if length >= 400:
        n_seq = min(n_seq, 10)
    pct = float(rng.choice([0.0, 0.0625, 0.1, 0.5, 1.0]))
    pens = (1.0, 1.0, 1.0) if rng.random() < 0.6 else tuple(float(v) for v in rng.choice([0.5, 1.0, 1.5], 3))
    frames, offsets = synth.make_sequences(n_seq, length, dim, seed=int(rng.integers(1 << 30)), jitter=int(rng.integers(0, max(length // 2, 1))),
                                           integer=bool(rng.random() < 0.3))
    poisoned = rng.random() < 0.7
    if poisoned:
        for _ in range(int(rng.integers(1, 6))):
            frames[int(rng.integers(0, len(frames))), int(rng.integers(0, dim))] = rng.choice([np.nan, np.inf, -np.inf])
    want = oracle.align_all(frames, offsets, pct, *pens, workers=16)
    ctx.set_distance_mode("hybrid" if rng.random() < 0.7 else "exact")
    got = AlignmentWorkers.new([NDSequence(s) for s in synth.split(frames, offsets)], ctx).align_all(
        Discovery(warping_band_percentage=pct, insertion_penalty=pens[0], deletion_penalty=pens[1], match_penalty=pens[2])).reshape(n_seq, n_seq)
    if poisoned:
        nan = np.isnan(want)
        ok = np.array_equal(nan, np.isnan(got)) and np.array_equal(got[~nan].view(np.uint32), want[~nan].view(np.uint32))
    else:
        fin = np.isfinite(want) & (want != 0)
        ok = np.array_equal(np.isfinite(want), np.isfinite(got)) and np.all(got[np.isfinite(want) & (want == 0)] == 0) and \
            (not fin.any() or float((np.abs(got[fin] - want[fin]) / np.abs(want[fin])).max()) <= 1e-4)
    if not ok:
        fails += 1
        print("FAIL case", case, dict(dim=dim, n_seq=n_seq, length=length, pct=pct, pens=pens, poisoned=poisoned), flush=True)
    if case % 50 == 49:
        print("case", case + 1, "fails", fails, "%.0fs" % (time.time() - t0), flush=True)
print("done: cases", n_cases, "fails", fails)
sys.exit(1 if fails else 0)

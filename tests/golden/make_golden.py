"""Generates tests/golden/*.npz with the CPU oracle (oracle/apd_oracle.c).

The reference has no tests or fixtures and cannot be built here (Rust, no toolchain), so
these vectors are the ORACLE's outputs on seeded inputs -- they pin the oracle against
regressions and travel to the GPU box; they are not reference outputs (parity unpinned).

Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from audio_pattern_discovery_amd import synth  # noqa: E402
from oracle import binding as ob  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = [
    # name, n_seq, nominal_len, dim, seed, integer, jitter, pct, ins, del, match, perc
    ("gauss_full", 8, 32, 13, 11, False, 8, 1.0, 1.0, 1.0, 1.0, 0.3),
    ("gauss_band", 8, 32, 13, 12, False, 8, 0.0625, 1.0, 1.0, 1.0, 0.3),
    ("gauss_pen", 8, 32, 13, 13, False, 8, 0.25, 0.5, 1.5, 0.75, 0.5),
    ("int_ties_full", 8, 32, 13, 14, True, 8, 1.0, 1.0, 1.0, 1.0, 0.3),
    ("int_ties_band", 8, 32, 13, 15, True, 8, 0.0625, 1.0, 1.0, 1.0, 0.3),
    ("int_ties_pen", 8, 32, 13, 16, True, 8, 0.25, 0.25, 0.75, 1.0, 0.5),
    ("latent8_band", 10, 48, 8, 17, False, 3, 0.0625, 1.0, 1.0, 1.0, 0.2),
    ("wide_gap", 6, 40, 13, 18, False, 30, 0.0625, 1.0, 1.0, 1.0, 0.4),
    ("dim1_int", 7, 20, 1, 19, True, 6, 0.1, 1.0, 1.0, 1.0, 0.4),
]


def main():
    for (name, n, ln, dim, seed, integer, jit, pct, ins, dele, mat, perc) in CASES:
        frames, offsets = synth.make_sequences(n, ln, dim, seed=seed, integer=integer, jitter=jit)
        dist = ob.align_all(frames, offsets, pct, ins, dele, mat, workers=4)
        ops, roots, thr = ob.clustering(dist, n, perc)
        sets = ob.cluster_sets(ops, roots, n)
        np.savez_compressed(
            os.path.join(HERE, name + ".npz"), frames=frames, offsets=offsets,
            params=np.array([pct, ins, dele, mat, perc], dtype=np.float32), dist=dist,
            op_ij=np.array([[o["merge_i"], o["merge_j"], o["into"]] for o in ops], dtype=np.int64).reshape(-1, 3),
            op_dist=np.array([o["distance"] for o in ops], dtype=np.float32),
            op_kind=np.array([ob.MERGE_NAMES.index(o["operation"]) for o in ops], dtype=np.int64),
            roots=np.array(roots, dtype=np.int64), threshold=np.float32(thr),
            set_members=np.array([m for s in sets for m in s], dtype=np.int64),
            set_sizes=np.array([len(s) for s in sets], dtype=np.int64))
        print(name, dist.shape, len(ops), "ops", "thr", thr)
    # companions: encoder + cepstrum
    rng = np.random.default_rng(21)
    x = rng.standard_normal((40, 13)).astype(np.float32)
    w = ((rng.random((13, 8)) - 0.5) / 8).astype(np.float32)
    b = ((rng.random(8) - 0.5) / 8).astype(np.float32)
    audio = synth.make_audio(256 + 128 * 12 + 5, seed=22)
    np.savez_compressed(os.path.join(HERE, "companions.npz"), x=x, w=w, b=b, enc=ob.encode(x, w, b),
                        audio=audio, ceps13=ob.cepstrum(audio, 256, 128, 18),
                        ceps26=ob.cepstrum(audio, 256, 128, 32))
    print("companions", ob.cepstrum(audio, 256, 128, 18).shape, ob.cepstrum(audio, 256, 128, 32).shape)


if __name__ == "__main__":
    main()

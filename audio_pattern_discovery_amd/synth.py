"""Synthetic inputs for the alignment path.

The reference ships no audio (``*.wav`` is git-ignored) and no fixtures, so every test and
bench input is generated here, deterministically from a seed:

* lengths ~ UniformInt[L - L/32, L + L/32]  (so |n-m| <= L/16 = band at pct 0.0625),
* frames: AR(1) trajectory per dimension, f[t] = 0.9 f[t-1] + N(0,1), then per-frame mean
  removal across dimensions (mimics spectrogram.rs:74-75),
* a quarter of the sequences are noisy time-warped copies of another sequence so that
  UPGMA has real clusters to find.
"""
import numpy as np


def _ar1(rng, length, dim, rho=0.9):
    noise = rng.standard_normal((length, dim)).astype(np.float32)
    out = np.empty_like(noise)
    acc = np.zeros(dim, dtype=np.float32)
    try:
        from scipy.signal import lfilter
        out = lfilter([1.0], [1.0, -rho], noise, axis=0).astype(np.float32)
    except Exception:  # pragma: no cover - scipy is present in the image
        for t in range(length):
            acc = rho * acc + noise[t]
            out[t] = acc
    out -= out.mean(axis=1, keepdims=True)
    return out.astype(np.float32)


def _warp_copy(rng, src, length, noise=0.25):
    """Noisy, monotonically time-warped resampling of ``src`` to ``length`` frames."""
    steps = rng.uniform(0.6, 1.4, size=length)
    pos = np.cumsum(steps)
    pos = (pos - pos[0]) / max(pos[-1] - pos[0], 1e-9) * (len(src) - 1)
    idx = np.clip(np.rint(pos).astype(np.int64), 0, len(src) - 1)
    out = src[idx] + noise * rng.standard_normal((length, src.shape[1])).astype(np.float32)
    out -= out.mean(axis=1, keepdims=True)
    return out.astype(np.float32)


def make_sequences(n_seq, nominal_len, dim, seed=0, copies=0.25, integer=False, jitter=None):
    """Returns (frames [sum len][dim] float32, offsets [n_seq+1] uint64).

    ``integer=True`` rounds features to small integers to force exact ties in the
    alignment select."""
    rng = np.random.default_rng(seed)
    if jitter is None:
        jitter = max(nominal_len // 32, 0)
    lens = rng.integers(max(nominal_len - jitter, 1), nominal_len + jitter + 1, size=n_seq)
    seqs = []
    for s in range(n_seq):
        ln = int(lens[s])
        if s >= 4 and rng.random() < copies:
            seqs.append(_warp_copy(rng, seqs[int(rng.integers(0, s))], ln))
        else:
            seqs.append(_ar1(rng, ln, dim))
    if integer:
        seqs = [np.rint(q).astype(np.float32) for q in seqs]
    offsets = np.zeros(n_seq + 1, dtype=np.uint64)
    offsets[1:] = np.cumsum([len(q) for q in seqs])
    return np.concatenate(seqs, axis=0).astype(np.float32), offsets


def split(frames, offsets):
    return [frames[int(offsets[i]):int(offsets[i + 1])] for i in range(len(offsets) - 1)]


def make_audio(n_samples, seed=0):
    """Synthetic i16 audio: a few chirps plus noise (cfg 5)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples, dtype=np.float64)
    sig = np.zeros(n_samples)
    for _ in range(4):
        f0, f1 = rng.uniform(0.005, 0.2, size=2)
        phase = 2 * np.pi * (f0 * t + 0.5 * (f1 - f0) * t * t / n_samples)
        sig += rng.uniform(1000, 6000) * np.sin(phase + rng.uniform(0, 6.28))
    sig += 300.0 * rng.standard_normal(n_samples)
    return np.clip(np.rint(sig), -32768, 32767).astype(np.int16)


def make_distance_matrix(n, kind="points", seed=0):
    """A synthetic n x n distance matrix for the clustering leg (clustering.rs:81-110), zero diagonal, float32:

    * "points": clustered points in 3-d with multiplicative directed noise (d[i][j] != d[j][i], as a binding band gives),
    * "ties":   small integers -- exact linkage ties everywhere, asymmetric,
    * "inf":    two blocks at +INF from each other (length-1 sequences score +INF, alignments.rs:122) plus one +INF row,
    * "nan":    5 % NaN entries (percentile drops them, numerics.rs:127-130; a NaN linkage never wins),
    * "big":    three tight clusters -- late merges face clusters of hundreds of members (linkage chains of 10^5 terms),
    * "neg":    i.i.d. U(-1, 1): negative "distances" (what non-positive penalties can produce),
    * "uniform": i.i.d. U(0, 1)."""
    rng = np.random.default_rng(seed)
    if kind == "points":
        centres = rng.standard_normal((max(n // 64, 4), 3)) * 4
        pts = centres[rng.integers(0, len(centres), n)] + rng.standard_normal((n, 3)) * 0.3
        d = np.zeros((n, n), dtype=np.float32)
        for k in range(3):
            d += (pts[:, None, k].astype(np.float32) - pts[None, :, k].astype(np.float32)) ** 2
        d = np.sqrt(d) * (1.0 + 0.05 * rng.random((n, n), dtype=np.float32))
    elif kind == "big":
        centres = rng.standard_normal((3, 24)) * 3
        pts = (centres[rng.integers(0, 3, n)] + rng.standard_normal((n, 24)) * (0.2 + 0.8 * rng.random((n, 1)))).astype(np.float32)
        sq = (pts * pts).sum(1)
        d = np.sqrt(np.maximum(sq[:, None] + sq[None, :] - 2.0 * (pts @ pts.T), 0.0)) * (1.0 + 0.02 * rng.random((n, n), dtype=np.float32))
    elif kind == "neg":
        d = rng.random((n, n), dtype=np.float32) * 2 - 1
    elif kind == "ties":
        d = rng.integers(1, 7, size=(n, n)).astype(np.float32)
    elif kind == "inf":
        d = rng.random((n, n), dtype=np.float32) * 10
        cut = max(n // 3, 1)
        d[:cut, cut:] = np.inf
        d[cut:, :cut] = np.inf
        d[n // 2, :] = np.inf
    elif kind == "nan":
        d = rng.random((n, n), dtype=np.float32) * 10
        d[rng.random((n, n)) < 0.05] = np.nan
    else:
        d = rng.random((n, n), dtype=np.float32)
    d = np.ascontiguousarray(d, dtype=np.float32)
    np.fill_diagonal(d, 0.0)
    return d

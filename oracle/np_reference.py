"""Independent second opinion for the C oracle (TEST INFRASTRUCTURE, small cases only).

A literal Python/numpy re-derivation of the reference recurrence that shares no code
with oracle/apd_oracle.c: the DP table is a dict keyed (i, j) exactly like the
reference's HashMap (alignments.rs:99-111), every scalar is an np.float32 so each
operation rounds once as in the Rust build.  PARITY UNPINNED (see apd_oracle.h).
"""
import numpy as np

F = np.float32
INF = F(np.inf)


def euclidean(x, y):
    """numerics.rs:114-120"""
    d = F(0.0)
    for a, b in zip(x, y):
        t = F(a) - F(b)
        d = F(d + F(t * t))
    return F(np.sqrt(d))


def warping_band(pct, n_size):
    """discovery.rs:38-45 -- f32 product, saturating truncation."""
    v = F(pct) * F(n_size)
    if np.isnan(v) or v <= 0:
        return 0
    return int(v)


def dtw_pair(x, y, band, ins=1.0, dele=1.0, match=1.0):
    """alignments.rs:107-180 (Alignment::new, construct_alignment, score)."""
    x = np.asarray(x, dtype=F)
    y = np.asarray(y, dtype=F)
    n, m = len(x), len(y)
    ins, dele, match = F(ins), F(dele), F(match)
    sparse = {(0, 0): F(0.0)}                       # :109
    if n == 0 and m == 0:
        return INF                                  # :117-118
    w = max(band, abs(n - m)) + 2                   # :173
    for i in range(1, n + 1):                       # :174
        for j in range(max(i - w, 1) if i > w else 1, min(i + w, m + 1)):   # :175
            d = euclidean(x[i - 1], y[j - 1])       # :137
            ms = sparse.get((i - 1, j - 1), INF)    # :139
            is_ = sparse.get((i - 1, j), INF)       # :144
            ds = sparse.get((i, j - 1), INF)        # :149
            if ds < ms and ds < is_:                # :153
                node = F(ds + F(dele * d))
            elif is_ < ms and is_ < ds:             # :155
                node = F(is_ + F(ins * d))
            else:
                node = F(ms + F(match * d))         # :158
            sparse[(i, j)] = node                   # :177
    cell = sparse.get((n - 1, m - 1))               # :120
    if cell is None:
        return INF
    return F(cell / F(n + m))                       # :121


def align_all(seqs, band_pct, ins=1.0, dele=1.0, match=1.0):
    """alignments.rs:17-67"""
    n = len(seqs)
    out = np.zeros((n, n), dtype=F)
    for i in range(n):
        for j in range(n):
            if i != j:
                ln = max(len(seqs[i]), len(seqs[j]))
                out[i, j] = dtw_pair(seqs[i], seqs[j], warping_band(band_pct, ln), ins, dele, match)
    return out


def percentile(x, perc):
    """numerics.rs:125-133"""
    x = np.asarray(x, dtype=F).ravel()
    nf = F(len(x)) * F(perc)
    numbers = np.sort(x[~np.isnan(x)], kind="stable")
    return numbers[int(nf)]


def clustering(dist, n, perc):
    """clustering.rs:81-210, literal; HashSet order replaced by ascending ids."""
    d = np.asarray(dist, dtype=F).reshape(n, n)
    parents = list(range(n))
    threshold = percentile(d, perc)
    ops = []

    def root(i):
        while parents[i] != i:
            i = parents[i]
        return i

    n_clusters, distance = n, F(0.0)
    while n_clusters > 1 and distance < threshold:
        assign = [root(i) for i in range(n)]
        clusters = sorted(set(assign))
        best, pq = INF, (0, 0)
        for ci in clusters:
            for cj in clusters:
                if ci == cj:
                    continue
                sx, sy, acc = F(0), F(0), F(0)
                for a in range(n):
                    if assign[a] == ci:
                        sy = F(0)
                        for b in range(n):
                            if assign[b] == cj:
                                acc = F(acc + d[a, b])
                                sy = F(sy + F(1))
                        sx = F(sx + F(1))
                link = F(acc / F(sx * sy))
                if link < best:
                    best, pq = link, (ci, cj)
        p, q = pq
        k = len(parents)
        parents[p] = k
        parents[q] = k
        parents.append(k)
        n_clusters -= 1
        if p < n and q < n:
            op = "Sequence2Sequence"
        elif p >= n and q >= n:
            op = "Cluster2Cluster"
        elif p >= n and q < n:
            op = "Cluster2Sequence"
        else:
            op = "Sequence2Cluster"
        ops.append(dict(merge_i=p, merge_j=q, into=k, distance=float(best), operation=op))
        distance = best
    return ops, sorted(set(root(i) for i in range(n))), float(threshold)

"""Lane-level numpy model of the fused-pair DTW kernel schedule (design validation tool).

Models what audio_pattern_discovery_amd/csrc/dtw_kernels.hip does for ONE unordered pair
(A, B): G lanes, lane l owns C consecutive union-band offsets u = C*l + c (u = j - i + w),
macro-step tau processes row i = tau - l in every lane, neighbour values move by one-lane
shifts (DPP wave_shr / wave_shl on the GPU).  Two DP lattices share every local distance:

  DP1 = score(x=A, y=B): band u in [0, 2w-1], left neighbour = DELETE, up neighbour = INSERT
  DP2 = score(x=B, y=A): band u in [1, 2w],   up neighbour  = DELETE, left neighbour = INSERT
        (DP2 is the reference recurrence for the swapped pair written in A-row/B-column
         coordinates: its cell (j, i) is our cell (i, j).)

Run: python tools/schedule_model.py   (checks the model against the oracle on random cases)
"""
import os
import sys

import numpy as np

F = np.float32
INF = F(np.inf)


def shift_up(v, fill):      # lane l receives lane l-1 (wave_shr:1)
    out = np.empty_like(v)
    out[0] = fill
    out[1:] = v[:-1]
    return out


def shift_down(v, fill):    # lane l receives lane l+1 (wave_shl:1)
    out = np.empty_like(v)
    out[-1] = fill
    out[:-1] = v[1:]
    return out


def select(dl, ins_v, m, d, del_pen, ins_pen, mat_pen):
    """alignments.rs:153-159 vectorised over lanes."""
    pick_d = (dl < m) & (dl < ins_v)
    pick_i = (ins_v < m) & (ins_v < dl) & ~pick_d
    base = np.where(pick_d, dl, np.where(pick_i, ins_v, m))
    pen = np.where(pick_d, del_pen, np.where(pick_i, ins_pen, mat_pen)).astype(F)
    return (base + (pen * d).astype(F)).astype(F)


def fused_pair(A, B, band, ins, dele, mat, G, C):
    A = np.asarray(A, F)
    B = np.asarray(B, F)
    n, m = len(A), len(B)
    if n == 1 and m == 1:
        return F(0), F(0)
    if n == 1 or m == 1:
        return INF, INF
    w = max(band, abs(n - m)) + 2
    assert 2 * w + 1 <= C * G, "band does not fit C*G"
    lane = np.arange(G)
    u = lane[:, None] * C + np.arange(C)[None, :]           # [G][C]
    prev1 = np.full((G, C), INF, F)
    prev2 = np.full((G, C), INF, F)
    res1 = np.full(G, np.nan, F)
    res2 = np.full(G, np.nan, F)
    g_act = (2 * w + 1 + C - 1) // C
    for tau in range(0, (n - 1) + g_act):
        i = tau - lane                                       # row per lane
        left1 = shift_up(prev1[:, C - 1], INF)               # (i, u-1) from lane l-1, previous macro-step
        left2 = shift_up(prev2[:, C - 1], INF)
        cur1 = np.empty((G, C), F)
        cur2 = np.empty((G, C), F)
        upr1 = upr2 = None
        for c in range(C):
            j = i + u[:, c] - w
            xi = np.clip(i, 1, n) - 1
            yj = np.clip(j, 1, m) - 1
            diff = A[xi] - B[yj]
            d = np.sqrt((diff * diff).sum(axis=1, dtype=F)).astype(F)
            if c < C - 1:
                up1, up2 = prev1[:, c + 1], prev2[:, c + 1]
            else:
                up1, up2 = upr1, upr2
            r1 = select(left1, up1, prev1[:, c], d, dele, ins, mat)      # DP1: left=delete, up=insert
            r2 = select(up2, left2, prev2[:, c], d, dele, ins, mat)      # DP2: up=delete, left=insert
            inside = (i >= 1) & (j >= 1)
            origin = (i == 0) & (j == 0)
            v1 = inside & (u[:, c] <= 2 * w - 1)
            v2 = inside & (u[:, c] >= 1) & (u[:, c] <= 2 * w)
            r1 = np.where(v1, r1, np.where(origin, F(0), INF)).astype(F)
            r2 = np.where(v2, r2, np.where(origin, F(0), INF)).astype(F)
            cur1[:, c], cur2[:, c] = r1, r2
            hit = (i == n - 1) & (j == m - 1)
            res1 = np.where(hit, r1, res1)
            res2 = np.where(hit, r2, res2)
            left1, left2 = r1, r2
            if c == 0:
                upr1 = shift_down(r1, INF)                   # (i-1, u+1) of the last offset: lane l+1's fresh c=0
                upr2 = shift_down(r2, INF)
        prev1, prev2 = cur1, cur2
    ustar = (m - 1) - (n - 1) + w
    ls = ustar // C
    return F(res1[ls] / F(n + m)), F(res2[ls] / F(n + m))


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from oracle import binding as ob
    rng = np.random.default_rng(0)
    worst = 0.0
    for trial in range(300):
        n, m = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        dim = int(rng.integers(1, 5))
        integer = trial % 2 == 0
        A = rng.standard_normal((n, dim)).astype(F) * 2
        B = rng.standard_normal((m, dim)).astype(F) * 2
        if integer:
            A, B = np.rint(A), np.rint(B)
        pct = float(rng.choice([1.0, 0.0625, 0.25, 0.0]))
        band = ob.warping_band(pct, max(n, m))
        pens = [(1, 1, 1), (0.5, 1.5, 0.75), (0.25, 0.75, 1.0)][trial % 3]
        w = max(band, abs(n - m)) + 2
        C = int(rng.integers(2, 6))
        G = (2 * w + 1 + C - 1) // C + int(rng.integers(0, 3))
        r1, r2 = fused_pair(A, B, band, pens[0], pens[1], pens[2], G, C)
        o1 = ob.dtw_pair(A, B, band, *pens)
        o2 = ob.dtw_pair(B, A, band, *pens)
        for got, want in ((r1, o1), (r2, o2)):
            if np.isinf(want):
                assert np.isinf(got), (trial, n, m, got, want)
            else:
                err = abs(float(got) - want) / max(abs(want), 1e-30)
                worst = max(worst, err)
                assert err < 1e-5, (trial, n, m, band, C, G, got, want)
    print("schedule model matches the oracle on 300 random pairs; worst rel err %.2e" % worst)


if __name__ == "__main__":
    main()

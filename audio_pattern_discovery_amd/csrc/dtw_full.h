// Full-matrix DTW kernel (gfx950): the reference's shipped configuration (warping_band_percentage = 1.0).
//
// When the band does not bind (w >= max(n, m): every cell of the n x m matrix is visited, alignments.rs:173-175) and
// the insertion and deletion penalties are equal, score(x=a, y=b) and score(x=b, y=a) are the SAME number: the local
// distance is symmetric bit for bit ((x-y)^2 == (y-x)^2 term by term) and swapping the DELETE / INSERT predecessors
// together with their equal penalties leaves every node unchanged (alignments.rs:153-159), by induction from
// D[0][0] = 0.  So one DP serves both ordered pairs -- and there is no band to follow: lanes own COLUMN strips.
//
// One wavefront per pair.  The columns are swept in passes of 64 * CW: in a pass lane gl owns the CW consecutive
// columns j = 1 + 64*CW*pass + CW*gl + c and keeps their frames in registers for the whole pass (loaded once).
// Macro-step tau: row i = tau - gl, cells left to right.  Dependences:
//   DELETE (i, j-1): previous cell of the lane; for its first column the last cell of lane gl-1 from the previous
//                    macro-step (DPP wave_shr); for lane 0, the boundary column the previous pass left in LDS;
//   INSERT (i-1, j): the lane's own previous row;
//   MATCH (i-1, j-1): the lane's own previous row, or what it received as DELETE input one macro-step earlier.
// One cross-lane move per macro-step, no barrier (one wave), no frame traffic besides the row frame (LDS ring, as in
// dtw_systolic.h).  The boundary column is one float per row, written by lane 63 for row tau - 63 while lane 0 reads
// row tau: in place.  Rows n and columns m are not swept (alignments.rs:120).
// (Several waves per pair with a mailbox and a barrier per macro-step ran at 55-75 % of this form; removed.)
#pragma once
#include "dtw_systolic.h"

namespace apd {

template <int D, int CW, bool HYBRID>
__global__ __launch_bounds__(64) void dtw_full_matrix(const AlignLaunch L)
{
    constexpr int DN = D + 1;
    constexpr int DP = (DN + 3) & ~3;
    constexpr int U = 8;                                          // macro-steps per row-ring refill
    constexpr int R = 128;                                        // row ring: R > 2U + 64 - 2
    constexpr int W = 64 * CW;                                    // columns per pass
    constexpr int LPF = DP / 4, FPF = 64 / LPF, NFILL = (U + FPF - 1) / FPF;
    constexpr uint32_t FB = DP * 4u;
    extern __shared__ float lds[];
    float *const xring = lds;                                     // [R][DP]
    float *const bound = xring + R * DP;                          // [rows]: D[i][last column of the previous pass]
    const int gl = threadIdx.x;
    const uint32_t tile = blockIdx.x / kSlotsPerTile, slot = blockIdx.x % kSlotsPerTile;
    const PairInfo P = decode_pair(L, tile, slot);                // one pair per wavefront: everything below is uniform
    if (!P.valid) return;
    if (P.n == 1 || P.m == 1) {                                   // absent result cell (alignments.rs:120-123)
        if (gl == 0) { const float s = (P.n == 1 && P.m == 1) ? 0.0f : APD_INF; store_pair(L, tile, P, s, s); }
        return;
    }
    // rows = the longer sequence, columns = the shorter one (the result is the same number either way)
    const bool swap = P.m > P.n;
    const int n = swap ? P.m : P.n, m = swap ? P.n : P.m;
    const float *const rows = swap ? P.B : P.A, *const cols = swap ? P.A : P.B;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)L.d_frames, 0, L.frames_bytes, 0x00020000);
    const uint32_t a_off = (uint32_t)(rows - L.d_frames) * 4u, b_off = (uint32_t)(cols - L.d_frames) * 4u;
    float pen = L.band.mat, tau_thr = L.tau;
    asm volatile("" : "+v"(pen), "+v"(tau_thr));
    const int jstar = m - 1;                                      // column of the result cell (n-1, m-1)
    const int pstar = (jstar - 1) / W, lstar = ((jstar - 1) % W) / CW, cstar = (jstar - 1) % CW;
    const int n_pass = (m - 1 + W - 1) / W;                       // columns 1 .. m-1
    const int tau_cap = (n - 1) + gl;
    float res = 0.0f;

    const int fill_f = gl / LPF, fill_q = gl % LPF;
    auto fill_load = [&](int first_row, apd_f32x4 (&regs)[NFILL]) __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < NFILL; ++f) {
            const int fi = f * FPF + fill_f;
            const bool act = (fill_f < FPF) & (fi < U);
            const uint32_t off = a_off + (uint32_t)(min(first_row + fi, n) - 1) * FB + 16u * fill_q;
            regs[f] = __builtin_bit_cast(apd_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, act ? off : kNoFrame, 0, 0));
        }
    };
    auto fill_store = [&](int first_row, const apd_f32x4 (&regs)[NFILL]) __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < NFILL; ++f) {
            const int fi = f * FPF + fill_f;
            if ((fill_f < FPF) & (fi < U))
                *reinterpret_cast<apd_f32x4 *>(&xring[((first_row + fi) & (R - 1)) * DP + 4 * fill_q]) = regs[f];
        }
    };
    auto read_row = [&](float (&dst)[DN], int row) __attribute__((always_inline)) {
        const float *p = &xring[(row & (R - 1)) * DP];
#pragma unroll
        for (int q = 0; q < LPF; ++q) {
            const apd_f32x4 t = *reinterpret_cast<const apd_f32x4 *>(p + 4 * q);
            if (4 * q + 0 < DN) dst[4 * q + 0] = t.x;
            if (4 * q + 1 < DN) dst[4 * q + 1] = t.y;
            if (4 * q + 2 < DN) dst[4 * q + 2] = t.z;
            if (4 * q + 3 < DN) dst[4 * q + 3] = t.w;
        }
    };

    for (int pass = 0; pass < n_pass; ++pass) {
        const int j0 = 1 + W * pass + CW * gl;                    // first column of the lane in this pass
        const bool more = pass + 1 < n_pass;
        // lanes holding at least one needed column; every lane when another pass follows (lane 63 writes its boundary)
        const int g_act = more ? 64 : (m - 1 - W * pass + CW - 1) / CW;
        const int total_r = (((n - 1) + g_act + U - 1) / U) * U;  // macro-steps 1 .. (n-1)+g_act-1, rounded up
        float yf[CW][DN];
#pragma unroll
        for (int c = 0; c < CW; ++c) load_frame<DN>(yf[c], rsrc, b_off + (uint32_t)(min(j0 + c, m) - 1) * FB);
        float prev[CW];
#pragma unroll
        for (int c = 0; c < CW; ++c) prev[c] = APD_INF;
        float diag = APD_INF, last = APD_INF;

        asm volatile("" ::: "memory");                            // the previous pass's ring reads stay above these writes
        for (int e = gl; e < 64 * DP; e += 64)                    // rows <= 0: sentinels
            xring[((-(e / DP)) & (R - 1)) * DP + (e % DP)] = HYBRID ? ((e % DP) == D ? APD_INF : 0.0f) : -APD_INF;
        {
            apd_f32x4 regs[NFILL];
            fill_load(1, regs);
            fill_store(1, regs);
        }
        asm volatile("" ::: "memory");                            // one wave, LDS runs in order: only the compiler needs telling
        float xs[DN];
        read_row(xs, 0 - gl);

        for (int tau0 = 0; tau0 < total_r; tau0 += U) {
            apd_f32x4 fill_regs[NFILL];
            fill_load(tau0 + U + 1, fill_regs);
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const int tau = tau0 + q;
                float d[CW];
                if (HYBRID) {
                    bool any = false;
#pragma unroll
                    for (int c = 0; c < CW; ++c) {
                        float sc;
                        d[c] = frame_sq_expanded<D>(xs, yf[c], sc);
                        any |= d[c] < sc * tau_thr;
                    }
                    if (__ballot(any) != 0ull) {
#pragma unroll
                        for (int c = 0; c < CW; ++c) {
                            const float sc = xs[D] + yf[c][D];
                            const float ex = frame_sq_exact<D>(xs, yf[c]);
                            d[c] = (d[c] < sc * tau_thr) ? ex : d[c];
                        }
                    }
#pragma unroll
                    for (int c = 0; c < CW; ++c) d[c] = __builtin_amdgcn_sqrtf(d[c]);
                } else {
#pragma unroll
                    for (int c = 0; c < CW; ++c) d[c] = frame_dist<D, DN>(xs, yf[c]);
                }
                weight_distances<CW>(d, pen);
                read_row(xs, tau + 1 - gl);                       // the row frame is dead: fetch the next one under the DP row
                // DELETE input of the first column: last cell of the lane below (row i); lane 0 takes the boundary column of
                // the previous pass (row tau; rows past n-1 are never used), column 0 does not exist in pass 0
                const float edge = (pass > 0) ? bound[min(tau, n - 1)] : APD_INF;
                const float left_in = from_lower_lane(last, edge);
                float left = left_in, mdiag = diag;
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    const float up = prev[c];
                    const float r = select_node<true>(left, up, mdiag, d[c], pen, pen, pen);
                    mdiag = up;
                    prev[c] = r;
                    left = r;
                }
                diag = left_in;                                   // (i, j0 - 1) is the MATCH input of (i + 1, j0)
                if (tau == 0 && pass == 0) diag = (gl == 0) ? 0.0f : diag;   // lane 0 just swept row 0: D[0][0] = 0 is the MATCH input of cell (1, 1)
                last = left;
                if (more && gl == 63 && tau >= 63 && tau <= tau_cap) bound[tau - 63] = last;   // row tau - 63, read by lane 0 next pass
                if (pass == pstar && tau == tau_cap && gl == lstar) {
#pragma unroll
                    for (int c = 0; c < CW; ++c) if (c == cstar) res = prev[c];
                }
            }
            fill_store(tau0 + U + 1, fill_regs);
            asm volatile("" ::: "memory");
        }
    }
    if (gl == lstar) {
        const float s = res / (float)(P.n + P.m);                 // alignments.rs:121
        store_pair(L, tile, P, s, s);                             // score(a, b) == score(b, a)
    }
}

// L.w_max bounds the longer length of every pair of the launch: the boundary column (one float per row) lives in LDS
template <int D, int CW>
static hipError_t launch_full_c(const AlignLaunch &L, hipStream_t stream)
{
    constexpr int DP = (D + 1 + 3) & ~3, R = 128;
    const size_t lds_bytes = ((size_t)R * DP + (size_t)L.w_max + 16) * sizeof(float);
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;      // the dispatcher keeps such tiles off this kernel
    const dim3 grid(L.n_tiles * kSlotsPerTile), block(64);
    const bool hybrid = L.hybrid && D >= 10;
    const void *fn = hybrid ? reinterpret_cast<const void *>(dtw_full_matrix<D, CW, true>) : reinterpret_cast<const void *>(dtw_full_matrix<D, CW, false>);
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    if (hybrid) hipLaunchKernelGGL((dtw_full_matrix<D, CW, true>), grid, block, lds_bytes, stream, L);
    else hipLaunchKernelGGL((dtw_full_matrix<D, CW, false>), grid, block, lds_bytes, stream, L);
    return hipGetLastError();
}

// geometry key of the full-matrix kernel: 20100 + CW (one wavefront per pair)
template <int D>
bool launch_full(const AlignLaunch &L, int nw, int cw, hipStream_t stream, hipError_t *err)
{
    if (nw != 1) return false;
#define APD_FCASE(CC) if constexpr (CC <= max_strip_columns(D)) { if (cw == CC) { *err = launch_full_c<D, CC>(L, stream); return true; } }
    APD_FCASE(3) APD_FCASE(5) APD_FCASE(7) APD_FCASE(9) APD_FCASE(11) APD_FCASE(13)
#undef APD_FCASE
    return false;
}

}  // namespace apd

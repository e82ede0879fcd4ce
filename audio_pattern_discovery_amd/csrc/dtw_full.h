// Full-matrix DTW kernel (gfx950): the reference's shipped configuration (warping_band_percentage = 1.0).
//
// When the band does not bind (w >= max(n, m): every cell of the n x m matrix is visited, alignments.rs:173-175) and
// the insertion and deletion penalties are equal, score(x=a, y=b) and score(x=b, y=a) are the SAME number: the local
// distance is symmetric bit for bit ((x-y)^2 == (y-x)^2 term by term) and swapping the DELETE / INSERT predecessors
// together with their equal penalties leaves every node unchanged (alignments.rs:153-159), by induction from
// D[0][0] = 0.  So one DP serves both ordered pairs -- and there is no band to follow: lanes own COLUMN strips.
//
// G lanes per pair (64 / G pairs per wavefront; the pairs of a wave share their row sequence a, as in the systolic
// kernel: with the batch in length order a is the longer one).  The columns are swept in passes of G * CW: in a pass
// lane gl owns the CW consecutive columns j = 1 + G*CW*pass + CW*gl + c and keeps their frames in registers for the
// whole pass (loaded once).  Macro-step tau: row i = tau - gl, cells left to right.  Dependences:
//   DELETE (i, j-1): previous cell of the lane; for its first column the last cell of lane gl-1 from the previous
//                    macro-step (DPP); for lane 0, the boundary column the previous pass left in LDS;
//   INSERT (i-1, j): the lane's own previous row;
//   MATCH (i-1, j-1): the lane's own previous row, or what it received as DELETE input one macro-step earlier.
// One cross-lane move per macro-step, no barrier (one wave), no frame traffic besides the row frame (LDS ring, as in
// dtw_systolic.h).  The boundary column is one float per row and pair, written by lane G-1 for row tau - (G-1) while
// lane 0 reads row tau: in place.  Short sequences take small G: the skew ramp of a pass is G macro-steps, and four
// 200-frame pairs fill a wave that one of them would leave two-thirds idle.  Rows n and columns m are not swept
// (alignments.rs:120).
// (Several waves per pair with a mailbox and a barrier per macro-step ran at 55-75 % of this form; removed.)
#pragma once
#include "dtw_systolic.h"

namespace apd {

// GENERAL_PEN (only with BANDED: unequal penalties make the two ordered scores differ even under a full band): the literal
// comparison chain with DELETE / INSERT roles per DP and the reference's arithmetic operation for operation, as in the
// systolic kernel's UNIFORM_PEN = false path -- bit-identical to the CPU code.
template <int D, int CW, int G, bool HYBRID, bool BANDED, bool GENERAL_PEN = false>
__global__ __launch_bounds__(64) void dtw_full_matrix(const AlignLaunch L)
{
    constexpr int DN = D + 1;
    constexpr int DP = (DN + 3) & ~3;
    constexpr int PPW = 64 / G;                                   // pairs per wavefront
    if (L.d_nonfinite != nullptr && *L.d_nonfinite != 0u) return;   // a NaN / infinite feature in the batch: the literal kernel's job
    constexpr int U = 8;                                          // macro-steps per row-ring refill
    constexpr int R = (G == 64) ? 128 : 64;                       // row ring: R > 2U + G - 2
    constexpr int W = G * CW;                                     // columns per pass
    constexpr int LPF = DP / 4, FPF = 64 / LPF, NFILL = (U + FPF - 1) / FPF;
    constexpr uint32_t FB = DP * 4u;
    extern __shared__ float lds[];
    float *const xring = lds;                                     // [R][DP]
    const int lane = threadIdx.x, gl = lane % G;
    // [rows] per pair and DP: D[i][last column of the previous pass]
    float *const bound = xring + R * DP + (lane / G) * (BANDED ? 2 : 1) * ((int)L.n_max + 4);
    float *const bound2 = bound + ((int)L.n_max + 4);
    constexpr uint32_t WPT = kSlotsPerTile / PPW;                 // wavefronts per tile
    const uint32_t tile = blockIdx.x / WPT, slot = (blockIdx.x % WPT) * PPW + lane / G;
    const PairInfo P = decode_pair(L, tile, slot);                // the pairs of a wave have the same a (slot / 16)
    const bool special = P.valid && (P.n == 1 || P.m == 1);       // absent result cell (alignments.rs:120-123)
    const bool sweep = P.valid && !special;
    if (special && gl == 0) { const float s = (P.n == 1 && P.m == 1) ? 0.0f : APD_INF; store_pair(L, tile, P, s, s); }
    if (__ballot(sweep) == 0ull) return;
    // rows = a, columns = b (either orientation gives the same number; in length order a is the longer one)
    const int lead = __builtin_ctzll(__ballot(sweep));
    const int n = __builtin_amdgcn_readlane(P.n, lead);           // shared by every sweeping pair of the wave
    int m = sweep ? P.m : 2;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)L.d_frames, 0, L.frames_bytes, 0x00020000);
    const uint32_t a_off = __builtin_amdgcn_readlane((uint32_t)(P.A - L.d_frames) * 4u, lead);
    uint32_t b_off = (uint32_t)(P.B - L.d_frames) * 4u;
    if (G == 64) {                                                // one pair per wave: its geometry is wave-uniform (scalar registers)
        m = __builtin_amdgcn_readfirstlane(m);
        b_off = __builtin_amdgcn_readfirstlane(b_off);
    }
    static_assert(!GENERAL_PEN || (BANDED && !HYBRID), "general penalties: two DPs on strict distances");
    float pen = L.band.mat, tau_thr = L.tau, p_ins = L.band.ins, p_del = L.band.del, nmax_ab = P.nmax_ab;
    asm volatile("" : "+v"(pen), "+v"(tau_thr), "+v"(p_ins), "+v"(p_del), "+v"(nmax_ab));
    // BANDED: the band binds.  Band offset u = j - i + w; score(a, b) lives on u in [0, 2w-1], score(b, a) -- the swapped
    // pair's recurrence transposed, the same select for equal penalties -- on u in [1, 2w] (see dtw_generic.hip).  Cells
    // outside get a local distance of +INF, which makes the node +INF whatever its predecessors are.
    const int two_w = BANDED ? 2 * (sweep ? P.w : 2) : 0;
    const int jstar = m - 1;                                      // column of the result cell (n-1, m-1)
    const int pstar = (jstar - 1) / W, lstar = ((jstar - 1) % W) / CW, cstar = (jstar - 1) % CW;
    const int my_pass = sweep ? (m - 1 + W - 1) / W : 0;          // passes of this pair: columns 1 .. m-1
    int n_pass = 0;
#pragma unroll
    for (int g = 0; g < PPW; ++g) n_pass = max(n_pass, __builtin_amdgcn_readlane(my_pass, g * G));
    const int tau_cap = (n - 1) + gl;
    const int total_r = (((n - 1) + G + U - 1) / U) * U;          // macro-steps 1 .. (n-1)+G-1 of a pass, rounded up
    float res = 0.0f, res2 = 0.0f;

    const int fill_f = lane / LPF, fill_q = lane % LPF;
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
            if ((fill_f < FPF) & (fi < U)) {
                apd_f32x4 v = regs[f];
                if (HYBRID) prescale_row_piece<D>(v, fill_q);      // -2 x, norm slot kept: see frame_sq_expanded_pre
                *reinterpret_cast<apd_f32x4 *>(&xring[((first_row + fi) & (R - 1)) * DP + 4 * fill_q]) = v;
            }
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
        const bool more = pass + 1 < my_pass;                     // this pair has another pass: lane G-1 leaves its boundary column
        float yf[CW][DN];
#pragma unroll
        for (int c = 0; c < CW; ++c) load_frame<DN>(yf[c], rsrc, b_off + (uint32_t)(min(j0 + c, m) - 1) * FB);
        float prev[CW], prev2[CW];
#pragma unroll
        for (int c = 0; c < CW; ++c) { prev[c] = APD_INF; prev2[c] = APD_INF; }
        float diag = APD_INF, last = APD_INF, diag2 = APD_INF, last2 = APD_INF;
        const int u_base = j0 + gl + (two_w >> 1);                // band offset of the lane's first cell at macro-step 0

        asm volatile("" ::: "memory");                            // the previous pass's ring reads stay above these writes
        for (int e = lane; e < G * DP; e += 64)                   // rows <= 0: sentinels
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
                    // one superset test per macro-step instead of one per cell (see dtw_systolic.h)
#pragma unroll
                    for (int c = 0; c < CW; ++c) d[c] = frame_sq_expanded_pre<D>(xs, yf[c]);
                    float dmin = d[0];
#pragma unroll
                    for (int c = 1; c + 1 < CW; c += 2) dmin = __builtin_fminf(__builtin_fminf(dmin, d[c]), d[c + 1]);
                    if (CW % 2 == 0) dmin = __builtin_fminf(dmin, d[CW - 1]);
                    const bool any = dmin < (xs[D] + nmax_ab) * tau_thr;
                    if (__builtin_expect(__ballot(any) != 0ull, 0)) {   // rare: placed out of the hot instruction stream
                        float nx = xs[D];
                        asm volatile("" : "+v"(nx));              // see dtw_systolic.h: keeps the hot path's norm sums out of this block
#pragma unroll
                        for (int c = 0; c < CW; ++c) {
                            const float sc = nx + yf[c][D];
                            const float ex = frame_sq_exact_pre<D>(xs, yf[c]);
                            d[c] = (d[c] < sc * tau_thr) ? ex : d[c];
                        }
                    }
#pragma unroll
                    for (int c = 0; c < CW; ++c) d[c] = __builtin_amdgcn_sqrtf(d[c]);
                } else {
#pragma unroll
                    for (int c = 0; c < CW; ++c) d[c] = GENERAL_PEN ? frame_dist_strict<D, DN>(xs, yf[c]) : frame_dist<D, DN>(xs, yf[c]);
                }
                if (!GENERAL_PEN && !HYBRID) weight_distances<CW>(d, pen);   // the hybrid form is only launched with unit penalties
                read_row(xs, tau + 1 - gl);                       // the row frame is dead: fetch the next one under the DP row
                // DELETE input of the first column: last cell of the lane below (row i); lane 0 of a pair takes the boundary
                // column of the previous pass (row tau; rows past n-1 are never used), column 0 does not exist in pass 0
                const float edge = (pass > 0) ? bound[min(tau, n - 1)] : APD_INF;
                const float left_in = group_from_lower<G>(last, edge, gl);
                float left_in2 = APD_INF;
                if (BANDED) {
                    const float edge2 = (pass > 0) ? bound2[min(tau, n - 1)] : APD_INF;
                    left_in2 = group_from_lower<G>(last2, edge2, gl);
                }
                float d2[CW];
                if (BANDED) {
                    // the lane's cells sit at offsets t .. t + CW - 1; only where a band edge crosses them is anything masked
                    const int t = u_base - tau;
                    const bool edge_here = (t < 1) | (t + CW - 1 > two_w - 1);
#pragma unroll
                    for (int c = 0; c < CW; ++c) d2[c] = d[c];
                    if (__ballot(edge_here) != 0ull) {
#pragma unroll
                        for (int c = 0; c < CW; ++c) {
                            const unsigned u = (unsigned)(t + c);
                            d2[c] = (u - 1u > (unsigned)(two_w - 1)) ? APD_INF : d[c];     // score(b, a): u in [1, 2w]
                            d[c] = (u > (unsigned)(two_w - 1)) ? APD_INF : d[c];           // score(a, b): u in [0, 2w-1]
                        }
                    }
                }
                float left = left_in, mdiag = diag, left2 = left_in2, mdiag2 = diag2;
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    const float up = prev[c];
                    const float r = GENERAL_PEN ? select_node<false>(left, up, mdiag, d[c], p_del, p_ins, pen)     // score(a, b): left = DELETE, up = INSERT
                                                : select_node<true>(left, up, mdiag, d[c], pen, pen, pen);
                    mdiag = up;
                    prev[c] = r;
                    left = r;
                    if (BANDED) {
                        const float up2 = prev2[c];
                        const float r2 = GENERAL_PEN ? select_node<false>(up2, left2, mdiag2, d2[c], p_del, p_ins, pen)   // score(b, a): up = DELETE, left = INSERT
                                                     : select_node<true>(left2, up2, mdiag2, d2[c], pen, pen, pen);
                        mdiag2 = up2;
                        prev2[c] = r2;
                        left2 = r2;
                    }
                }
                diag = left_in;                                   // (i, j0 - 1) is the MATCH input of (i + 1, j0)
                diag2 = left_in2;
                if (tau == 0 && pass == 0) { diag = (gl == 0) ? 0.0f : diag; diag2 = (gl == 0) ? 0.0f : diag2; }   // lane 0 just swept row 0: D[0][0] = 0 is the MATCH input of cell (1, 1)
                last = left;
                last2 = left2;
                if (more && gl == G - 1 && tau >= G - 1 && tau <= tau_cap) {   // row tau - (G-1), read by lane 0 next pass
                    bound[tau - (G - 1)] = last;
                    if (BANDED) bound2[tau - (G - 1)] = last2;
                }
                if (pass == pstar && tau == tau_cap && gl == lstar) {
#pragma unroll
                    for (int c = 0; c < CW; ++c) if (c == cstar) { res = prev[c]; res2 = prev2[c]; }
                }
            }
            fill_store(tau0 + U + 1, fill_regs);
            asm volatile("" ::: "memory");
        }
    }
    if (sweep && gl == lstar) {
        const float denom = (float)(P.n + P.m);                   // alignments.rs:121
        store_pair(L, tile, P, res / denom, (BANDED ? res2 : res) / denom);   // without a binding band score(a, b) == score(b, a)
    }
}

// L.n_max bounds the longer length of every pair of the launch: the boundary column (one float per row, pair and DP) lives in LDS
template <int D, int CW, int G>
static hipError_t launch_full_general(const AlignLaunch &L, hipStream_t stream)
{
    constexpr int DP = (D + 1 + 3) & ~3, R = (G == 64) ? 128 : 64, PPW = 64 / G;
    const size_t lds_bytes = ((size_t)R * DP + (size_t)PPW * 2 * (L.n_max + 4) + 16) * sizeof(float);
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;
    const dim3 grid(L.n_tiles * (kSlotsPerTile / PPW)), block(64);
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(dtw_full_matrix<D, CW, G, false, true, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((dtw_full_matrix<D, CW, G, false, true, true>), grid, block, lds_bytes, stream, L);
    return hipGetLastError();
}

template <int D, int CW, int G, bool BANDED>
static hipError_t launch_full_c(const AlignLaunch &L, hipStream_t stream)
{
    constexpr int DP = (D + 1 + 3) & ~3, R = (G == 64) ? 128 : 64, PPW = 64 / G;
    const size_t lds_bytes = ((size_t)R * DP + (size_t)PPW * (BANDED ? 2 : 1) * (L.n_max + 4) + 16) * sizeof(float);
    if (lds_bytes > 160 * 1024) return hipErrorInvalidValue;      // the dispatcher keeps such tiles off this kernel
    const dim3 grid(L.n_tiles * (kSlotsPerTile / PPW)), block(64);
    const bool hybrid = L.hybrid && D >= 10 && L.band.mat == 1.0f;   // (equal penalties here: all three are 1)
    const void *fn = hybrid ? reinterpret_cast<const void *>(dtw_full_matrix<D, CW, G, true, BANDED>)
                            : reinterpret_cast<const void *>(dtw_full_matrix<D, CW, G, false, BANDED>);
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    if (hybrid) hipLaunchKernelGGL((dtw_full_matrix<D, CW, G, true, BANDED>), grid, block, lds_bytes, stream, L);
    else hipLaunchKernelGGL((dtw_full_matrix<D, CW, G, false, BANDED>), grid, block, lds_bytes, stream, L);
    return hipGetLastError();
}

// geometry key of the column-strip kernels: 20000 (band never binds, one DP) or 30000 (banded, two DPs)
//                                          + (pairs per wavefront = 64 / G) * 100 + CW
template <int D>
bool launch_full(const AlignLaunch &L, bool banded, int ppw, int cw, hipStream_t stream, hipError_t *err)
{
#define APD_FCASE(CC) if constexpr (CC <= max_strip_columns(D)) { if (cw == CC && !banded) { \
        if (ppw == 1) { *err = launch_full_c<D, CC, 64, false>(L, stream); return true; } \
        if (ppw == 2) { *err = launch_full_c<D, CC, 32, false>(L, stream); return true; } \
        if (ppw == 4) { *err = launch_full_c<D, CC, 16, false>(L, stream); return true; } } }
    APD_FCASE(3) APD_FCASE(5) APD_FCASE(7) APD_FCASE(9) APD_FCASE(11) APD_FCASE(13)
#undef APD_FCASE
    const bool general = !((L.band.ins == L.band.del) && (L.band.del == L.band.mat)) || L.strict;   // unequal penalties (or strict mode): literal select, strict distances
#define APD_BCASE(CC) if constexpr (CC <= max_cells_per_lane(D)) { if (cw == CC && banded) { \
        if (ppw == 1) { *err = general ? launch_full_general<D, CC, 64>(L, stream) : launch_full_c<D, CC, 64, true>(L, stream); return true; } \
        if (ppw == 4) { *err = general ? launch_full_general<D, CC, 16>(L, stream) : launch_full_c<D, CC, 16, true>(L, stream); return true; } } }
    APD_BCASE(5) APD_BCASE(9)
#undef APD_BCASE
    return false;
}

}  // namespace apd

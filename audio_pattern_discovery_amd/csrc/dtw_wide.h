// Wide-band fused-pair DTW: the systolic kernel of dtw_systolic.h stretched over NW wavefronts (gfx950).
//
// dtw_fused_systolic keeps a pair inside one wavefront, so it tops out at 64 lanes x 9 offsets = 576 band offsets --
// full DTW (the reference's shipped warping_band_percentage = 1.0) on anything longer than ~285 frames would fall to
// the generic kernel, ~40x slower per cell.  Here ONE workgroup of NW waves (128, 256 or 512 lanes) sweeps one unordered
// pair: lane gl = threadIdx.x owns offsets u = C*gl + c exactly as before, neighbours inside a wave talk through DPP,
// and the NW-1 wave seams go through tiny LDS mailboxes:
//   left[wv+1]  last DP cells of lane 63 of wave wv      (read by lane 0 of wave wv+1 at the next macro-step)
//   up[wv]      first DP cells of lane 0 of wave wv      (read by lane 63 of wave wv-1 in the SAME macro-step)
//   ycol[wv]    the column frame lane 0 of wave wv hands down (read by lane 63 of wave wv-1 at the shift)
// with two workgroup barriers per macro-step (after the first cell, after the last).  Entry ycol[NW] is fed by a
// one-macro-step-ahead prefetch of the column entering the band, up[NW] / left[0] are constant +INF.  The row-frame
// ring is shared by the whole workgroup.  Capacity: NW = 8, C = 9 -> 4608 offsets (full DTW up to ~2300 frames).
#pragma once
#include "dtw_systolic.h"

namespace apd {

template <int D, int C, int NW, bool HYBRID>
__global__ __launch_bounds__(64 * NW) void dtw_fused_wide(const AlignLaunch L)
{
    constexpr int G = 64 * NW;
    constexpr int DN = D + 1;
    if (L.d_nonfinite != nullptr && *L.d_nonfinite != 0u) return;   // a NaN / infinite feature in the batch: the literal kernel's job
    constexpr int DP = (DN + 3) & ~3;
    constexpr int S = C + 1;
    constexpr int U = (S % 2 == 0) ? S : 2 * S;
    constexpr int R = (G <= 256) ? 512 : 1024;                   // row ring: R > 2U + G - 2
    constexpr int LPF = DP / 4, FPF = 64 / LPF, NFILL = (U + FPF - 1) / FPF;
    constexpr uint32_t FB = DP * 4u;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *const xring = lds;                                     // [R][DP]
    float *const mb_y = xring + R * DP;                           // [NW + 1][DP]
    float *const mb_left = mb_y + (NW + 1) * DP;                  // [NW + 1][2]
    float *const mb_up = mb_left + (NW + 1) * 2;                  // [NW + 1][2]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, gl = threadIdx.x;
    const uint32_t tile = blockIdx.x / kSlotsPerTile, slot = blockIdx.x % kSlotsPerTile;
    const PairInfo P = decode_pair(L, tile, slot);                // one pair per workgroup: everything below is uniform
    if (!P.valid) return;
    if (P.n == 1 || P.m == 1) {                                   // absent result cell (alignments.rs:120-123)
        if (gl == 0) { const float s = (P.n == 1 && P.m == 1) ? 0.0f : APD_INF; store_pair(L, tile, P, s, s); }
        return;
    }
    const int n = P.n, m = P.m, w = P.w;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)L.d_frames, 0, L.frames_bytes, 0x00020000);
    const uint32_t a_off = (uint32_t)(P.A - L.d_frames) * 4u, b_off = (uint32_t)(P.B - L.d_frames) * 4u;
    float pen = L.band.mat, tau_thr = L.tau, nmax_ab = P.nmax_ab;   // uniform penalties only (the dispatcher guarantees it)
    asm volatile("" : "+v"(pen), "+v"(tau_thr), "+v"(nmax_ab));
    const int u0 = C * gl, two_w = 2 * w;
    bool g1[C], g2[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int u = u0 + c;
        g1[c] = u >= two_w;
        g2[c] = (u == 0) | (u > two_w);
    }
    const int cw = w - u0;
    const int ustar = (m - 1) - (n - 1) + w, cstar = ustar - u0;
    const int tau_cap = (n - 1) + gl;
    const int g_act = (two_w + 1 + C - 1) / C;
    const int total_r = (((n - 1) + g_act + U - 1) / U) * U;
    const int a_end = min(((G + U - 1) / U) * U, total_r);
    const int b_end = min(max(((n - 1) / U) * U, a_end), total_r);

    float prev1[C], prev2[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { prev1[c] = APD_INF; prev2[c] = APD_INF; }
    float res1 = 0.0f, res2 = 0.0f;
    const int ysent = HYBRID ? m : m + 1;                         // sentinel H / sentinel E for columns <= 0
    float yf[S][DN];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int j = u0 - gl + c - w;
        load_frame<DN>(yf[c], rsrc, b_off + (uint32_t)((j >= 1) ? (min(j, m) - 1) : ysent) * FB);
    }
#pragma unroll
    for (int k = 0; k < DN; ++k) yf[C][k] = 0.0f;

    // ---- shared state: row ring (rows <= 0 are sentinels), constant mailboxes
    const int fill_f = lane / LPF, fill_q = lane % LPF;
    auto fill_load = [&](int first_row, apd_f32x4 (&regs)[NFILL]) __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < NFILL; ++f) {
            const int fi = f * FPF + fill_f;
            const bool act = (wv == 0) & (fill_f < FPF) & (fi < U);
            const uint32_t off = a_off + (uint32_t)(min(first_row + fi, n) - 1) * FB + 16u * fill_q;
            regs[f] = __builtin_bit_cast(apd_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, act ? off : kNoFrame, 0, 0));
        }
    };
    auto fill_store = [&](int first_row, const apd_f32x4 (&regs)[NFILL]) __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < NFILL; ++f) {
            const int fi = f * FPF + fill_f;
            if ((wv == 0) & (fill_f < FPF) & (fi < U)) {
                apd_f32x4 v = regs[f];
                if (HYBRID) prescale_row_piece<D>(v, fill_q);      // -2 x, norm slot kept: see frame_sq_expanded_pre
                *reinterpret_cast<apd_f32x4 *>(&xring[((first_row + fi) & (R - 1)) * DP + 4 * fill_q]) = v;
            }
        }
    };
    for (int e = gl; e < G * DP; e += G)
        xring[((-(e / DP)) & (R - 1)) * DP + (e % DP)] = HYBRID ? ((e % DP) == D ? APD_INF : 0.0f) : -APD_INF;
    if (gl < 2 * (NW + 1)) { mb_left[gl] = APD_INF; mb_up[gl] = APD_INF; }      // left[0] and up[NW] stay +INF for good
    for (int e = gl; e < (NW + 1) * DP; e += G) mb_y[e] = 0.0f;
    {
        apd_f32x4 regs[NFILL];
        fill_load(1, regs);
        fill_store(1, regs);
    }
    // column entering the band at macro-step 1, fetched one macro-step ahead by the top wave (one dword per lane)
    const bool pre_lane = (wv == NW - 1) & (lane < DP);
    auto pre_load = [&](int tau_next) __attribute__((always_inline)) {
        const int jt = tau_next + (C - 1) * G - w;
        const uint32_t yo = b_off + (uint32_t)((jt >= 1) ? (min(jt, m) - 1) : ysent) * FB + 4u * lane;
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, pre_lane ? yo : kNoFrame, 0, 0));
    };
    float ypre = pre_load(1);
    __syncthreads();
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
    float xs[DN];
    read_row(xs, 0 - gl);

    auto macro_steps = [&](int tau_begin, int tau_end, auto slow_tag) __attribute__((always_inline)) {
        constexpr bool SLOW = decltype(slow_tag)::value;
        for (int tau0 = tau_begin; tau0 < tau_end; tau0 += U) {
            apd_f32x4 fill_regs[NFILL];
            fill_load(tau0 + U + 1, fill_regs);
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const int tau = tau0 + q;
                const int r = q % S, e = (r + C) % S;
                // seam traffic of this macro-step: hand the column frame down, publish the prefetched entering column
                if (lane == 0) {                                   // 16-byte stores: one LDS instruction per four components
#pragma unroll
                    for (int q4 = 0; q4 < LPF; ++q4) {
                        apd_f32x4 t;
                        t.x = yf[(r + 1) % S][4 * q4];
                        t.y = (4 * q4 + 1 < DN) ? yf[(r + 1) % S][4 * q4 + 1] : 0.0f;
                        t.z = (4 * q4 + 2 < DN) ? yf[(r + 1) % S][4 * q4 + 2] : 0.0f;
                        t.w = (4 * q4 + 3 < DN) ? yf[(r + 1) % S][4 * q4 + 3] : 0.0f;
                        *reinterpret_cast<apd_f32x4 *>(&mb_y[wv * DP + 4 * q4]) = t;
                    }
                }
                if (pre_lane) mb_y[NW * DP + lane] = ypre;
                ypre = pre_load(tau + 2);
                const float2 lf = *reinterpret_cast<const float2 *>(&mb_left[wv * 2]);   // written before the previous barrier
                const float lf1 = lf.x, lf2 = lf.y;
                // distances
                float d[C];
                if (HYBRID) {
                    // one superset test per macro-step instead of one per cell (see dtw_systolic.h)
#pragma unroll
                    for (int c = 0; c < C; ++c) d[c] = frame_sq_expanded_pre<D>(xs, yf[(r + c) % S]);
                    float dmin = d[0];
#pragma unroll
                    for (int c = 1; c + 1 < C; c += 2) dmin = __builtin_fminf(__builtin_fminf(dmin, d[c]), d[c + 1]);
                    if (C % 2 == 0) dmin = __builtin_fminf(dmin, d[C - 1]);
                    const bool any = dmin < (xs[D] + nmax_ab) * tau_thr;
                    if (__builtin_expect(__ballot(any) != 0ull, 0)) {   // rare: placed out of the hot instruction stream
                        float nx = xs[D];
                        asm volatile("" : "+v"(nx));              // see dtw_systolic.h: keeps the hot path's norm sums out of this block
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            const float sc = nx + yf[(r + c) % S][D];
                            const float ex = frame_sq_exact_pre<D>(xs, yf[(r + c) % S]);
                            d[c] = (d[c] < sc * tau_thr) ? ex : d[c];
                        }
                    }
#pragma unroll
                    for (int c = 0; c < C; ++c) d[c] = __builtin_amdgcn_sqrtf(d[c]);
                } else {
#pragma unroll
                    for (int c = 0; c < C; ++c) d[c] = frame_dist<D, DN>(xs, yf[(r + c) % S]);
                }
                if (!HYBRID) weight_distances<C>(d, pen);          // the hybrid form is only launched with unit penalties
                // first cell, then the seam exchange of the "up" neighbour
                float left1 = from_lower_lane(prev1[C - 1], lf1);
                float left2 = from_lower_lane(prev2[C - 1], lf2);
                {
                    const float r1 = select_node<true>(left1, prev1[1 % C], prev1[0], d[0], pen, pen, pen, g1[0]);
                    const float r2 = select_node<true>(prev2[1 % C], left2, prev2[0], d[0], pen, pen, pen, g2[0]);
                    prev1[0] = r1; prev2[0] = r2;
                    left1 = r1; left2 = r2;
                    if (lane == 0) *reinterpret_cast<float2 *>(&mb_up[wv * 2]) = make_float2(r1, r2);
                }
                __syncthreads();                                  // barrier A
                const float2 uf = *reinterpret_cast<const float2 *>(&mb_up[(wv + 1) * 2]);
                const float upr1 = from_upper_lane(prev1[0], uf.x), upr2 = from_upper_lane(prev2[0], uf.y);
#pragma unroll
                for (int q4 = 0; q4 < LPF; ++q4) {                // broadcast read into the dead slot: lane 63's hand-down
                    const apd_f32x4 t = *reinterpret_cast<const apd_f32x4 *>(&mb_y[(wv + 1) * DP + 4 * q4]);
                    yf[e][4 * q4] = t.x;
                    if (4 * q4 + 1 < DN) yf[e][4 * q4 + 1] = t.y;
                    if (4 * q4 + 2 < DN) yf[e][4 * q4 + 2] = t.z;
                    if (4 * q4 + 3 < DN) yf[e][4 * q4 + 3] = t.w;
                }
                read_row(xs, tau + 1 - gl);                       // the row frame is dead: fetch the next one
#pragma unroll
                for (int c = 1; c < C; ++c) {
                    const float up1 = (c < C - 1) ? prev1[(c + 1) % C] : upr1;
                    const float up2 = (c < C - 1) ? prev2[(c + 1) % C] : upr2;
                    const float r1 = select_node<true>(left1, up1, prev1[c], d[c], pen, pen, pen, g1[c]);
                    const float r2 = select_node<true>(up2, left2, prev2[c], d[c], pen, pen, pen, g2[c]);
                    prev1[c] = r1; prev2[c] = r2;
                    left1 = r1; left2 = r2;
                }
                if (lane == 63) *reinterpret_cast<float2 *>(&mb_left[(wv + 1) * 2]) = make_float2(left1, left2);
                if (SLOW) {
                    if (tau == gl) {
#pragma unroll
                        for (int c = 0; c < C; ++c) if (c == cw) { prev1[c] = 0.0f; prev2[c] = 0.0f; }
                    }
                    if (tau == tau_cap) {
#pragma unroll
                        for (int c = 0; c < C; ++c) if (c == cstar) { res1 = prev1[c]; res2 = prev2[c]; }
                    }
                }
#pragma unroll
                for (int k = 0; k < (HYBRID ? DN : D); ++k) yf[e][k] = from_upper_lane(yf[(r + 1) % S][k], yf[e][k]);
                __syncthreads();                                  // barrier B
            }
            fill_store(tau0 + U + 1, fill_regs);
        }
    };
    macro_steps(0, a_end, std::true_type{});
    macro_steps(a_end, b_end, std::false_type{});
    macro_steps(b_end, total_r, std::true_type{});

    if (gl == ustar / C) {
        const float denom = (float)(n + m);                       // alignments.rs:121
        store_pair(L, tile, P, res1 / denom, res2 / denom);
    }
}

template <int D, int C, int NW>
static hipError_t launch_wide_cn(const AlignLaunch &L, hipStream_t stream)
{
    constexpr int G = 64 * NW, DP = (D + 1 + 3) & ~3, R = (G <= 256) ? 512 : 1024;
    const size_t lds_bytes = ((size_t)R * DP + (NW + 1) * DP + 4 * (NW + 1)) * sizeof(float);
    const dim3 grid(L.n_tiles * kSlotsPerTile), block(G);
    const bool hybrid = L.hybrid && D >= 10 && L.band.mat == 1.0f;   // (equal penalties here: all three are 1)
    const void *fn = hybrid ? reinterpret_cast<const void *>(dtw_fused_wide<D, C, NW, true>) : reinterpret_cast<const void *>(dtw_fused_wide<D, C, NW, false>);
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    if (hybrid) hipLaunchKernelGGL((dtw_fused_wide<D, C, NW, true>), grid, block, lds_bytes, stream, L);
    else hipLaunchKernelGGL((dtw_fused_wide<D, C, NW, false>), grid, block, lds_bytes, stream, L);
    return hipGetLastError();
}

// geometry key for the wide kernel: 10000 + NW * 100 + C
template <int D>
bool launch_wide(const AlignLaunch &L, int nw, int c, hipStream_t stream, hipError_t *err)
{
#define APD_WCASE(NN, CC) if constexpr (CC <= max_cells_per_lane(D)) { if (nw == NN && c == CC) { *err = launch_wide_cn<D, CC, NN>(L, stream); return true; } }
    APD_WCASE(2, 5) APD_WCASE(2, 7) APD_WCASE(2, 9) APD_WCASE(4, 5) APD_WCASE(4, 7) APD_WCASE(4, 9) APD_WCASE(8, 5) APD_WCASE(8, 7) APD_WCASE(8, 9)
#undef APD_WCASE
    return false;
}

}  // namespace apd

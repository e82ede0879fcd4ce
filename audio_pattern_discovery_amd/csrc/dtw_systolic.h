// Systolic fused-pair banded DTW (the production kernel), gfx950.
//
// See dtw_generic.hip for the recurrence and the DP1/DP2 fusion.  This kernel adds:
//  * G lanes per unordered pair (64/G pairs per wavefront), lane gl owns C band offsets u = C*gl + c, so that a
//    band of 2w+1 offsets fills the wave instead of leaving lanes idle (w = 66 -> G = 16, C = 9: 133/144 busy);
//  * frames never touch LDS: the row frame x[i] enters at lane 0 of a group and moves one lane up per
//    macro-step, column frames y[j] enter at lane G-1 and move one lane down (DPP row/wave shifts), each lane
//    keeping its C-column window in registers.  One frame of each sequence is read from memory per macro-step,
//    by one lane: the algorithmic minimum 4*D*(n+m) bytes per ordered pair is also what the kernel fetches;
//  * no per-cell boundary tests: rows <= 0 use an x frame of -INF, columns <= 0 the +INF sentinel frame stored
//    behind every sequence, so those cells evaluate to +INF by arithmetic (penalties must be > 0; otherwise the
//    dispatcher takes the generic kernel); D[0][0] = 0 (alignments.rs:109) is injected and the result cell
//    (n-1, m-1) (alignments.rs:120) captured only in the first G and last G macro-steps ("slow" phases);
//  * the static band edges (DP1 stops at u = 2w-1, DP2 spans u = 1..2w) are two per-lane masks per offset.
#pragma once
#include <type_traits>

#include "dtw_common.h"

namespace apd {

typedef float apd_f32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t kNoFrame = 0xFFFFFF00u;   // beyond every descriptor (frames_bytes < kNoFrame), +imm offsets cannot wrap

// One padded frame through a buffer descriptor.  A lane whose byte offset is out of range (kNoFrame) gets
// zeros from the range check without touching memory: the edge-lane fetches need no branch and no select.
template <int D>
__device__ __forceinline__ void load_frame(float (&dst)[D], __amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off)
{
    constexpr int DP4 = (D + 3) / 4;
#pragma unroll
    for (int q = 0; q < DP4; ++q) {
        const apd_f32x4 t = __builtin_bit_cast(apd_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off + 16u * q, 0, 0));
        if (4 * q + 0 < D) dst[4 * q + 0] = t.x;
        if (4 * q + 1 < D) dst[4 * q + 1] = t.y;
        if (4 * q + 2 < D) dst[4 * q + 2] = t.z;
        if (4 * q + 3 < D) dst[4 * q + 3] = t.w;
    }
}

// Same, one dword per instruction: no register-tuple constraint on the destination, so a frame can be fetched in
// place into registers that the DPP shifts also write (the allocator then needs no copies on the loop back-edge).
template <int D>
__device__ __forceinline__ void load_frame_dwords(float (&dst)[D], __amdgpu_buffer_rsrc_t rsrc, uint32_t byte_off)
{
#pragma unroll
    for (int k = 0; k < D; ++k)
        dst[k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, byte_off + 4u * k, 0, 0));
}

// numerics.rs:114-120 as an fma chain, k ascending; v_sqrt_f32 (<= 1 ulp).
template <int D>
__device__ __forceinline__ float frame_dist(const float (&x)[D], const float (&y)[D])
{
    float t = x[0] - y[0];
    float acc = t * t;
#pragma unroll
    for (int k = 1; k < D; ++k) {
        t = x[k] - y[k];
        acc = __builtin_fmaf(t, t, acc);
    }
    return __builtin_amdgcn_sqrtf(acc);
}

template <int D, int C, int G, bool UNIFORM_PEN>
__global__ __launch_bounds__(256) void dtw_fused_systolic(const AlignLaunch L)
{
    static_assert(C >= 2 && (64 % G) == 0, "bad geometry");
    constexpr int DP = (D + 3) & ~3;
    constexpr int PPW = 64 / G;                                // pairs per wave
    constexpr int WPT = kSlotsPerTile / (4 * PPW);             // workgroups per tile
    const int lane = threadIdx.x & 63;
    const int gl = lane % G;
    // XCD-aware placement: blocks b and b+8 share an XCD (and its L2); keep a tile's workgroups on one XCD.
    const uint32_t xcd = blockIdx.x & 7u, q = blockIdx.x >> 3;
    const uint32_t tile = (q / WPT) * 8u + xcd;
    const uint32_t slot = ((q % WPT) * 4u + (threadIdx.x >> 6)) * PPW + lane / G;
    const PairInfo P = decode_pair(L, tile, slot);
    const bool special = P.valid && (P.n == 1 || P.m == 1);   // absent result cell (alignments.rs:120-123)
    const bool sweep = P.valid && !special;
    if (__ballot(sweep) == 0ull) {
        if (special && gl == 0) { const float s = (P.n == 1 && P.m == 1) ? 0.0f : APD_INF; store_pair(L, tile, P, s, s); }
        return;
    }
    const int n = sweep ? P.n : 1, m = sweep ? P.m : 1, w = sweep ? P.w : 2;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)L.d_frames, 0, L.frames_bytes, 0x00020000);
    const uint32_t a_off = (uint32_t)(P.A - L.d_frames) * 4u, b_off = (uint32_t)(P.B - L.d_frames) * 4u;   // bytes
    constexpr uint32_t FB = DP * 4u;                             // bytes per padded frame
    const float ins = L.band.ins, del = L.band.del, mat = L.band.mat;
    const int u0 = C * gl;
    const int two_w = 2 * w;

    // static band guards
    bool g1[C], g2[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int u = u0 + c;
        g1[c] = u >= two_w;                       // DP1 spans u in [0, 2w-1]
        g2[c] = (u == 0) | (u > two_w);           // DP2 spans u in [1, 2w]
    }
    const int cw = w - u0;                        // offset index holding u == w (cell (0,0)) if 0 <= cw < C
    const int ustar = (m - 1) - (n - 1) + w;      // band offset of the result cell (n-1, m-1)
    const int cstar = ustar - u0;
    const int tau_cap = (n - 1) + gl;             // macro-step in which this lane sweeps row n-1

    // wave-uniform loop bounds
    const int g_act = (two_w + 1 + C - 1) / C;
    int my_total = sweep ? (n - 1) + g_act : 0;
    int my_min = sweep ? (n - 1) : 0x7fffffff;
    int total = 0, min_rows = 0x7fffffff;
#pragma unroll
    for (int g = 0; g < PPW; ++g) {
        total = max(total, __builtin_amdgcn_readlane(my_total, g * G));
        min_rows = min(min_rows, __builtin_amdgcn_readlane(my_min, g * G));
    }
    // Column window: S = C + 1 register slots used as a ring (slot = column mod S): C live columns plus the one
    // entering at the next macro-step.  Row frames: two sets, ping-pong.  With the loop unrolled U = lcm(S, 2)
    // macro-steps every slot keeps its physical registers (no copies on the back-edge).
    constexpr int S = C + 1;
    constexpr int U = (S % 2 == 0) ? S : 2 * S;
    const int total_r = ((total + U - 1) / U) * U;
    const int a_end = min(((G + U - 1) / U) * U, total_r);
    const int b_end = min(max((min_rows / U) * U, a_end), total_r);

    float prev1[C], prev2[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { prev1[c] = APD_INF; prev2[c] = APD_INF; }
    float res1 = 0.0f, res2 = 0.0f;

    // window at macro-step 0: column j = -gl + u0 + c - w in slot c; columns <= 0 read the +INF sentinel (index m)
    float yf[S][D];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int j = u0 - gl + c - w;
        const int idx = (j >= 1) ? (min(j, m) - 1) : m;
        load_frame<D>(yf[c], rsrc, b_off + (uint32_t)idx * FB);
    }
#pragma unroll
    for (int k = 0; k < D; ++k) yf[C][k] = 0.0f;
    float xs[2][D];
#pragma unroll
    for (int k = 0; k < D; ++k) { xs[0][k] = -APD_INF; xs[1][k] = -APD_INF; }   // rows <= 0

    auto macro_steps = [&](int tau_begin, int tau_end, auto slow_tag) {
        constexpr bool SLOW = decltype(slow_tag)::value;
        for (int tau0 = tau_begin; tau0 < tau_end; tau0 += U) {
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const int tau = tau0 + q;
                const int r = q % S;               // slot of this step's first column
                const int xa = q & 1, xb = xa ^ 1; // current / next row-frame set
                const int e = (r + C) % S;         // slot of the column entering at tau + 1 (dead during this step)
                // frames entering the group at tau + 1: fetched by the two edge lanes straight into the dead
                // registers (other lanes' offsets are out of range: no memory access, and the DPP below overwrites them)
                {
                    const uint32_t xo = a_off + (uint32_t)(min(tau + 1, n) - 1) * FB;
                    const int jt = tau + 1 + (C - 1) * G - w;
                    const uint32_t yo = b_off + (uint32_t)((jt >= 1) ? (min(jt, m) - 1) : m) * FB;
                    load_frame<D>(xs[xb], rsrc, gl == 0 ? xo : kNoFrame);
                    load_frame<D>(yf[e], rsrc, gl == G - 1 ? yo : kNoFrame);
                }
                // the C local distances of this row
                float d[C];
#pragma unroll
                for (int c = 0; c < C; ++c) d[c] = frame_dist<D>(xs[xa], yf[(r + c) % S]);
                // the two DP rows
                float left1 = group_from_lower<G>(prev1[C - 1], APD_INF, gl);
                float left2 = group_from_lower<G>(prev2[C - 1], APD_INF, gl);
                float upr1 = APD_INF, upr2 = APD_INF;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float up1 = (c < C - 1) ? prev1[(c + 1) % C] : upr1;
                    const float up2 = (c < C - 1) ? prev2[(c + 1) % C] : upr2;
                    float r1 = select_node<UNIFORM_PEN>(left1, up1, prev1[c], d[c], del, ins, mat);   // left = DELETE
                    float r2 = select_node<UNIFORM_PEN>(up2, left2, prev2[c], d[c], del, ins, mat);   // up   = DELETE
                    r1 = g1[c] ? APD_INF : r1;
                    r2 = g2[c] ? APD_INF : r2;
                    prev1[c] = r1; prev2[c] = r2;
                    left1 = r1; left2 = r2;
                    if (c == 0) {
                        upr1 = group_from_upper<G>(r1, APD_INF, gl);
                        upr2 = group_from_upper<G>(r2, APD_INF, gl);
                    }
                }
                if (SLOW) {
                    if (tau == gl) {                              // this lane just swept row 0: D[0][0] = 0
#pragma unroll
                        for (int c = 0; c < C; ++c) if (c == cw) { prev1[c] = 0.0f; prev2[c] = 0.0f; }
                    }
                    if (tau == tau_cap) {                         // row n-1 done: keep cell (n-1, m-1)
#pragma unroll
                        for (int c = 0; c < C; ++c) if (c == cstar) { res1 = prev1[c]; res2 = prev2[c]; }
                    }
                }
                // advance both systolic pipelines: x one lane up, the column window one lane down
#pragma unroll
                for (int k = 0; k < D; ++k) {
                    xs[xb][k] = group_from_lower<G>(xs[xa][k], xs[xb][k], gl);
                    yf[e][k] = group_from_upper<G>(yf[(r + 1) % S][k], yf[e][k], gl);
                }
            }
        }
    };
    macro_steps(0, a_end, std::true_type{});
    macro_steps(a_end, b_end, std::false_type{});
    macro_steps(b_end, total_r, std::true_type{});

    if (P.valid && gl == (sweep ? ustar / C : 0)) {
        if (sweep) {
            const float denom = (float)(n + m);                  // alignments.rs:121
            store_pair(L, tile, P, res1 / denom, res2 / denom);
        } else {
            const float s = (P.n == 1 && P.m == 1) ? 0.0f : APD_INF;
            store_pair(L, tile, P, s, s);
        }
    }
}

template <int D, int C, int G>
static void launch_systolic_cg(const AlignLaunch &L, bool uniform, hipStream_t stream)
{
    constexpr int PPW = 64 / G;
    constexpr int WPT = kSlotsPerTile / (4 * PPW);
    const uint32_t tiles8 = (L.n_tiles + 7u) / 8u * 8u;
    const dim3 grid(tiles8 * WPT), block(256);
    if (uniform) hipLaunchKernelGGL((dtw_fused_systolic<D, C, G, true>), grid, block, 0, stream, L);
    else hipLaunchKernelGGL((dtw_fused_systolic<D, C, G, false>), grid, block, 0, stream, L);
}

template <int D>
bool launch_systolic(const AlignLaunch &L, int g, int c, bool uniform, hipStream_t stream)
{
#define APD_CASE(GG, CC) if (g == GG && c == CC) { launch_systolic_cg<D, CC, GG>(L, uniform, stream); return true; }
    APD_CASE(16, 2) APD_CASE(16, 3) APD_CASE(16, 5) APD_CASE(16, 9)
    APD_CASE(64, 3) APD_CASE(64, 5) APD_CASE(64, 9)
#undef APD_CASE
    return false;
}

}  // namespace apd

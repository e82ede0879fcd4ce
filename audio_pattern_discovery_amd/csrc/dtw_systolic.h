// Systolic fused-pair banded DTW (the production kernel), gfx950.
//
// See dtw_generic.hip for the recurrence and the DP1/DP2 fusion.  This kernel adds:
//  * G lanes per unordered pair (64/G pairs per wavefront), lane gl owns C band offsets u = C*gl + c, so that a
//    band of 2w+1 offsets fills the wave instead of leaving lanes idle (w = 66 -> G = 16, C = 9: 133/144 busy);
//  * column frames y[j] enter at lane G-1 of a group and move one lane down per macro-step (DPP row/wave shifts),
//    each lane keeping its C-column window in registers; row frames x[i] (shared by every group of the wave: the
//    pairs of a wave have the same a) are staged once per wave in a small LDS ring, filled by coalesced 16-byte
//    loads U rows ahead, and read back by every lane (row tau + 1 - gl).  Each frame is fetched from memory once
//    per wave: the algorithmic minimum 4*D*(n+m) bytes per ordered pair is an upper bound of what the kernel reads;
//  * no per-cell boundary tests: rows <= 0 use an x frame of -INF, columns <= 0 the +INF sentinel frame stored
//    behind every sequence, so those cells evaluate to +INF by arithmetic (penalties must be > 0; otherwise the
//    dispatcher takes the generic kernel); D[0][0] = 0 (alignments.rs:109) is injected and the result cell
//    (n-1, m-1) (alignments.rs:120) captured only in the first G and last G macro-steps ("slow" phases);
//  * the static band edges (DP1 stops at u = 2w-1, DP2 spans u = 1..2w) are two per-lane scalar masks per offset
//    OR-ed into the select (see select_node): no vector instruction;
//  * hybrid distance form (unit penalties): |x|^2 + |y|^2 - 2 x.y with the row frames pre-scaled by -2 when they are staged
//    (one add + D fmac per cell), ONE superset threshold test per macro-step in front of a rarely taken recompute branch, the
//    window moves issued between that test and its branch, the entering column fetched a macro-step ahead;
//  * steady-state macro-steps carry no address arithmetic (running column offset, contiguous LDS row window) and the DP rows'
//    edge exchanges are rotations (what wraps around is +INF or feeds a guarded node).
// DESIGN.md section 4.1 has the instruction budget, the measurements and what was tried and dropped.
#pragma once
#include <type_traits>

// Timing-only ablations for kernel tuning (results become wrong): build one unit with -DAPD_ABLATE=<bits>.
//   1 no sqrt   2 no DP rows   4 no frame shifts   8 no edge fetches   16 no band guards   32 software-pipelined distances (valid results)
//   64 select = min3 + d (no tie rule, no guards: what a deferred tie check could save at most)   128 no hybrid threshold test
//   512 G = 32 without the one-instruction wave shift + masked column fetch (valid results)
#ifndef APD_ABLATE
#define APD_ABLATE 0
#endif

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
template <int D, int DN>
__device__ __forceinline__ float frame_dist(const float (&x)[DN], const float (&y)[DN])
{
    float t = x[0] - y[0];
    float acc = t * t;
#pragma unroll
    for (int k = 1; k < D; ++k) {
        t = x[k] - y[k];
        acc = __builtin_fmaf(t, t, acc);
    }
    if (APD_ABLATE & 1) return acc;
    return __builtin_amdgcn_sqrtf(acc);
}

// numerics.rs:114-120 operation for operation: every difference, square and partial sum rounded on its own, then the
// correctly rounded square root -- the bits the CPU code produces.  Used with non-unit penalties, where the node update
// adds a penalty that depends on WHICH predecessor won: a near-tie resolved the other way moves the score by percents,
// so the comparison inputs must be the reference's to the last bit.
template <int D, int DN>
__device__ __forceinline__ float frame_sq_strict(const float (&x)[DN], const float (&y)[DN])
{
    float t = x[0] - y[0];
    float acc = t * t;                                            // 0.0 + t*t == t*t
#pragma unroll
    for (int k = 1; k < D; ++k) {
        t = x[k] - y[k];
        const float sq = t * t;
        acc = acc + sq;
    }
    return acc;
}
template <int D, int DN>
__device__ __forceinline__ float frame_dist_strict(const float (&x)[DN], const float (&y)[DN])
{
    return __builtin_sqrtf(frame_sq_strict<D, DN>(x, y));         // correctly rounded (hipcc default: -fhip-fp32-correctly-rounded-divide-sqrt)
}

// Squared distance by norm expansion, |x|^2 + |y|^2 - 2 x.y: x[D], y[D] hold the squared norms, and the row frame comes
// pre-scaled: xm[k] = -2 x[k] (done once per row when the wave stages it in LDS), xm[D] = |x|^2.
// |x|^2 + |y|^2 + sum xm_k y_k: one add and D v_fmac -- the multiply that opens the dot product and the closing fma(-2, ., .) are gone.
template <int D>
__device__ __forceinline__ float frame_sq_expanded_pre(const float (&xm)[D + 1], const float (&y)[D + 1])
{
    float acc = xm[D] + y[D];
#pragma unroll
    for (int k = 0; k < D; ++k) acc = __builtin_fmaf(xm[k], y[k], acc);
    return acc;
}
// One 16-byte piece (components 4 piece .. 4 piece + 3) of a row frame on its way into the LDS ring: times -2, the norm kept.
template <int D>
__device__ __forceinline__ void prescale_row_piece(apd_f32x4 &v, int piece)
{
    const bool np = piece == D / 4;                              // the piece that holds the norm
    v.x = (np && D % 4 == 0) ? v.x : -2.0f * v.x;
    v.y = (np && D % 4 == 1) ? v.y : -2.0f * v.y;
    v.z = (np && D % 4 == 2) ? v.z : -2.0f * v.z;
    v.w = (np && D % 4 == 3) ? v.w : -2.0f * v.w;
}
// Difference form from the pre-scaled row frame: fma(-0.5, xm_k, -y_k) is x_k - y_k with its one rounding (the scaling is exact).
template <int D>
__device__ __forceinline__ float frame_sq_exact_pre(const float (&xm)[D + 1], const float (&y)[D + 1])
{
    float t = __builtin_fmaf(-0.5f, xm[0], -y[0]);
    float acc = t * t;
#pragma unroll
    for (int k = 1; k < D; ++k) {
        t = __builtin_fmaf(-0.5f, xm[k], -y[k]);
        acc = __builtin_fmaf(t, t, acc);
    }
    return acc;
}
template <int D, int C, int G, bool UNIFORM_PEN, bool HYBRID>
__global__ __launch_bounds__(256) void dtw_fused_systolic(const AlignLaunch L)
{
    static_assert(C >= 2 && (64 % G) == 0, "bad geometry");
    if (L.d_nonfinite != nullptr && *L.d_nonfinite != 0u) return;   // a NaN / infinite feature in the batch: the literal kernel's job
    constexpr int DN = D + 1;                                  // frame components + squared norm
    constexpr int DP = (DN + 3) & ~3;                          // floats per resident frame
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
    float ins = L.band.ins, del = L.band.del, mat = L.band.mat;
    float tau_thr = L.tau, nmax_b = P.nmax_b;
    asm volatile("" : "+v"(ins), "+v"(del), "+v"(mat), "+v"(tau_thr), "+v"(nmax_b));   // keep them in VGPRs: an SGPR operand doubles a VOP2's issue cost
    // row-frame ring of this wave: slot = row & (R - 1); rows <= 0 hold -INF.  R > 2U + G - 2 keeps a refill from
    // overwriting a row some lane still needs.
    constexpr int R = (G == 64) ? 128 : 64;                      // G = 32, U <= 10: 2U + G - 2 = 50 < 64
    constexpr int LPF = DP / 4;                                  // lanes (16-byte pieces) per frame
    constexpr int FPF = 64 / LPF;                                // frames per wave-wide fill
    // ring rows are RS = DP + 4 floats apart: the 16 lanes of a ds_read_b128 group read 16 consecutive rows, and a stride of
    // 16 floats would put every fourth of them on the same banks (4-way conflict); 20 floats spreads them over all 64
    constexpr int RS = DP + 4;
    // S = C + 1 and U (rows per refill) are fixed further down; the ring carries a copy of its first U rows behind row R - 1, so
    // that the U consecutive rows a lane reads during one unrolled block are contiguous: one address per block, the rest are
    // immediate offsets of the ds_read
    constexpr int U_ROWS = ((C + 1) % 2 == 0) ? (C + 1) : 2 * (C + 1);
    __shared__ float xring_all[4][(R + U_ROWS) * RS];
    float *const xring = xring_all[threadIdx.x >> 6];
    // sequence a (and its length) is the same for every sweeping group of the wave
    const int lead = __builtin_ctzll(__ballot(sweep));
    const uint32_t a_off_w = __builtin_amdgcn_readlane(a_off, lead);
    const int n_w = __builtin_amdgcn_readlane(n, lead);
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
    static_assert(U == U_ROWS, "ring copy sized for another unroll");
    // G = 32 (two DPP rows per group), hybrid form: a shift inside the group takes two DPP instructions per register (the rows
    // that read across a row boundary, and the rows that end a group and must keep the fetched frame).  Instead the whole wave
    // shifts by one lane with ONE instruction -- the two group-top lanes receive a frame of the neighbouring pair or keep a
    // stale one -- and only those lanes then fetch their entering column, under EXEC, on top of it.
    // G = 8 (two groups per DPP row) is the same case one level down: row_shl instead of wave_shl.
    constexpr bool MASKED_FETCH = HYBRID && (G == 32 || G == 8) && !(APD_ABLATE & 512);
    const int total_r = ((total + U - 1) / U) * U;
    // Strict mode (unit penalties, difference form): the steady-state loop takes square roots with sqrt_rn_finite, whose domain
    // excludes +INF, so every macro-step in which some lane still holds a sentinel column (j <= 0: tau <= w - (C-1) gl, at most w)
    // or a row <= 0 (tau < G) stays in the slow phase, whose square root is the compiler's general sequence.
    int a_steps = G;
    if (UNIFORM_PEN && !HYBRID) {
        int w_max = 0;
#pragma unroll
        for (int g = 0; g < PPW; ++g) w_max = max(w_max, __builtin_amdgcn_readlane(w, g * G));
        a_steps = max(G, w_max + 1);
    }
    const int a_end = min(((a_steps + U - 1) / U) * U, total_r);
    const int b_end = min(max((min_rows / U) * U, a_end), total_r);

    float prev1[C], prev2[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { prev1[c] = APD_INF; prev2[c] = APD_INF; }
    float res1 = 0.0f, res2 = 0.0f;

    // window at macro-step 0: column j = -gl + u0 + c - w in slot c; columns <= 0 read the +INF sentinel (index m)
    float yf[S][DN];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int j = u0 - gl + c - w;
        const int idx = (j >= 1) ? (min(j, m) - 1) : (HYBRID ? m : m + 1);   // sentinel H / sentinel E
        load_frame<DN>(yf[c], rsrc, b_off + (uint32_t)idx * FB);
    }
#pragma unroll
    for (int k = 0; k < DN; ++k) yf[C][k] = 0.0f;
    if (HYBRID && !MASKED_FETCH && !(APD_ABLATE & 8)) {          // hybrid form: columns are fetched one macro-step ahead (fetch_column)
        const int jt = 1 + (C - 1) * G - w;
        const uint32_t yo = b_off + (uint32_t)((jt >= 1) ? (min(jt, m) - 1) : m) * FB;
        load_frame<DN>(yf[C], rsrc, gl == G - 1 ? yo : kNoFrame);
    }
    // ring prologue: rows -(G-1)..0 are -INF, rows 1..U come from memory
    constexpr int NFILL = (U + FPF - 1) / FPF;
    const int fill_f = lane / LPF, fill_q = lane % LPF;          // this lane's frame and 16-byte piece inside a fill
    auto fill_load = [&](int first_row, apd_f32x4 (&regs)[NFILL]) __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < NFILL; ++f) {
            const int fi = f * FPF + fill_f;
            const bool act = (fill_f < FPF) & (fi < U);
            const uint32_t off = a_off_w + (uint32_t)(min(first_row + fi, n_w) - 1) * FB + 16u * fill_q;
            regs[f] = __builtin_bit_cast(apd_f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, act ? off : kNoFrame, 0, 0));
        }
    };
    auto fill_store = [&](int first_row, const apd_f32x4 (&regs)[NFILL]) __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < NFILL; ++f) {
            const int fi = f * FPF + fill_f;
            if ((fill_f < FPF) & (fi < U)) {
                apd_f32x4 v = regs[f];
                if (HYBRID) prescale_row_piece<D>(v, fill_q);    // stage -2 x (the norm slot stays): see frame_sq_expanded_pre
                const int slot = (first_row + fi) & (R - 1);
                *reinterpret_cast<apd_f32x4 *>(&xring[slot * RS + 4 * fill_q]) = v;
                if (slot < U) *reinterpret_cast<apd_f32x4 *>(&xring[(slot + R) * RS + 4 * fill_q]) = v;   // the copy behind the ring
            }
        }
    };
    for (int e = lane; e < G * DP; e += 64) {                    // rows <= 0: -INF components, or (hybrid) zeros with norm +INF
        const int slot = (-(e / DP)) & (R - 1);
        const float v = HYBRID ? ((e % DP) == D ? APD_INF : 0.0f) : -APD_INF;
        xring[slot * RS + (e % DP)] = v;
        if (slot < U) xring[(slot + R) * RS + (e % DP)] = v;
    }
    {
        apd_f32x4 regs[NFILL];
        fill_load(1, regs);
        fill_store(1, regs);
    }
    asm volatile("" ::: "memory");
    auto read_row = [&](float (&dst)[DN], int row) __attribute__((always_inline)) {
        const float *p = &xring[(row & (R - 1)) * RS];
#pragma unroll
        for (int q = 0; q < LPF; ++q) {
            const apd_f32x4 t = *reinterpret_cast<const apd_f32x4 *>(p + 4 * q);
            if (4 * q + 0 < DN) dst[4 * q + 0] = t.x;
            if (4 * q + 1 < DN) dst[4 * q + 1] = t.y;
            if (4 * q + 2 < DN) dst[4 * q + 2] = t.z;
            if (4 * q + 3 < DN) dst[4 * q + 3] = t.w;
        }
    };
    auto read_row_at = [&](float (&dst)[DN], const float *p) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < LPF; ++q) {
            const apd_f32x4 t = *reinterpret_cast<const apd_f32x4 *>(p + 4 * q);
            if (4 * q + 0 < DN) dst[4 * q + 0] = t.x;
            if (4 * q + 1 < DN) dst[4 * q + 1] = t.y;
            if (4 * q + 2 < DN) dst[4 * q + 2] = t.z;
            if (4 * q + 3 < DN) dst[4 * q + 3] = t.w;
        }
    };
    constexpr int NX = HYBRID ? 1 : 2;                           // hybrid: one row-frame set (registers are the limit)
    float xs[NX][DN];
    read_row(xs[0], 0 - gl);                                     // macro-step 0: row -gl (all -INF)
#pragma unroll
    for (int k = 0; k < DN; ++k) xs[NX - 1][k] = (NX == 1) ? xs[0][k] : 0.0f;
    // Software pipeline: the first C-1 distances of macro-step tau+1 only need columns the lane already holds, so they
    // are computed during macro-step tau, next to the serial DP chain; only the last column waits for the shift.
    constexpr bool PIPE = (APD_ABLATE & 32) != 0;                  // measured: no gain on MI355X (cfg 3: -2 %), kept as a tuning switch
    float dn[C];
#pragma unroll
    for (int c = 0; c < C; ++c) dn[c] = (PIPE && !HYBRID && c < C - 1) ? frame_dist_strict<D, DN>(xs[0], yf[c]) : 0.0f;

    // advance the column window: every column moves one lane down (q: position inside the unrolled block)
    auto advance_window = [&](int q) __attribute__((always_inline)) {
        const int r = q % S, e = (r + C) % S;
#pragma unroll
        for (int k = 0; k < (HYBRID ? DN : D); ++k) {
            if (APD_ABLATE & 4) { yf[e][k] = yf[(r + 1) % S][k] + yf[e][k]; continue; }
            yf[e][k] = MASKED_FETCH ? dpp_move<(G == 8) ? 0x101 : 0x130>(yf[(r + 1) % S][k], yf[e][k])   // row_shl:1 / wave_shl:1, every row
                                    : group_from_upper<G>(yf[(r + 1) % S][k], yf[e][k], gl);
        }
    };
    auto macro_steps = [&](int tau_begin, int tau_end, auto slow_tag) __attribute__((always_inline)) {
        constexpr bool SLOW = decltype(slow_tag)::value;
        // Steady state (tau >= G, so the entering column jt = tau + 1 + (C-1) G - w is >= 1: G C >= 2w + 1): the top lane's fetch
        // offset advances by one frame per macro-step, clamped at column m; the other lanes stay out of range.
        uint32_t yo_cur = kNoFrame, yo_cap = kNoFrame, yo_step = 0u;
        if (!SLOW && gl == G - 1) {                              // (hybrid form: fetched one macro-step ahead, see below)
            yo_cap = b_off + (uint32_t)(m - 1) * FB;
            yo_cur = min(b_off + (uint32_t)(tau_begin + ((HYBRID && !MASKED_FETCH) ? 1 : 0) + (C - 1) * G - w) * FB, yo_cap);
            yo_step = FB;
        }
        // the column entering the group at step t + 1, fetched by the top lane
        auto fetch_column = [&](float (&dst)[DN], int t) __attribute__((always_inline)) {
            if (APD_ABLATE & 8) return;
            if (SLOW) {
                const int jt = t + 1 + (C - 1) * G - w;
                const uint32_t yo = b_off + (uint32_t)((jt >= 1) ? (min(jt, m) - 1) : (HYBRID ? m : m + 1)) * FB;
                load_frame<DN>(dst, rsrc, gl == G - 1 ? yo : kNoFrame);
            } else {
                load_frame<DN>(dst, rsrc, yo_cur);
                yo_cur = min(yo_cur + yo_step, yo_cap);
            }
        };
        for (int tau0 = tau_begin; tau0 < tau_end; tau0 += U) {
            apd_f32x4 fill_regs[NFILL];
            fill_load(tau0 + U + 1, fill_regs);                  // rows of the NEXT block, stored at this block's end
            const float *const xrows = &xring[((tau0 + 1 - gl) & (R - 1)) * RS];   // rows tau0 + 1 - gl ... + U - 1, contiguous
#pragma unroll
            for (int q = 0; q < U; ++q) {
                const int tau = tau0 + q;
                const int r = q % S;               // slot of this step's first column
                const int xa = HYBRID ? 0 : (q & 1), xb = HYBRID ? 0 : (xa ^ 1);   // current / next row-frame set
                const int e = (r + C) % S;         // slot of the column entering at tau + 1 (dead during this step)
                // column frame entering the group at tau + 1: fetched by the top lane straight into the dead slot
                // (other lanes' offsets are out of range: no memory access, and the DPP below overwrites them)
                // (hybrid form: the window moves in the middle of the step, so the frame is fetched a step ahead, see fetch_column)
                if (!HYBRID) {
                    fetch_column(yf[e], tau);
                    read_row_at(xs[xb], xrows + q * RS);         // next row frame from the wave's LDS ring
                }
                // the C local distances of this row
                float d[C];
                if (HYBRID) {
                    // |x|^2 + |y|^2 - 2 x.y; where that is below tau (|x|^2 + |y|^2) cancellation has eaten the digits:
                    // such cells are recomputed in the difference form (wave-uniform branch, rare for unrelated frames)
                    // One test per macro-step instead of one per cell: if some cell has d2_c < tau (|x|^2 + |y_c|^2) then the
                    // smallest d2 of the lane is below tau (|x|^2 + N_b), N_b the largest frame norm of sequence b -- a superset
                    // of the per-cell condition (never misses a cell; on a hit the exact per-cell test below decides).
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        d[c] = frame_sq_expanded_pre<D>(xs[xa], yf[(r + c) % S]);
                    }
                    float dmin = d[0];
#pragma unroll
                    for (int c = 1; c + 1 < C; c += 2) dmin = __builtin_fminf(__builtin_fminf(dmin, d[c]), d[c + 1]);
                    if (C % 2 == 0) dmin = __builtin_fminf(dmin, d[C - 1]);
                    const bool any = (APD_ABLATE & 128) ? false : dmin < (xs[xa][D] + nmax_b) * tau_thr;
                    // The window moves here, between the test and the branch on it: the slot it writes is dead, nothing below
                    // reads the one it vacates before the next step, and the 14 moves cover the latency of compare -> branch.
                    advance_window(q);
                    if (MASKED_FETCH) {
                        if (gl == G - 1) fetch_column(yf[e], tau);   // the column entering at tau + 1, on top of what the shift left there
                    }
                    if (__builtin_expect(__ballot(any) != 0ull, 0)) {   // rare: placed out of the hot instruction stream
                        float nx = xs[xa][D];
                        asm volatile("" : "+v"(nx));              // a norm sum of its own: reusing the hot path's would keep nine of them
                                                                 // alive across the branch and turn their v_fmac into 3-operand v_fma
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            const float sc = nx + yf[(r + c) % S][D];
                            const float ex = frame_sq_exact_pre<D>(xs[xa], yf[(r + c) % S]);
                            d[c] = (d[c] < sc * tau_thr) ? ex : d[c];
                        }
                    }
                    // slot r (this step's first column) is dead from here on, and it is the slot the NEXT step's window move fills:
                    // fetch that step's entering column now, a whole macro-step before it is needed
                    if (!MASKED_FETCH) fetch_column(yf[r], tau + 1);
#pragma unroll
                    for (int c = 0; c < C; ++c) d[c] = (APD_ABLATE & 1) ? d[c] : __builtin_amdgcn_sqrtf(d[c]);
                    read_row_at(xs[0], xrows + q * RS);           // the row frame is dead now: fetch the next one under the DP rows
                } else if (UNIFORM_PEN && !SLOW && !PIPE) {
                    // Strict mode, steady state.  The difference form of the band kernels IS the reference's arithmetic
                    // (numerics.rs:114-120 operation for operation); the square root is sqrt_rn_finite (dtw_common.h): v_sqrt_f32 and
                    // an exact two-sided fix-up, 7 vector ops instead of the 16 of the compiler's general sequence.  Its domain is
                    // 2^-96 <= d2 < +INF: the steady state sees no sentinel frame (strict_a_end below keeps those macro-steps in the
                    // slow phase) and no overflow (features >= 2^60 flag the batch for the literal kernel), and a macro-step in
                    // which some cell's d2 is below 2^-96 (zero included: identical frames) takes the general sequence instead,
                    // behind one wave-uniform branch.
                    float d2[C];
#pragma unroll
                    for (int c = 0; c < C; ++c) d2[c] = frame_sq_strict<D, DN>(xs[xa], yf[(r + c) % S]);
                    float dmin = d2[0];
#pragma unroll
                    for (int c = 1; c + 1 < C; c += 2) dmin = __builtin_fminf(__builtin_fminf(dmin, d2[c]), d2[c + 1]);
                    if (C % 2 == 0) dmin = __builtin_fminf(dmin, d2[C - 1]);
                    if (__builtin_expect(__ballot(!(dmin >= 0x1p-96f)) != 0ull, 0)) {
#pragma unroll
                        for (int c = 0; c < C; ++c) d[c] = __builtin_sqrtf(d2[c]);
                    } else {
#pragma unroll
                        for (int c = 0; c < C; ++c) d[c] = sqrt_rn_finite(d2[c]);
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < C; ++c)
                        // The difference form of the band kernels IS the reference's arithmetic (numerics.rs:114-120 operation for
                        // operation), with either select: with unit penalties `base + 1.0 * d` is `base + d` and the fast select
                        // picks the reference's predecessor for every non-NaN input, so <.., UNIFORM_PEN = true, HYBRID = false> is
                        // bit-identical to the CPU code at 4 instead of ~12 vector ops per node (strict mode, apd_set_distance_mode 2).
                        d[c] = (UNIFORM_PEN && PIPE && c < C - 1) ? dn[c] : frame_dist_strict<D, DN>(xs[xa], yf[(r + c) % S]);
                    if (PIPE) {
#pragma unroll
                        for (int c = 0; c < C - 1; ++c) dn[c] = frame_dist_strict<D, DN>(xs[xb], yf[(r + 1 + c) % S]);
                    }
                }
                // the two DP rows
                // Exchanges across lanes wrap around the group (rotations: no +INF fill to materialise).  What the edge lanes
                // receive is either +INF already or never looked at: lane 0's DELETE/INSERT neighbour comes from offset
                // G*C - 1 >= 2w, outside DP1's band and therefore +INF (guarded nodes sit on their +INF MATCH predecessor), and
                // its DP2 node (u = 0) is guarded itself; lane G-1's last node is guarded in DP1, and in DP2 either guarded
                // (G*C - 1 > 2w) or handed lane 0's guarded u = 0 node, +INF.  (G = 8, 32: the rotation spans two groups / the wave; the nodes
                // that wrap in come from the neighbouring group's edges, which meet the same conditions -- G*C >= 2w + 1 holds
                // for every pair of the tile, and idle groups sweep a 1 x 1 dummy with w = 2.)
                float left1 = group_from_lower_wrap<G>(prev1[C - 1], APD_INF, gl);
                float left2 = group_from_lower_wrap<G>(prev2[C - 1], APD_INF, gl);
                float upr1 = APD_INF, upr2 = APD_INF;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    if (APD_ABLATE & 2) { prev1[c] += d[c]; prev2[c] -= d[c]; continue; }
                    const float up1 = (c < C - 1) ? prev1[(c + 1) % C] : upr1;
                    const float up2 = (c < C - 1) ? prev2[(c + 1) % C] : upr2;
                    // guards: outside a DP's band the node is forced onto its MATCH predecessor, which is +INF
                    const float r1 = select_node<UNIFORM_PEN>(left1, up1, prev1[c], d[c], del, ins, mat, (APD_ABLATE & 16) ? false : g1[c]);   // left = DELETE
                    const float r2 = select_node<UNIFORM_PEN>(up2, left2, prev2[c], d[c], del, ins, mat, (APD_ABLATE & 16) ? false : g2[c]);   // up   = DELETE
                    prev1[c] = r1; prev2[c] = r2;
                    left1 = r1; left2 = r2;
                    if (c == 0) {
                        upr1 = group_from_upper_wrap<G>(r1, APD_INF, gl);
                        upr2 = group_from_upper_wrap<G>(r2, APD_INF, gl);
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
                if (!HYBRID) advance_window(q);
            }
            fill_store(tau0 + U + 1, fill_regs);
            asm volatile("" ::: "memory");                       // rows written by other lanes are read in the next block: no
                                                                 // barrier needed inside one wave (LDS runs in order), but the
                                                                 // compiler must not move the reads up
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

// The three instantiations of one (D, C, G) live in two translation units -- dtw_sys_d<D>.hip: hybrid form and the literal select
// (non-unit penalties); dtw_sysx_d<D>.hip (APD_SYSTOLIC_STRICT_UNIT): strict mode's <.., true, false> -- so that each is
// compiled once, in parallel, with the scheduler flags measured for its own instruction mix (csrc/Makefile).
template <int D, int C, int G, bool UNIFORM_PEN, bool HYBRID>
static void launch_systolic_kernel(const AlignLaunch &L, hipStream_t stream)
{
    constexpr int PPW = 64 / G;
    constexpr int WPT = kSlotsPerTile / (4 * PPW);
    const uint32_t tiles8 = (L.n_tiles + 7u) / 8u * 8u;
    hipLaunchKernelGGL((dtw_fused_systolic<D, C, G, UNIFORM_PEN, HYBRID>), dim3(tiles8 * WPT), dim3(256), 0, stream, L);
}

// C = 9 keeps 10 column frames per lane in registers: only for D <= 13 (max_cells_per_lane)
#define APD_SYSTOLIC_GEOMETRIES(X) \
    X(8, 5) X(8, 7) X(8, 9) X(16, 2) X(16, 3) X(16, 5) X(16, 7) X(16, 9) X(32, 5) X(32, 7) X(32, 9) X(64, 3) X(64, 5) X(64, 7) X(64, 9)

#ifdef APD_SYSTOLIC_STRICT_UNIT
template <int D>
bool launch_systolic_strict(const AlignLaunch &L, int g, int c, hipStream_t stream)
{
#define APD_CASE(GG, CC) if constexpr (CC <= max_cells_per_lane(D)) { if (g == GG && c == CC) { launch_systolic_kernel<D, CC, GG, true, false>(L, stream); return true; } }
    APD_SYSTOLIC_GEOMETRIES(APD_CASE)
#undef APD_CASE
    return false;
}
#else
template <int D>
bool launch_systolic(const AlignLaunch &L, int g, int c, bool unit, hipStream_t stream)
{
    // unit penalties: the fast select, either distance form; anything else: literal select on strict (bit-faithful) distances
    if (unit && !L.hybrid) return launch_systolic_strict<D>(L, g, c, stream);
#define APD_CASE(GG, CC) if constexpr (CC <= max_cells_per_lane(D)) { if (g == GG && c == CC) { \
        if (!unit) launch_systolic_kernel<D, CC, GG, false, false>(L, stream); else launch_systolic_kernel<D, CC, GG, true, true>(L, stream); \
        return true; } }
    APD_SYSTOLIC_GEOMETRIES(APD_CASE)
#undef APD_CASE
    return false;
}
#endif

}  // namespace apd

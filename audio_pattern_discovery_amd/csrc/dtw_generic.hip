// Generic fused-pair banded DTW + layout/unpack helpers + kernel dispatch (gfx950).
//
// Replaces the per-pair body of AlignmentWorkers::align_all (reference src/alignments.rs:50-58):
// Alignment::new + construct_alignment (:165-180) + alignment_score (:129-160) + score (:116-125).
//
// Work unit: one UNORDERED pair (a, b), a < b.  One sweep computes BOTH ordered scores
//   DP1 = score(x = a, y = b)  and  DP2 = score(x = b, y = a),
// because the two recurrences visit (almost) the same cells and share every local distance
// euclidean(A[i], B[j]) (numerics.rs:114-120).  In A-row / B-column coordinates (i over A, j over B, band
// offset o = j - i, u = o + w):
//   DP1: u in [0, 2w-1], left neighbour (i, j-1) is the DELETE branch, up (i-1, j) the INSERT branch;
//   DP2: u in [1, 2w],   up   neighbour is the DELETE branch, left the INSERT branch
// (DP2's cell (j, i) of the swapped problem is our cell (i, j); the reference band j'-i' in [-w, w-1] of
// alignments.rs:175 becomes o in [-w+1, w]).  Each DP is exactly the reference recurrence for its ordered pair.
//
// Lane mapping: a lane owns C consecutive offsets u = C*l + c.  Macro-step tau: every lane processes row
// i = tau - l, its C cells left to right.  Dependences across lanes:
//   (i, u-1) of the first cell  = last cell of lane l-1 from the PREVIOUS macro-step  -> DPP shift up
//   (i-1, u+1) of the last cell = first cell of lane l+1 from THIS macro-step          -> DPP shift down
// so C >= 2 and the min(INS, DEL, MATCH) dependency never goes through memory.
// Only rows 1..n-1 and columns 1..m-1 are swept: score() reads cell (n-1, m-1) (alignments.rs:120); row n and
// column m never influence it.
//
// The kernel in this file is the slow, fully general one: any frame dimension, any band up to what 160 KB of LDS
// holds (C chosen at run time, per-lane DP rows in LDS, every boundary tested per cell, any penalty values).
// dtw_systolic.h holds the production kernel.
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include "dtw_common.h"

namespace apd {

__device__ __forceinline__ void generic_pair(const AlignLaunch &L, int c_max, uint32_t wave, float *lds)
{
    const int lane = threadIdx.x;
    const uint32_t tile = wave / kSlotsPerTile, slot = wave % kSlotsPerTile;
    const PairInfo P = decode_pair(L, tile, slot);
    if (!P.valid) return;
    const int n = P.n, m = P.m, w = P.w;
    if (n == 1 || m == 1) {                                   // alignments.rs:116-125 with an absent cell
        if (lane == 0) { const float s = (n == 1 && m == 1) ? 0.0f : APD_INF; store_pair(L, tile, P, s, s); }
        return;
    }
    const float ins = L.band.ins, del = L.band.del, mat = L.band.mat;
    const int two_w = 2 * w;
    int C = (two_w + 1 + 63) / 64;
    C = max(C, 2);
    float *p1 = lds, *p2 = lds + c_max * 64;               // [c][lane]
    for (int c = 0; c < C; ++c) { p1[c * 64 + lane] = APD_INF; p2[c * 64 + lane] = APD_INF; }
    const int dp4 = (int)L.dpad / 4, dim = (int)L.dim;
    const int u0 = C * lane;
    const int g_act = (two_w + 1 + C - 1) / C;
    const int total = (n - 1) + g_act;
    float res1 = 0.0f, res2 = 0.0f;
    float last1 = APD_INF, last2 = APD_INF;
    for (int tau = 0; tau < total; ++tau) {
        const int i = tau - lane;
        const int jb = i + u0 - w;
        const float4 *xa = reinterpret_cast<const float4 *>(P.A + (uint64_t)(min(max(i, 1), n) - 1) * L.dpad);
        float left1 = from_lower_lane(last1, APD_INF);
        float left2 = from_lower_lane(last2, APD_INF);
        float upr1 = APD_INF, upr2 = APD_INF;
        float nxt1 = p1[lane], nxt2 = p2[lane];             // prev[c] for c = 0
        for (int c = 0; c < C; ++c) {
            const int j = jb + c, u = u0 + c;
            const float4 *yb = reinterpret_cast<const float4 *>(P.B + (uint64_t)(min(max(j, 1), m) - 1) * L.dpad);
            // numerics.rs:114-120 operation for operation (difference, square, sum each rounded; correctly rounded sqrt):
            // this kernel serves every penalty setting, and with unequal penalties near-ties must resolve as on the CPU
            float acc = 0.0f;
            for (int q = 0; q < dp4; ++q) {                  // slots >= dim hold the squared norm / padding: skipped
                const float4 xv = xa[q], yv = yb[q];
                const int k0 = 4 * q;
                float t = xv.x - yv.x, sq = t * t;
                acc = (k0 < dim) ? acc + sq : acc;
                t = xv.y - yv.y; sq = t * t; acc = (k0 + 1 < dim) ? acc + sq : acc;
                t = xv.z - yv.z; sq = t * t; acc = (k0 + 2 < dim) ? acc + sq : acc;
                t = xv.w - yv.w; sq = t * t; acc = (k0 + 3 < dim) ? acc + sq : acc;
            }
            const float d = __builtin_sqrtf(acc);
            const float m1 = nxt1, m2 = nxt2;
            float up1, up2;
            if (c < C - 1) { up1 = p1[(c + 1) * 64 + lane]; up2 = p2[(c + 1) * 64 + lane]; }
            else { up1 = upr1; up2 = upr2; }
            nxt1 = up1; nxt2 = up2;                          // prev[c+1] is the next cell's MATCH predecessor
            float r1 = select_node<false>(left1, up1, m1, d, del, ins, mat);   // DP1: left = DELETE, up = INSERT
            float r2 = select_node<false>(up2, left2, m2, d, del, ins, mat);   // DP2: up = DELETE, left = INSERT
            const bool inside = (i >= 1) & (j >= 1);
            const float inv = ((i == 0) & (j == 0)) ? 0.0f : APD_INF;          // D[0][0] = 0 (alignments.rs:109)
            r1 = (inside & (u <= two_w - 1)) ? r1 : inv;
            r2 = (inside & (u >= 1) & (u <= two_w)) ? r2 : inv;
            if ((i == n - 1) & (j == m - 1)) { res1 = r1; res2 = r2; }
            p1[c * 64 + lane] = r1; p2[c * 64 + lane] = r2;
            left1 = r1; left2 = r2;
            if (c == 0) { upr1 = from_upper_lane(r1, APD_INF); upr2 = from_upper_lane(r2, APD_INF); }
        }
        last1 = left1; last2 = left2;
    }
    const int ustar = (m - 1) - (n - 1) + w;
    if (lane == ustar / C) {
        const float denom = (float)(n + m);                  // alignments.rs:121
        store_pair(L, tile, P, res1 / denom, res2 / denom);
    }
}

// one wavefront per pair slot
__global__ __launch_bounds__(64) void dtw_fused_generic(const AlignLaunch L, int c_max)
{
    extern __shared__ float lds[];
    generic_pair(L, c_max, blockIdx.x, lds);
}

// The same as the fallback behind the fast kernels: a small persistent grid over all pair slots of the launch that returns at
// once unless the batch's non-finite flag is raised (then the fast kernels have returned at once, and this one does the work).
__global__ __launch_bounds__(64) void dtw_fused_generic_fallback(const AlignLaunch L, int c_max, uint32_t total_waves)
{
    extern __shared__ float lds[];
    if (*L.d_nonfinite == 0u) return;
    for (uint32_t wave = blockIdx.x; wave < total_waves; wave += gridDim.x) {
        generic_pair(L, c_max, wave, lds);
        __syncthreads();                                     // the DP rows in LDS are reused by the next pair
    }
}

// ------------------------------------------------------------------------------------------------
// Resident layout: sequences in length order (apd_length_order), per sequence [frames | sentinel H | sentinel E],
// seq_off[p] = resident_offsets[p] + 2 p; src_off[p] = first frame of that sequence in the caller's array.  A frame is dpad =
// ceil4(dim + 1) floats: the dim components, then the squared norm sum_k x_k^2 (slot dim), then zeros.  Sentinel H has
// zero components and norm +INF (hybrid distances), sentinel E has +INF components (difference-form distances).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pad_frames_kernel(const float *__restrict__ src, float *__restrict__ dst, const uint32_t *__restrict__ seq_off,
                                  const uint32_t *__restrict__ src_off, uint32_t n_seq, uint64_t n_frames_padded, uint32_t src_dim,
                                  uint32_t dim, uint32_t dpad, uint32_t *__restrict__ flags, float *__restrict__ seq_nmax)
{
    // One thread per 16-byte piece of a resident frame (dpad / 4 pieces per frame, dpad / 4 consecutive lanes -- a power of
    // two up to 8 for the instantiated dimensions, any count otherwise): one search for the frame's sequence per piece,
    // contiguous reads of the source frame, one aligned 16-byte store; the squared norm is summed in f64 over the lanes of
    // the frame.  (The first version ran one thread, one search and one 4-byte store per FLOAT: 0.48 TB/s.)
    const uint32_t ppf = dpad / 4;                                // pieces per frame
    const uint64_t total = n_frames_padded * ppf;
    bool nonfinite = false;
    const bool shuffle_ok = (ppf & (ppf - 1)) == 0 && ppf <= 64;  // the lanes of a frame sit in one wavefront, aligned
    for (uint64_t e0 = (uint64_t)blockIdx.x * blockDim.x; e0 < total; e0 += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t e = e0 + threadIdx.x;
        const bool live = e < total;
        const uint32_t f = live ? (uint32_t)(e / ppf) : 0u, q = live ? (uint32_t)(e - (uint64_t)f * ppf) : 0u;
        // largest s with seq_off[s] <= f: one search per wavefront (for its first frame), then a step or two forward per lane
        const uint32_t f_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)f);
        uint32_t lo = 0, hi = n_seq;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (seq_off[mid] <= f_first) lo = mid; else hi = mid; }
        while (live && lo + 1 < n_seq && seq_off[lo + 1] <= f) ++lo;
        const bool sent_e = live && (f + 1 == seq_off[lo + 1]), sent_h = live && (f + 2 == seq_off[lo + 1]);
        const bool real = live && !sent_e && !sent_h;
        float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        double part = 0.0;                                        // this piece's share of sum_k x_k^2
        if (real) {
            const float *fr = src + (uint64_t)(src_off[lo] + (f - seq_off[lo])) * src_dim;
#pragma unroll
            for (uint32_t i = 0; i < 4; ++i) {
                const uint32_t k = 4 * q + i;
                if (k < src_dim) {                                // components src_dim .. dim - 1 stay zero
                    v[i] = fr[k];
                    nonfinite |= !(__builtin_fabsf(v[i]) < kFeatureBound);   // NaN, +-INF, or so large that a squared distance could overflow
                    part += (double)v[i] * (double)v[i];
                }
            }
        }
        double norm = part;
        if (ppf == 4) {                                           // D <= 14: the four pieces of a frame are one DPP quad
            auto quad_swap = [](double x, auto ctrl) __attribute__((always_inline)) {
                constexpr int CTRL = decltype(ctrl)::value;
                const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
                const unsigned int lo32 = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)b, CTRL, 0xf, 0xf, false);
                const unsigned int hi32 = (unsigned int)__builtin_amdgcn_update_dpp(0, (int)(unsigned int)(b >> 32), CTRL, 0xf, 0xf, false);
                return __builtin_bit_cast(double, ((unsigned long long)hi32 << 32) | lo32);
            };
            norm += quad_swap(norm, std::integral_constant<int, 0xB1>{});   // quad_perm [1,0,3,2]
            norm += quad_swap(norm, std::integral_constant<int, 0x4E>{});   // quad_perm [2,3,0,1]
        } else if (shuffle_ok) {
            for (uint32_t o = 1; o < ppf; o <<= 1) norm += __shfl_xor(norm, (int)o);
        } else if (real) {                                        // generic dimension: the piece holding slot `dim` re-reads the frame
            norm = 0.0;
            if (dim / 4 == q) {
                const float *fr = src + (uint64_t)(src_off[lo] + (f - seq_off[lo])) * src_dim;
                for (uint32_t t = 0; t < src_dim; ++t) norm += (double)fr[t] * (double)fr[t];
            }
        }
#pragma unroll
        for (uint32_t i = 0; i < 4; ++i) {
            const uint32_t k = 4 * q + i;
            if (sent_e) v[i] = (k <= dim) ? APD_INF : 0.0f;
            else if (sent_h) v[i] = (k == dim) ? APD_INF : 0.0f;
            else if (real && k == dim) v[i] = (float)norm;
        }
        {
            // largest norm of the sequence: non-negative floats order like their bit patterns (a NaN norm flags the batch anyway).
            // The frames of a wavefront almost always belong to one sequence: one atomic per wavefront then, not one per frame.
            const float mine = (real && dim / 4 == q && norm == norm) ? (float)norm : 0.0f;
            const bool has = real && dim / 4 == q;
            const uint32_t lo_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
            const unsigned long long any_has = __ballot(has);
            if (__ballot(has && lo != lo_first) == 0ull) {
                float m = mine;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
                if ((threadIdx.x & 63) == 0 && any_has != 0ull) atomicMax(reinterpret_cast<unsigned int *>(&seq_nmax[lo_first]), __builtin_bit_cast(unsigned int, m));
            } else if (has) atomicMax(reinterpret_cast<unsigned int *>(&seq_nmax[lo]), __builtin_bit_cast(unsigned int, mine));
        }
        if (live) *reinterpret_cast<float4 *>(dst + e * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
    // the fast kernels assume finite features (fminf-based select, +INF sentinels, norm expansion) and finite squared distances
    // (sqrt_rn_finite): a batch with a NaN, an infinity or a feature of magnitude >= 2^60 anywhere is routed to the literal kernel,
    // where NaN compares false and takes MATCH as in alignments.rs:153-159 and an overflowing distance is +INF as on the CPU
    if (__ballot(nonfinite) != 0ull && (threadIdx.x & 63) == 0) atomicOr(flags, 1u);
}

// gathered: `world` slabs of slab_floats each; slab r holds tiles r, r+world, ... in order.
__global__ void unpack_tiles_kernel(const float *__restrict__ gathered, float *__restrict__ out,
                                    const uint32_t *__restrict__ order, uint32_t n_seq, uint32_t world, uint64_t slab_floats, uint32_t n_tiles_side,
                                    const uint32_t *__restrict__ flags, uint32_t *__restrict__ status)
{
    const uint32_t g = blockIdx.x;                            // (ta, tb), ta <= tb, row-major over the upper triangle
    uint32_t ta = 0, rem = g, row = n_tiles_side;
    while (rem >= row) { rem -= row; ++ta; --row; }
    const uint32_t tb = ta + rem;
    const uint32_t rank = g % world, local = g / world;
    const float *slab = gathered + (uint64_t)rank * slab_floats + (uint64_t)local * 2 * kSlotsPerTile;
    const int sa = threadIdx.x / kTile, sb = threadIdx.x % kTile;
    const uint32_t pa = ta * kTile + sa, pb = tb * kTile + sb;   // positions in the resident (length) order
    bool poisoned = false;
    if (pa < pb && pb < n_seq) {
        const uint32_t a = order[pa], b = order[pb];
        const float s1 = slab[sa * kTile + sb], s2 = slab[kSlotsPerTile + sa * kTile + sb];
        out[(uint64_t)a * n_seq + b] = s1;
        out[(uint64_t)b * n_seq + a] = s2;
        poisoned = (s1 != s1) | (s2 != s2);
    } else if (pa == pb && pb < n_seq) {
        const uint32_t a = order[pa];
        out[(uint64_t)a * n_seq + a] = 0.0f;                  // the diagonal is never aligned (alignments.rs:21-23, 51)
    }
    // every slab is filled with NaN before the alignment launches: with finite frames no score is NaN, so a NaN here is a
    // pair no kernel wrote (a launch cut short or skipped).  It stays NaN in the matrix and is reported, never a silent 0.
    if (__ballot(poisoned) != 0ull && (threadIdx.x & 63) == 0 && flags[0] == 0u) atomicOr(status, 1u);
}

__global__ void selftest_kernel(int *result)
{
    const int lane = threadIdx.x;
    const float v = (float)lane;
    bool ok = true;
    ok &= from_lower_lane(v, -1.0f) == (lane == 0 ? -1.0f : (float)(lane - 1));
    ok &= from_upper_lane(v, -2.0f) == (lane == 63 ? -2.0f : (float)(lane + 1));
    ok &= group_from_lower<16>(v, -3.0f, lane % 16) == (lane % 16 == 0 ? -3.0f : (float)(lane - 1));
    ok &= group_from_upper<16>(v, -4.0f, lane % 16) == (lane % 16 == 15 ? -4.0f : (float)(lane + 1));
    ok &= group_from_lower<32>(v, -5.0f, lane % 32) == (lane % 32 == 0 ? -5.0f : (float)(lane - 1));
    ok &= group_from_upper<32>(v, -7.0f, lane % 32) == (lane % 32 == 31 ? -7.0f : (float)(lane + 1));
    ok &= group_from_upper<32>(v, (float)(100 + lane), lane % 32) == (lane % 32 == 31 ? (float)(100 + lane) : (float)(lane + 1));   // per-lane keep value
    ok &= group_from_upper<8>(v, -6.0f, lane % 8) == (lane % 8 == 7 ? -6.0f : (float)(lane + 1));
    const unsigned long long all = __ballot(ok);
    if (lane == 0) *result = (all == ~0ull) ? 1 : 0;
}

// Exhaustive check of sqrt_rn_finite (dtw_common.h) against the compiler's correctly rounded sqrtf over the bit patterns
// [first, first + count): out[0] = patterns inside the domain (2^-96 <= x < +INF) where the two differ, out[1] = the first such
// pattern + 1, out[2..6] = how often the raw v_sqrt_f32 is off by -2 or less, -1, 0, +1, +2 or more ulps there.
__global__ __launch_bounds__(256) void sqrt_sweep_kernel(uint32_t first, uint64_t count, unsigned long long *out)
{
    unsigned long long bad = 0, hist[5] = {0, 0, 0, 0, 0};
    uint32_t first_bad = 0xFFFFFFFFu;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t bits = first + (uint32_t)e;
        const float x = __builtin_bit_cast(float, bits);
        if (!(x >= 0x1p-96f && x < APD_INF)) continue;
        const float want = __builtin_sqrtf(x), got = sqrt_rn_finite(x), raw = __builtin_amdgcn_sqrtf(x);
        if (__builtin_bit_cast(uint32_t, want) != __builtin_bit_cast(uint32_t, got)) { ++bad; first_bad = min(first_bad, bits); }
        const int off = __builtin_bit_cast(int, raw) - __builtin_bit_cast(int, want);
        ++hist[off <= -2 ? 0 : off >= 2 ? 4 : off + 2];
    }
    if (bad) { atomicAdd(&out[0], bad); atomicMin(&out[1], (unsigned long long)first_bad + 1ull); }
#pragma unroll
    for (int k = 0; k < 5; ++k) if (hist[k]) atomicAdd(&out[2 + k], hist[k]);
}

hipError_t launch_sqrt_sweep(uint32_t first, uint64_t count, unsigned long long *d_out, hipStream_t stream)
{
    hipLaunchKernelGGL(sqrt_sweep_kernel, dim3(4096), dim3(256), 0, stream, first, count, d_out);
    return hipGetLastError();
}

hipError_t launch_selftest(int *d_result, hipStream_t stream)
{
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, stream, d_result);
    return hipGetLastError();
}

hipError_t launch_pad(const float *d_src, float *d_dst, const uint32_t *d_seq_off, const uint32_t *d_src_off, uint32_t n_seq,
                      uint64_t n_frames_padded, uint32_t src_dim, uint32_t dim, uint32_t dpad, uint32_t *d_flags, float *d_seq_nmax,
                      hipStream_t stream)
{
    if (n_frames_padded == 0) return hipSuccess;
    const uint64_t total = n_frames_padded * (dpad / 4);
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((total + 255) / 256, 16384);
    hipLaunchKernelGGL(pad_frames_kernel, dim3(blocks), dim3(256), 0, stream, d_src, d_dst, d_seq_off, d_src_off, n_seq, n_frames_padded,
                       src_dim, dim, dpad, d_flags, d_seq_nmax);
    return hipGetLastError();
}

hipError_t launch_unpack(const float *d_gathered, float *d_out, const uint32_t *d_order, uint32_t n_seq, uint32_t world,
                         uint64_t slab_floats, const uint32_t *d_flags, uint32_t *d_status, hipStream_t stream)
{
    const uint32_t side = (n_seq + kTile - 1) / kTile;
    const uint64_t n_tiles = (uint64_t)side * (side + 1) / 2;
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_tiles_kernel, dim3((uint32_t)n_tiles), dim3(kSlotsPerTile), 0, stream, d_gathered, d_out,
                       d_order, n_seq, world, slab_floats, side, d_flags, d_status);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------------------
struct Geometry { int g, c; };

// Pick the (lanes per pair, offsets per lane) that wastes the fewest lanes for a band of `need` offsets; 0 = generic.
int pick_geometry_key(uint32_t need, uint32_t dim, int variant, bool uniform_pen, bool fast_shift)
{
    if (variant == 1) return 0;                                        // forced generic kernel
    if (variant >= 20000) variant = 0;                                 // a forced full-matrix geometry (pick_full_key) says nothing about band-form tiles
    if (!is_kernel_dim(dim)) return 0;                                 // instantiated frame dimensions
    if (variant >= 10000) {                                            // forced wide geometry (tuning)
        const int nw = (variant - 10000) / 100, c = variant % 100;
        return (uniform_pen && (uint32_t)(64 * nw * c) >= need) ? variant : 0;
    }
    if (variant >= 100) {                                              // forced geometry (tuning)
        const int g = variant / 100, c = variant % 100;
        return ((uint32_t)(g * c) >= need) ? variant : 0;
    }
    // score = lanes busy x how cheap a cell is at this C.  A macro-step costs about 116 SIMD cycles (window moves, exchanges,
    // threshold test) plus 76 per cell pair, so relative to C = 9 a cell costs 1.04 / 1.11 / 1.29 / 1.50
    // at C = 7 / 5 / 3 / 2.  G = 8 and G = 32 span half / two DPP rows: one-instruction moves only in the hybrid form with unit
    // penalties (`fast_shift`: masked column fetch), two instructions or a select per move otherwise.
    static const Geometry all[] = {{8, 5}, {8, 7}, {8, 9}, {16, 2}, {16, 3}, {16, 5}, {16, 7}, {16, 9}, {32, 5}, {32, 7}, {32, 9}, {64, 3}, {64, 5}, {64, 7}, {64, 9}};
    auto cell_eff = [](int c) { return c >= 9 ? 1.0 : c >= 7 ? 0.96 : c >= 5 ? 0.90 : c >= 3 ? 0.78 : 0.67; };
    Geometry best{0, 0};
    double best_util = 0.0;
    for (const Geometry &q : all) {
        if ((uint32_t)(q.g * q.c) < need || q.c > max_cells_per_lane(dim)) continue;
        const bool split_rows = q.g == 8 || q.g == 32;
        const double util = (double)need / (double)(q.g * q.c) * cell_eff(q.c) * (split_rows ? (fast_shift ? 0.99 : 0.90) : 1.0);
        if (util > best_util) { best_util = util; best = q; }
    }
    if (best.g != 0) return best.g * 100 + best.c;
    // beyond one wavefront: NW waves per pair (dtw_wide.h), uniform penalties only
    if (uniform_pen) {
        static const Geometry wide[] = {{2, 5}, {2, 7}, {2, 9}, {4, 5}, {4, 7}, {4, 9}, {8, 5}, {8, 7}, {8, 9}};   // (NW, C), ascending capacity
        for (const Geometry &q : wide)
            if ((uint32_t)(64 * q.g * q.c) >= need && q.c <= max_cells_per_lane(dim)) return 10000 + q.g * 100 + q.c;
    }
    return 0;
}

// Full-matrix kernel: 20000 + (pairs per wavefront) * 100 + CW.  With G = 64 / ppw lanes per pair, a pair of `rows` x `cols`
// takes ceil(cols / (G CW)) passes of rows + G macro-steps, each serving ppw pairs.  The cost of a macro-step was fitted on
// MI355X (tools/debug/strip_grid.sh: every forced geometry on bench.py's short9 / full6 / ship / full8): proportional to
// CW (+0.5 / +1.5 for the two-DPP moves and per-lane bookkeeping of 32- / 16-lane groups), 8 % more once CW frames of D + 1
// floats push the kernel to two waves per SIMD, and inversely to the wavefronts per CU that the LDS boundary columns
// (ppw * rows floats per wavefront) leave room for.  Returns +inf for a geometry that is not instantiated or does not fit.
double full_key_cost(uint32_t cols, uint32_t rows, uint32_t dim, int key)
{
    const bool banded = key >= 30000;
    const int ppw = (key % 10000) / 100, cw = key % 100;
    if (key < 20000 || key >= 40000) return INFINITY;
    if (!banded && (!(ppw == 1 || ppw == 2 || ppw == 4) || cw < 3 || cw > max_strip_columns(dim) || cw % 2 == 0)) return INFINITY;
    if (banded && (!(ppw == 1 || ppw == 4) || !(cw == 5 || cw == 9) || cw > max_cells_per_lane(dim))) return INFINITY;   // instantiated: dtw_full.h
    const uint32_t g = 64u / ppw;
    const double lds_bytes = 4.0 * ((g == 64 ? 128.0 : 64.0) * ((dim + 4) & ~3u) + (double)ppw * (banded ? 2 : 1) * (rows + 4) + 16);
    if (lds_bytes > 160.0 * 1024) return INFINITY;
    const double waves_per_cu = std::floor(160.0 * 1024 / lds_bytes);
    const double lds_factor = waves_per_cu >= 8.0 ? 1.0 : 8.0 / waves_per_cu;
    const double group_steps = ppw == 1 ? 0.0 : (ppw == 2 ? 0.5 : 1.5);
    const double occupancy_factor = (uint32_t)cw * (dim + 3) > 150 ? 1.08 : 1.0;
    const double passes = (double)((cols + g * cw - 1) / (g * cw));
    return passes * ((double)rows + g) * (cw + group_steps) * occupancy_factor * lds_factor / ppw;
}

int pick_full_key(uint32_t cols, uint32_t rows, uint32_t dim, int variant)
{
    if (variant != 0 && (variant < 20000 || variant >= 30000)) return 0;   // another kernel was requested
    if (!is_kernel_dim(dim)) return 0;
    if (variant >= 20000) return std::isfinite(full_key_cost(cols, rows, dim, variant)) ? variant : 0;
    int best = 0;
    double best_cost = INFINITY;
    for (int ppw = 1; ppw <= 4; ppw *= 2)
        for (int cw = 5; cw <= max_strip_columns(dim); cw += 2) {          // 3-column strips only on request: measured slower than the model says
            const double cost = full_key_cost(cols, rows, dim, 20000 + ppw * 100 + cw);
            if (cost < best_cost) { best = 20000 + ppw * 100 + cw; best_cost = cost; }
        }
    return best;
}

int pick_banded_strip_key(uint32_t cols, uint32_t rows, uint32_t dim, int variant)
{
    if (variant != 0 && (variant < 30000 || variant >= 40000)) return 0;
    if (!is_kernel_dim(dim)) return 0;
    if (variant >= 30000) return std::isfinite(full_key_cost(cols, rows, dim, variant)) ? variant : 0;
    int best = 0;
    double best_cost = INFINITY;
    for (int ppw = 1; ppw <= 4; ppw *= 4)
        for (int cw = 5; cw <= 9; cw += 4) {
            const double cost = full_key_cost(cols, rows, dim, 30000 + ppw * 100 + cw);
            if (cost < best_cost) { best = 30000 + ppw * 100 + cw; best_cost = cost; }
        }
    return best;
}

static hipError_t launch_align_chunk(const AlignLaunch &L, int geom_key, hipStream_t stream, std::string &err, int *status);

// A launch may not exceed 2^32 work-items (beyond, the grid is silently cut short on this runtime): launches are cut
// into runs of tiles that stay below 2^31.  Work-items per tile: 256 pairs x lanes per pair.
hipError_t launch_align(const AlignLaunch &L, int geom_key, hipStream_t stream, std::string &err, int *status)
{
    *status = APD_OK;
    uint32_t lanes_per_pair = 64;                                         // generic kernel: one wavefront per pair
    if (geom_key >= 20000) lanes_per_pair = 64;                                           // full-matrix: at most one wavefront per pair
    else if (geom_key >= 10000) lanes_per_pair = 64u * (uint32_t)((geom_key % 10000) / 100);   // wide: NW waves per pair
    else if (geom_key != 0) lanes_per_pair = (uint32_t)(geom_key / 100);                   // systolic: G lanes per pair
    const uint32_t kTilesPerLaunch = (1u << 31) / (kSlotsPerTile * std::max(lanes_per_pair, 64u));
    for (uint32_t first = 0; first < L.n_tiles; first += kTilesPerLaunch) {
        AlignLaunch part = L;
        part.d_tiles = L.d_tiles + first;
        part.n_tiles = std::min(kTilesPerLaunch, L.n_tiles - first);
        const hipError_t e = launch_align_chunk(part, geom_key, stream, err, status);
        if (e != hipSuccess || *status != APD_OK) return e;
    }
    return hipSuccess;
}

static hipError_t launch_align_chunk(const AlignLaunch &L, int geom_key, hipStream_t stream, std::string &err, int *status)
{
    *status = APD_OK;
    if (L.n_tiles == 0) return hipSuccess;
    const BandSpec &b = L.band;
    const bool unit_pens = (b.ins == 1.0f) && (b.del == 1.0f) && (b.mat == 1.0f);
    // the systolic kernel's UNIFORM_PEN path (no weighting at all).  Strict mode keeps it: with unit penalties the fast select on
    // the difference form -- which in the band kernels is the reference's arithmetic operation for operation -- gives the CPU
    // code's bits (see dtw_systolic.h); only the hybrid distance form is switched off.
    const bool unit = unit_pens;
    bool done = false;
    AlignLaunch LL = L;
    // The norm expansion trades D - 1 vector ops per cell (systolic kernel, pre-scaled rows; D - 4 in the strip kernels) for a
    // branch per macro-step: measured worth it from D = 8 in the systolic kernel (cfg 4: -2.6 %), from D = 10 elsewhere.
    static const int hybrid_min_env = std::getenv("APD_HYBRID_MIN_DIM") ? std::atoi(std::getenv("APD_HYBRID_MIN_DIM")) : 0;   // tuning aid
    const int hybrid_min_dim = hybrid_min_env ? hybrid_min_env : (geom_key >= 100 && geom_key < 10000 ? 8 : 10);
    if ((int)LL.dim < hybrid_min_dim) LL.hybrid = 0;
    if (L.strict && geom_key >= 100 && geom_key < 10000) LL.hybrid = 0;
    if (geom_key >= 20000) {
        const bool banded = geom_key >= 30000;
        const int nw = (geom_key % 10000) / 100, cw = geom_key % 100;
        hipError_t fe = hipSuccess;
        switch (L.dim) {
            case 8: done = launch_full<8>(LL, banded, nw, cw, stream, &fe); break;
            case 10: done = launch_full<10>(LL, banded, nw, cw, stream, &fe); break;
            case 13: done = launch_full<13>(LL, banded, nw, cw, stream, &fe); break;
            case 16: done = launch_full<16>(LL, banded, nw, cw, stream, &fe); break;
            case 20: done = launch_full<20>(LL, banded, nw, cw, stream, &fe); break;
            case 26: done = launch_full<26>(LL, banded, nw, cw, stream, &fe); break;
            default: break;
        }
        if (done && fe != hipSuccess) return fe;
    } else if (geom_key >= 10000) {
        const int nw = (geom_key - 10000) / 100, c = geom_key % 100;
        hipError_t we = hipSuccess;
        switch (L.dim) {
            case 8: done = launch_wide<8>(LL, nw, c, stream, &we); break;
            case 10: done = launch_wide<10>(LL, nw, c, stream, &we); break;
            case 13: done = launch_wide<13>(LL, nw, c, stream, &we); break;
            case 16: done = launch_wide<16>(LL, nw, c, stream, &we); break;
            case 20: done = launch_wide<20>(LL, nw, c, stream, &we); break;
            case 26: done = launch_wide<26>(LL, nw, c, stream, &we); break;
            default: break;
        }
        if (done && we != hipSuccess) return we;
    } else if (geom_key != 0) {
        const int g = geom_key / 100, c = geom_key % 100;
        switch (L.dim) {
            case 8: done = launch_systolic<8>(LL, g, c, unit, stream); break;
            case 10: done = launch_systolic<10>(LL, g, c, unit, stream); break;
            case 13: done = launch_systolic<13>(LL, g, c, unit, stream); break;
            case 16: done = launch_systolic<16>(LL, g, c, unit, stream); break;
            case 20: done = launch_systolic<20>(LL, g, c, unit, stream); break;
            case 26: done = launch_systolic<26>(LL, g, c, unit, stream); break;
            default: break;
        }
    }
    if (!done) {
        int c_max = (int)((2 * (uint64_t)L.w_max + 1 + 63) / 64);
        if (c_max < 2) c_max = 2;
        const size_t lds_bytes = (size_t)c_max * 64 * 2 * sizeof(float);
        if (lds_bytes > 160 * 1024) { *status = APD_ERR_BAND_TOO_WIDE; err = "band too wide for the generic kernel"; return hipSuccess; }
        if (lds_bytes > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(dtw_fused_generic),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) return e;
        }
        const uint64_t waves = (uint64_t)L.n_tiles * kSlotsPerTile;
        hipLaunchKernelGGL(dtw_fused_generic, dim3((uint32_t)waves), dim3(64), lds_bytes, stream, L, c_max);
    }
    return hipGetLastError();
}

static size_t generic_lds_bytes(uint32_t w_max, int *c_max_out)
{
    int c_max = (int)((2 * (uint64_t)w_max + 1 + 63) / 64);
    if (c_max < 2) c_max = 2;
    if (c_max_out) *c_max_out = c_max;
    return (size_t)c_max * 64 * 2 * sizeof(float);
}

bool generic_fallback_fits(uint32_t w_max)
{
    const size_t lds_bytes = generic_lds_bytes(w_max, nullptr);
    if (lds_bytes > 160 * 1024) return false;                             // gfx950: 160 KB of LDS per workgroup
    if (lds_bytes > 64 * 1024 &&                                          // beyond the default limit: ask now, not after the fast kernels are enqueued
        hipFuncSetAttribute(reinterpret_cast<const void *>(dtw_fused_generic_fallback), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds_bytes) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return true;
}

hipError_t launch_generic_fallback(const AlignLaunch &L, hipStream_t stream, bool *fits)
{
    *fits = true;
    if (L.n_tiles == 0 || L.d_nonfinite == nullptr) return hipSuccess;
    int c_max = 2;
    const size_t lds_bytes = generic_lds_bytes(L.w_max, &c_max);
    if (lds_bytes > 160 * 1024) { *fits = false; return hipSuccess; }
    if (lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(dtw_fused_generic_fallback),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    const uint64_t waves = (uint64_t)L.n_tiles * kSlotsPerTile;
    if (waves > 0xFFFFFFFFull) { *fits = false; return hipSuccess; }
    const uint32_t grid = (uint32_t)std::min<uint64_t>(waves, 256u * 32u);   // 32 single-wave workgroups per CU when it has to work
    hipLaunchKernelGGL(dtw_fused_generic_fallback, dim3(grid), dim3(64), lds_bytes, stream, L, c_max, (uint32_t)waves);
    return hipGetLastError();
}

}  // namespace apd

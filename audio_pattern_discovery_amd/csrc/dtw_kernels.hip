// Fused-pair banded DTW for gfx950 (MI355X).
//
// Replaces the per-pair body of AlignmentWorkers::align_all (reference src/alignments.rs:50-58):
// Alignment::new + construct_alignment (:165-180) + alignment_score (:129-160) + score (:116-125).
//
// Work unit: one UNORDERED pair (a, b), a < b.  One wavefront computes BOTH ordered scores
//   DP1 = score(x = a, y = b)  and  DP2 = score(x = b, y = a)
// in one sweep, because the two recurrences visit (almost) the same cells and share every local
// distance euclidean(A[i], B[j]) (numerics.rs:114-120).  In A-row / B-column coordinates
// (i over A, j over B, band offset o = j - i, u = o + w):
//   DP1: u in [0, 2w-1], left neighbour (i, j-1) is the DELETE branch, up (i-1, j) the INSERT branch;
//   DP2: u in [1, 2w],   up   neighbour is the DELETE branch, left the INSERT branch
// (DP2's cell (j, i) of the swapped problem is our cell (i, j); the reference band j'-i' in [-w, w-1]
//  of alignments.rs:175 becomes o in [-w+1, w]).
//
// Lane mapping: lane l owns C consecutive offsets u = C*l + c.  Macro-step tau: every lane processes
// row i = tau - l, its C cells left to right.  Dependences across lanes:
//   (i, u-1) of the first cell  = last cell of lane l-1 from the PREVIOUS macro-step  -> DPP wave_shr:1
//   (i-1, u+1) of the last cell = first cell of lane l+1 from THIS macro-step          -> DPP wave_shl:1
// so C >= 2 and no LDS round trip is needed for the min(INS, DEL, MATCH) dependency.
// Only rows 1..n-1 and columns 1..m-1 are swept: score() reads cell (n-1, m-1) (alignments.rs:120),
// row n and column m never influence it.
#include "apd_internal.h"

namespace apd {

#define APD_INF __builtin_inff()

// lane l receives the value of lane l-1; lane 0 receives `fill`
__device__ __forceinline__ float from_lower_lane(float v, float fill)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill),
                                                                 __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
// lane l receives the value of lane l+1; lane 63 receives `fill`
__device__ __forceinline__ float from_upper_lane(float v, float fill)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill),
                                                                 __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

__device__ __forceinline__ uint32_t band_from_pct(float pct, uint32_t len)
{
    float p = pct * (float)len;                 // discovery.rs:40, one f32 rounding
    if (!(p > 0.0f)) return 0u;
    if (p >= 4294967040.0f) return 0xFFFFFFFFu;
    return (uint32_t)p;
}

__device__ __forceinline__ int pair_w(const BandSpec &b, int n, int m)
{
    uint32_t mx = (uint32_t)max(n, m), gap = (uint32_t)abs(n - m);
    uint32_t band = b.use_explicit ? b.explicit_band : band_from_pct(b.pct, mx);
    band = min(band, mx);
    return (int)(max(band, gap) + 2u);          // alignments.rs:173
}

// alignments.rs:153-159.  `del_v`/`ins_v` are the DELETE / INSERT predecessors, m_v the MATCH one.
template <bool UNIFORM_PEN>
__device__ __forceinline__ float select_node(float del_v, float ins_v, float m_v, float d, float del_pen,
                                             float ins_pen, float mat_pen)
{
    const bool pick_d = (del_v < m_v) & (del_v < ins_v);
    const bool pick_i = (ins_v < m_v) & (ins_v < del_v);
    float base = pick_i ? ins_v : m_v;
    base = pick_d ? del_v : base;
    if (UNIFORM_PEN) return __builtin_fmaf(mat_pen, d, base);
    float pen = pick_i ? ins_pen : mat_pen;
    pen = pick_d ? del_pen : pen;
    return __builtin_fmaf(pen, d, base);
}

struct PairInfo {
    const float *A, *B;   // padded frames of a (rows) and b (columns)
    int n, m, w;
    int slot_a, slot_b;   // position inside the tile
    bool valid;
};

__device__ __forceinline__ PairInfo decode_pair(const AlignLaunch &L, uint32_t tile, uint32_t slot)
{
    PairInfo p;
    const uint2 t = L.d_tiles[tile];
    p.slot_a = slot / kTile;
    p.slot_b = slot % kTile;
    const uint32_t a = t.x * kTile + p.slot_a, b = t.y * kTile + p.slot_b;
    p.valid = (a < b) && (b < L.n_seq);
    if (!p.valid) { p.A = p.B = nullptr; p.n = p.m = p.w = 0; return p; }
    const uint32_t oa = L.d_seq_off[a], ob = L.d_seq_off[b];
    p.n = (int)(L.d_seq_off[a + 1] - oa);
    p.m = (int)(L.d_seq_off[b + 1] - ob);
    p.A = L.d_frames + (uint64_t)oa * L.dpad;
    p.B = L.d_frames + (uint64_t)ob * L.dpad;
    p.w = pair_w(L.band, p.n, p.m);
    return p;
}

__device__ __forceinline__ void store_pair(const AlignLaunch &L, uint32_t tile, const PairInfo &p, float s1, float s2)
{
    float *slab = L.d_slab + (uint64_t)tile * 2 * kSlotsPerTile;
    slab[p.slot_a * kTile + p.slot_b] = s1;                       // score(x=a, y=b)
    slab[kSlotsPerTile + p.slot_a * kTile + p.slot_b] = s2;       // score(x=b, y=a)
}

template <int D>
__device__ __forceinline__ void load_frame(float (&dst)[D], const float *__restrict__ p)
{
    constexpr int DP4 = (D + 3) / 4;
#pragma unroll
    for (int q = 0; q < DP4; ++q) {
        const float4 t = reinterpret_cast<const float4 *>(p)[q];
        if (4 * q + 0 < D) dst[4 * q + 0] = t.x;
        if (4 * q + 1 < D) dst[4 * q + 1] = t.y;
        if (4 * q + 2 < D) dst[4 * q + 2] = t.z;
        if (4 * q + 3 < D) dst[4 * q + 3] = t.w;
    }
}

// numerics.rs:114-120 with an fma chain (k ascending); v_sqrt_f32 is within 1 ulp.
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

// ------------------------------------------------------------------------------------------------
// v0 "direct" kernel: one wave per unordered pair, lanes own C offsets, frames read straight from
// global memory (L1/L2).  Correctness-first; every mask evaluated per cell.
// ------------------------------------------------------------------------------------------------
template <int D, int C, bool UNIFORM_PEN>
__global__ __launch_bounds__(256) void dtw_fused_direct(const AlignLaunch L)
{
    static_assert(C >= 2, "a lane must own at least two band offsets");
    constexpr int DP = (D + 3) & ~3;
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * (blockDim.x >> 6)) + (threadIdx.x >> 6);
    const uint32_t tile = wave / kSlotsPerTile, slot = wave % kSlotsPerTile;
    if (tile >= L.n_tiles) return;
    const PairInfo P = decode_pair(L, tile, slot);
    if (!P.valid) return;
    const int n = P.n, m = P.m, w = P.w;
    if (n == 1 || m == 1) {                                   // alignments.rs:116-125 with an absent cell
        if (lane == 0) { const float s = (n == 1 && m == 1) ? 0.0f : APD_INF; store_pair(L, tile, P, s, s); }
        return;
    }
    const float ins = L.band.ins, del = L.band.del, mat = L.band.mat;
    const int u0 = C * lane;
    const int two_w = 2 * w;
    const int g_act = (two_w + 1 + C - 1) / C;
    int total = (n - 1) + g_act;
    total = ((total + C - 1) / C) * C;

    float prev1[C], prev2[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { prev1[c] = APD_INF; prev2[c] = APD_INF; }
    float res1 = 0.0f, res2 = 0.0f;

    float yf[C][D];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int j = 0 - lane + u0 + c - w;
        load_frame<D>(yf[c], P.B + (uint64_t)(min(max(j, 1), m) - 1) * DP);
    }

    for (int tau0 = 0; tau0 < total; tau0 += C) {
#pragma unroll
        for (int r = 0; r < C; ++r) {
            const int tau = tau0 + r;
            const int i = tau - lane;
            const int jb = i + u0 - w;
            float xr[D];
            load_frame<D>(xr, P.A + (uint64_t)(min(max(i, 1), n) - 1) * DP);
            float left1 = from_lower_lane(prev1[C - 1], APD_INF);
            float left2 = from_lower_lane(prev2[C - 1], APD_INF);
            float upr1 = APD_INF, upr2 = APD_INF;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const int j = jb + c, u = u0 + c;
                const float d = frame_dist<D>(xr, yf[(r + c) % C]);
                const float up1 = (c < C - 1) ? prev1[(c + 1) % C] : upr1;
                const float up2 = (c < C - 1) ? prev2[(c + 1) % C] : upr2;
                float r1 = select_node<UNIFORM_PEN>(left1, up1, prev1[c], d, del, ins, mat);
                float r2 = select_node<UNIFORM_PEN>(up2, left2, prev2[c], d, del, ins, mat);
                const bool inside = (i >= 1) & (j >= 1);
                const float inv = ((i == 0) & (j == 0)) ? 0.0f : APD_INF;   // D[0][0] = 0 (alignments.rs:109)
                r1 = (inside & (u <= two_w - 1)) ? r1 : inv;
                r2 = (inside & (u >= 1) & (u <= two_w)) ? r2 : inv;
                if ((i == n - 1) & (j == m - 1)) { res1 = r1; res2 = r2; }
                prev1[c] = r1; prev2[c] = r2;
                left1 = r1; left2 = r2;
                if (c == 0) {
                    upr1 = from_upper_lane(r1, APD_INF);
                    upr2 = from_upper_lane(r2, APD_INF);
                    // column entering the lane's window at the next macro-step replaces the one just used
                    const int jn = jb + C;
                    load_frame<D>(yf[r % C], P.B + (uint64_t)(min(max(jn, 1), m) - 1) * DP);
                }
            }
        }
    }
    const int ustar = (m - 1) - (n - 1) + w;
    if (lane == ustar / C) {
        const float denom = (float)(n + m);                  // alignments.rs:121
        store_pair(L, tile, P, res1 / denom, res2 / denom);
    }
}

// ------------------------------------------------------------------------------------------------
// Generic kernel: any dim, any band that fits LDS (C chosen at run time, per-lane DP state in LDS).
// Slow path for unusual dims and very wide bands; also an independent check of the templated path.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void dtw_fused_generic(const AlignLaunch L, int c_max)
{
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    const uint32_t wave = blockIdx.x;
    const uint32_t tile = wave / kSlotsPerTile, slot = wave % kSlotsPerTile;
    if (tile >= L.n_tiles) return;
    const PairInfo P = decode_pair(L, tile, slot);
    if (!P.valid) return;
    const int n = P.n, m = P.m, w = P.w;
    if (n == 1 || m == 1) {
        if (lane == 0) { const float s = (n == 1 && m == 1) ? 0.0f : APD_INF; store_pair(L, tile, P, s, s); }
        return;
    }
    const float ins = L.band.ins, del = L.band.del, mat = L.band.mat;
    const int two_w = 2 * w;
    int C = (two_w + 1 + 63) / 64;
    C = max(C, 2);
    float *p1 = lds, *p2 = lds + c_max * 64;               // [c][lane]
    for (int c = 0; c < C; ++c) { p1[c * 64 + lane] = APD_INF; p2[c * 64 + lane] = APD_INF; }
    const int dp4 = (int)L.dpad / 4;
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
            float acc = 0.0f;
            for (int q = 0; q < dp4; ++q) {
                const float4 xv = xa[q], yv = yb[q];
                float t = xv.x - yv.x;
                acc = (q == 0) ? t * t : __builtin_fmaf(t, t, acc);
                t = xv.y - yv.y; acc = __builtin_fmaf(t, t, acc);
                t = xv.z - yv.z; acc = __builtin_fmaf(t, t, acc);
                t = xv.w - yv.w; acc = __builtin_fmaf(t, t, acc);
            }
            const float d = __builtin_amdgcn_sqrtf(acc);
            const float m1 = nxt1, m2 = nxt2;
            float up1, up2;
            if (c < C - 1) { up1 = p1[(c + 1) * 64 + lane]; up2 = p2[(c + 1) * 64 + lane]; }
            else { up1 = upr1; up2 = upr2; }
            nxt1 = up1; nxt2 = up2;                          // prev[c+1] is the next cell's MATCH predecessor
            float r1 = select_node<false>(left1, up1, m1, d, del, ins, mat);
            float r2 = select_node<false>(up2, left2, m2, d, del, ins, mat);
            const bool inside = (i >= 1) & (j >= 1);
            const float inv = ((i == 0) & (j == 0)) ? 0.0f : APD_INF;
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
        const float denom = (float)(n + m);
        store_pair(L, tile, P, res1 / denom, res2 / denom);
    }
}

// ------------------------------------------------------------------------------------------------
// helpers: padding, unpack, self-test
// ------------------------------------------------------------------------------------------------
__global__ void pad_frames_kernel(const float *__restrict__ src, float *__restrict__ dst, uint64_t n_frames,
                                  uint32_t dim, uint32_t dpad)
{
    const uint64_t total = n_frames * dpad;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t f = e / dpad;
        const uint32_t k = (uint32_t)(e - f * dpad);
        dst[e] = (k < dim) ? src[f * dim + k] : 0.0f;
    }
}

// gathered: `world` slabs of slab_floats each; slab r holds tiles r, r+world, ... in order.
__global__ void unpack_tiles_kernel(const float *__restrict__ gathered, float *__restrict__ out, uint32_t n_seq,
                                    uint32_t world, uint64_t slab_floats, uint32_t n_tiles_side)
{
    // global tile index g enumerates (ta, tb), ta <= tb, row-major over the upper triangle
    const uint32_t g = blockIdx.x;
    // invert g -> (ta, tb)
    uint32_t ta = 0, rem = g, row = n_tiles_side;
    while (rem >= row) { rem -= row; ++ta; --row; }
    const uint32_t tb = ta + rem;
    const uint32_t rank = g % world, local = g / world;
    const float *slab = gathered + (uint64_t)rank * slab_floats + (uint64_t)local * 2 * kSlotsPerTile;
    const int sa = threadIdx.x / kTile, sb = threadIdx.x % kTile;
    const uint32_t a = ta * kTile + sa, b = tb * kTile + sb;
    if (a < b && b < n_seq) {
        out[(uint64_t)a * n_seq + b] = slab[sa * kTile + sb];
        out[(uint64_t)b * n_seq + a] = slab[kSlotsPerTile + sa * kTile + sb];
    }
}

__global__ void selftest_kernel(int *result)
{
    const int lane = threadIdx.x;
    const float v = (float)lane;
    const float lo = from_lower_lane(v, -1.0f), up = from_upper_lane(v, -2.0f);
    const float want_lo = lane == 0 ? -1.0f : (float)(lane - 1);
    const float want_up = lane == 63 ? -2.0f : (float)(lane + 1);
    const bool ok = (lo == want_lo) && (up == want_up);
    const unsigned long long all = __ballot(ok);
    if (lane == 0) *result = (all == ~0ull) ? 1 : 0;
}

hipError_t launch_selftest(int *d_result, hipStream_t stream)
{
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, stream, d_result);
    return hipGetLastError();
}

hipError_t launch_pad(const float *d_src, float *d_dst, uint64_t n_frames, uint32_t dim, uint32_t dpad, hipStream_t stream)
{
    if (n_frames == 0) return hipSuccess;
    const uint64_t total = n_frames * dpad;
    const uint32_t blocks = (uint32_t)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(pad_frames_kernel, dim3(blocks), dim3(256), 0, stream, d_src, d_dst, n_frames, dim, dpad);
    return hipGetLastError();
}

hipError_t launch_unpack(const float *d_gathered, float *d_out, uint32_t n_seq, uint32_t world, uint64_t slab_floats,
                         hipStream_t stream)
{
    const uint32_t side = (n_seq + kTile - 1) / kTile;
    const uint64_t n_tiles = (uint64_t)side * (side + 1) / 2;
    if (n_tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(unpack_tiles_kernel, dim3((uint32_t)n_tiles), dim3(kSlotsPerTile), 0, stream, d_gathered, d_out,
                       n_seq, world, slab_floats, side);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// dispatch
// ------------------------------------------------------------------------------------------------
template <int D, int C>
static void launch_direct(const AlignLaunch &L, bool uniform, hipStream_t stream)
{
    const uint64_t waves = (uint64_t)L.n_tiles * kSlotsPerTile;
    const dim3 grid((uint32_t)((waves + 3) / 4)), block(256);
    if (uniform) hipLaunchKernelGGL((dtw_fused_direct<D, C, true>), grid, block, 0, stream, L);
    else hipLaunchKernelGGL((dtw_fused_direct<D, C, false>), grid, block, 0, stream, L);
}

template <int D>
static bool launch_direct_c(const AlignLaunch &L, int c, bool uniform, hipStream_t stream)
{
    switch (c) {
        case 2: launch_direct<D, 2>(L, uniform, stream); return true;
        case 3: launch_direct<D, 3>(L, uniform, stream); return true;
        case 5: launch_direct<D, 5>(L, uniform, stream); return true;
        case 9: launch_direct<D, 9>(L, uniform, stream); return true;
        default: return false;
    }
}

static int pick_c(uint32_t w_max)
{
    const uint32_t need = 2 * w_max + 1;
    const int cs[] = {2, 3, 5, 9};
    for (int c : cs) if ((uint32_t)c * 64u >= need) return c;
    return 0;
}

hipError_t launch_align(const AlignLaunch &L, hipStream_t stream, std::string &err, int *status)
{
    *status = APD_OK;
    if (L.n_tiles == 0) return hipSuccess;
    const bool uniform = (L.band.ins == L.band.del) && (L.band.del == L.band.mat);
    const int c = pick_c(L.w_max);
    bool done = false;
    if (L.variant != 1 && c != 0) {            // variant 1 forces the generic kernel
        switch (L.dim) {
            case 8: done = launch_direct_c<8>(L, c, uniform, stream); break;
            case 10: done = launch_direct_c<10>(L, c, uniform, stream); break;
            case 13: done = launch_direct_c<13>(L, c, uniform, stream); break;
            case 26: done = launch_direct_c<26>(L, c, uniform, stream); break;
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

}  // namespace apd

// Device helpers shared by the alignment kernels (gfx950).
#pragma once
#include "apd_internal.h"

// No implicit fusing of a*b+c anywhere in the alignment kernels (the Makefile passes -ffp-contract=off as well): the fma chains
// are written out with __builtin_fmaf, everything else rounds operation by operation as the reference does.
#pragma clang fp contract(off)

namespace apd {

#define APD_INF __builtin_inff()

// DPP controls (LLVM AMDGPU): row_shl:1 0x101, row_shr:1 0x111, wave_shl:1 0x130, wave_shr:1 0x138.
// "shr" moves data towards higher lanes: lane l reads lane l-1.  A lane without a source keeps `fill`.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v, float fill)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v),
                                                                 CTRL, 0xf, 0xf, false));
}
// whole-wave shifts (G = 64)
__device__ __forceinline__ float from_lower_lane(float v, float fill) { return dpp_move<0x138>(v, fill); }
__device__ __forceinline__ float from_upper_lane(float v, float fill) { return dpp_move<0x130>(v, fill); }

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move_rows(float v, float keep)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, keep), __builtin_bit_cast(int, v),
                                                                 CTRL, ROW_MASK, 0xf, false));
}

// Shifts inside groups of G lanes: lane 0 of a group (no lower neighbour) / lane G-1 (no upper) gets `fill`.
// G = 16 maps onto DPP rows and G = 64 onto the wave shifts (one instruction).  G = 32 takes two, each writing two of
// the four 16-lane rows: the rows whose source crosses a row boundary inside a group use the wave shift, the rows
// that end (begin) a group use the row shift, whose edge lane has no source and keeps `fill`.  Other sizes fix the
// group edges with a select.
template <int G>
__device__ __forceinline__ float group_from_lower(float v, float fill, int gl)
{
    if (G == 64) return dpp_move<0x138>(v, fill);
    if (G == 16) return dpp_move<0x111>(v, fill);
    if (G == 32) return dpp_move_rows<0x111, 0x5>(v, dpp_move_rows<0x138, 0xA>(v, fill));   // rows 1,3 read across; rows 0,2 start a group
    const float t = dpp_move<0x138>(v, fill);
    return gl == 0 ? fill : t;
}
template <int G>
__device__ __forceinline__ float group_from_upper(float v, float fill, int gl)
{
    if (G == 64) return dpp_move<0x130>(v, fill);
    if (G == 16) return dpp_move<0x101>(v, fill);
    if (G == 32) return dpp_move_rows<0x101, 0xA>(v, dpp_move_rows<0x130, 0x5>(v, fill));   // rows 0,2 read across; rows 1,3 end a group
    const float t = dpp_move<0x130>(v, fill);
    return gl == G - 1 ? fill : t;
}

// The same shifts for values whose group-edge lane does not care what it receives (or is known to receive +INF from the
// other end of the group): rotations, row_ror:1 0x121 / row_ror:15 0x12F, wave_ror:1 0x13C / wave_rol:1 0x134.  Every lane
// has a source, so no `fill` register has to be materialised before the move (one v_mov less per exchange).
template <int CTRL>
__device__ __forceinline__ float dpp_rotate(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
template <int G>
__device__ __forceinline__ float group_from_lower_wrap(float v, float fill, int gl)
{
    if (G == 16 || G == 8) return dpp_rotate<0x121>(v);          // G = 8, 32: the other group's edge node is as good as one's own
    if (G == 64 || G == 32) return dpp_rotate<0x13C>(v);
    return group_from_lower<G>(v, fill, gl);
}
template <int G>
__device__ __forceinline__ float group_from_upper_wrap(float v, float fill, int gl)
{
    if (G == 16 || G == 8) return dpp_rotate<0x12F>(v);
    if (G == 64 || G == 32) return dpp_rotate<0x134>(v);
    return group_from_upper<G>(v, fill, gl);
}

// Correctly rounded f32 square root (the bits of Rust's f32::sqrt, numerics.rs:119) for 2^-96 <= x < +INF, in 7 vector ops:
// s = v_sqrt_f32(x) is within one ulp; with s- / s+ its neighbours, the fused residuals x - s s- and x - s s+ are computed
// without rounding error that could change their sign (LLVM's own lowering rests on the same fact), and
//   x - s s- <= 0  <=>  sqrt(x) < the midpoint of (s-, s)  =>  s-          x - s s+ > 0  <=>  sqrt(x) > the midpoint of (s, s+)  =>  s+
// (sqrt(x) is never a midpoint).  The two selects are integer clamps of the residuals' bit patterns -- [r > 0] = med3(bits(r), 0,
// 1) for any non-NaN r -- summed onto bits(s-): s- + 0, + 1 (= s) or + 2 (= s+).  Outside the domain (zero, subnormal-range
// residuals below 2^-96, +INF, NaN) the result is meaningless: callers branch to __builtin_sqrtf there.  Exhaustively checked
// against the compiler's correctly rounded sqrtf on MI355X for every f32 in the domain (apd_selftest_sqrt, tests/test_gpu_sqrt.py).
__device__ __forceinline__ float sqrt_rn_finite(float x)
{
    const float s = __builtin_amdgcn_sqrtf(x);
    const int sb = __builtin_bit_cast(int, s);
    const float s_dn = __builtin_bit_cast(float, sb - 1), s_up = __builtin_bit_cast(float, sb + 1);
    const int rp = __builtin_bit_cast(int, __builtin_fmaf(-s_dn, s, x));
    const int rs = __builtin_bit_cast(int, __builtin_fmaf(-s_up, s, x));
    int up, keep;                                                 // written as max(min(r, 1), 0) the compiler canonicalises the clamps
    asm("v_med3_i32 %0, %1, 0, 1" : "=v"(up) : "v"(rs));          // back into compare + select pairs (twice the issue cost)
    asm("v_med3_i32 %0, %1, 0, 1" : "=v"(keep) : "v"(rp));
    return __builtin_bit_cast(float, (sb - 1) + up + keep);
}

__device__ __forceinline__ uint32_t band_from_pct(float pct, uint32_t len)
{
    float p = pct * (float)len;                 // discovery.rs:40, one f32 rounding
    if (!(p > 0.0f)) return 0u;                 // Rust `as usize` saturates: NaN and negatives -> 0
    if (p >= 4294967040.0f) return 0xFFFFFFFFu;
    return (uint32_t)p;
}

__device__ __forceinline__ int pair_w(const BandSpec &b, int n, int m)
{
    uint32_t mx = (uint32_t)max(n, m), gap = (uint32_t)abs(n - m);
    uint32_t band = b.use_explicit ? b.explicit_band : band_from_pct(b.pct, mx);
    band = min(band, mx);                       // any band >= max(n,m) already spans the whole matrix
    return (int)(max(band, gap) + 2u);          // alignments.rs:173
}

// alignments.rs:153-159.  del_v / ins_v are the DELETE / INSERT predecessors, m_v the MATCH one.
// `force_match` makes the node take the MATCH branch regardless: the systolic kernel sets it on the cells just
// outside a DP's band, whose MATCH predecessor (the same band offset one row up) is +INF by induction, so the cell
// evaluates to +INF at the price of one scalar mask OR instead of a vector select.
// The node is predecessor + (penalty * d) with the product rounded on its own, as the reference computes it: an fma
// here would move last bits and with them the exact ties the select rule is sensitive to (a DELETE/INSERT tie takes
// MATCH even when MATCH is larger).  UNIFORM_PEN: `d` is the already weighted distance (all penalties 1.0 in the
// systolic kernel, where 1.0 * d == d; weight_distances elsewhere).
template <bool UNIFORM_PEN>
__device__ __forceinline__ float select_node(float del_v, float ins_v, float m_v, float d, float del_pen, float ins_pen,
                                             float mat_pen, bool force_match = false)
{
    if (UNIFORM_PEN) {
        // With one penalty only the predecessor VALUE matters.  del_v < ins_v: DELETE iff del_v < m_v, i.e. the value
        // min(del_v, m_v); ins_v < del_v: likewise min(ins_v, m_v); del_v == ins_v: neither strict test can hold ->
        // m_v.  So base = (del_v == ins_v) ? m_v : min3(del_v, ins_v, m_v): 3 VALU ops, no scalar mask arithmetic.
        // (Identical to the branch chain for non-NaN inputs; NaN features are outside the supported domain.)
        const float lo = __builtin_fminf(__builtin_fminf(del_v, ins_v), m_v);
#if defined(APD_ABLATE) && (APD_ABLATE & 64)
        return lo + d;                                            // timing only: no tie rule, no band guards (results wrong)
#endif
        return (((del_v == ins_v) | force_match) ? m_v : lo) + d;
    }
    const bool pick_d = (del_v < m_v) & (del_v < ins_v) & !force_match;
    const bool pick_i = (ins_v < m_v) & (ins_v < del_v) & !force_match;
    float base = pick_i ? ins_v : m_v;
    base = pick_d ? del_v : base;
    float pen = pick_i ? ins_pen : mat_pen;
    pen = pick_d ? del_pen : pen;
    const float weighted = pen * d;                               // rounded on its own (alignments.rs:154-158)
    return base + weighted;
}

// Uniform penalties (wide / full-matrix kernels): the distances of a row are weighted once, ahead of the DP rows.
template <int C>
__device__ __forceinline__ void weight_distances(float (&d)[C], float pen)
{
#pragma unroll
    for (int c = 0; c < C; ++c) d[c] = d[c] * pen;
}

// One unordered pair (a < b) of a tile.  Frames of sequence s live at d_frames[(seq_off[s] + t) * dpad], t in
// [0, len), followed by two sentinel frames used for columns j <= 0 (index len: zero components, norm +INF;
// index len + 1: +INF components).
struct PairInfo {
    const float *A, *B;
    int n, m, w;
    int slot_a, slot_b;
    uint32_t slab_tile;     // position of the tile in the rank's slab
    float nmax_b;           // largest squared frame norm of sequence b
    float nmax_ab;          // ... of either sequence (kernels that may take their columns from a or b)
    bool valid;
};

__device__ __forceinline__ PairInfo decode_pair(const AlignLaunch &L, uint32_t tile, uint32_t slot)
{
    PairInfo p;
    p.slot_a = slot / kTile;
    p.slot_b = slot % kTile;
    p.valid = false;
    p.slab_tile = 0;
    p.nmax_b = 0.0f;
    p.nmax_ab = 0.0f;
    p.A = p.B = L.d_frames; p.n = p.m = 2; p.w = 2;
    if (tile >= L.n_tiles) return p;
    const uint4 t = L.d_tiles[tile];
    p.slab_tile = t.z;
    const uint32_t a = t.x * kTile + p.slot_a, b = t.y * kTile + p.slot_b;
    if (!((a < b) && (b < L.n_seq))) return p;
    p.valid = true;
    const uint32_t oa = L.d_seq_off[a], ob = L.d_seq_off[b];
    p.n = (int)(L.d_seq_off[a + 1] - oa) - 2;
    p.m = (int)(L.d_seq_off[b + 1] - ob) - 2;
    p.A = L.d_frames + (uint64_t)oa * L.dpad;
    p.B = L.d_frames + (uint64_t)ob * L.dpad;
    p.w = pair_w(L.band, p.n, p.m);
    p.nmax_b = L.d_seq_nmax[b];
    p.nmax_ab = fmaxf(L.d_seq_nmax[a], p.nmax_b);
    return p;
}

__device__ __forceinline__ void store_pair(const AlignLaunch &L, uint32_t, const PairInfo &p, float s1, float s2)
{
    float *slab = L.d_slab + (uint64_t)p.slab_tile * 2 * kSlotsPerTile;
    slab[p.slot_a * kTile + p.slot_b] = s1;                       // score(x=a, y=b)
    slab[kSlotsPerTile + p.slot_a * kTile + p.slot_b] = s2;       // score(x=b, y=a)
}

// Host launchers of the templated systolic kernel, one translation unit per frame dimension.
// Returns false when (G, C) is not instantiated.
template <int D>
bool launch_systolic(const AlignLaunch &L, int g, int c, bool uniform_pen, hipStream_t stream);   // L.hybrid selects the distance form
extern template bool launch_systolic<8>(const AlignLaunch &, int, int, bool, hipStream_t);
extern template bool launch_systolic<10>(const AlignLaunch &, int, int, bool, hipStream_t);
extern template bool launch_systolic<13>(const AlignLaunch &, int, int, bool, hipStream_t);
extern template bool launch_systolic<16>(const AlignLaunch &, int, int, bool, hipStream_t);
extern template bool launch_systolic<20>(const AlignLaunch &, int, int, bool, hipStream_t);
extern template bool launch_systolic<26>(const AlignLaunch &, int, int, bool, hipStream_t);

// ... and of its strict-mode instantiations <D, C, G, unit penalties, difference form>, in units of their own (dtw_sysx_d<D>.hip).
template <int D>
bool launch_systolic_strict(const AlignLaunch &L, int g, int c, hipStream_t stream);
extern template bool launch_systolic_strict<8>(const AlignLaunch &, int, int, hipStream_t);
extern template bool launch_systolic_strict<10>(const AlignLaunch &, int, int, hipStream_t);
extern template bool launch_systolic_strict<13>(const AlignLaunch &, int, int, hipStream_t);
extern template bool launch_systolic_strict<16>(const AlignLaunch &, int, int, hipStream_t);
extern template bool launch_systolic_strict<20>(const AlignLaunch &, int, int, hipStream_t);
extern template bool launch_systolic_strict<26>(const AlignLaunch &, int, int, hipStream_t);

// Wide-band kernel (dtw_wide.h): NW waves per pair; returns false when (NW, C) is not instantiated.
template <int D>
bool launch_wide(const AlignLaunch &L, int nw, int c, hipStream_t stream, hipError_t *err);
extern template bool launch_wide<8>(const AlignLaunch &, int, int, hipStream_t, hipError_t *);
extern template bool launch_wide<10>(const AlignLaunch &, int, int, hipStream_t, hipError_t *);
extern template bool launch_wide<13>(const AlignLaunch &, int, int, hipStream_t, hipError_t *);
extern template bool launch_wide<16>(const AlignLaunch &, int, int, hipStream_t, hipError_t *);
extern template bool launch_wide<20>(const AlignLaunch &, int, int, hipStream_t, hipError_t *);
extern template bool launch_wide<26>(const AlignLaunch &, int, int, hipStream_t, hipError_t *);

// Full-matrix kernel (dtw_full.h): column strips, one DP for both ordered pairs.
template <int D>
bool launch_full(const AlignLaunch &L, bool banded, int ppw, int cw, hipStream_t stream, hipError_t *err);
extern template bool launch_full<8>(const AlignLaunch &, bool, int, int, hipStream_t, hipError_t *);
extern template bool launch_full<10>(const AlignLaunch &, bool, int, int, hipStream_t, hipError_t *);
extern template bool launch_full<13>(const AlignLaunch &, bool, int, int, hipStream_t, hipError_t *);
extern template bool launch_full<16>(const AlignLaunch &, bool, int, int, hipStream_t, hipError_t *);
extern template bool launch_full<20>(const AlignLaunch &, bool, int, int, hipStream_t, hipError_t *);
extern template bool launch_full<26>(const AlignLaunch &, bool, int, int, hipStream_t, hipError_t *);

}  // namespace apd

// Percentile threshold + UPGMA on the GPU (reference src/numerics.rs:125-133, src/clustering.rs:81-210).
//
// percentile: the reference sorts all n*n values and indexes the sorted array; here a 4-pass radix SELECT
//   (8 bits per pass, MSB first) finds the same element with 4 streaming reads of the matrix.
// clustering: the reference re-derives every average linkage from the raw pair distances at every merge
//   (O(c^2 n + c n^2) per merge, clustering.rs:153-187).  Here a directed cluster-SUM matrix S stays resident in HBM:
//   linkage(p, q) = S[p][q] / (|p| * |q|).  Only the row and the column of a freshly merged cluster change, and they
//   are recomputed from the raw distances in EXACTLY the reference's order (x ascending over Cp, y ascending over Cq,
//   one running f32 accumulator, clustering.rs:157-169) from sorted member lists, so every linkage -- and with it the
//   whole merge sequence, including the p/q order of mathematically tied directed pairs that the reference settles
//   by rounding noise -- is bit-identical to the literal algorithm.  Each merge is THREE launches (round 3: eight), TWO while no
//   merge needs the third (a second captured batch without it, see apd_clustering):
//     upgma_select_kernel   row minima of the rows whose cached best pair went stale (a compact list), then -- in the
//                           workgroup that finishes last -- the global arg-min, merge_clusters, the merged member list, and
//                           one 32-byte record of the merge for the other two launches;
//     upgma_chain_kernel    every chain of the new cluster's row and column of S that one lane or one wavefront sums
//                           whole; long chains get their segments allocated (atomic bump) and predicted here;
//     upgma_segment_kernel  the segments' integer maps, and -- in the wavefront that finishes a chain's last segment --
//                           the in-order commit of that chain.
//   No workgroup ever waits for another one: "last to arrive does the serial part" (an atomic counter and a fence), so the
//   kernels cannot deadlock whatever the scheduler does.  Between two replays of a batch the host may have the working copies
//   of d re-laid out so that a cluster's members are neighbouring rows and columns (upgma_permute_kernel): which elements are
//   added, and in which order, never changes -- only where they lie.  Exact ties resolve to the lowest (id_p, id_q): what the
//   reference does when its HashSet happens to iterate in ascending order (clustering.rs:180-187); any other order is
//   equally "reference".
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>

#include "apd_internal.h"

using namespace apd;

namespace {

#define HIP_TRY(ctx, call)                                                             \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e_);     \
            return e_ == hipErrorOutOfMemory ? APD_ERR_OOM : APD_ERR_HIP;              \
        }                                                                              \
    } while (0)

// ---------------------------------------------------------------------------------- radix select

__device__ __forceinline__ uint32_t order_key(float v)
{
    const uint32_t u = __builtin_bit_cast(uint32_t, v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // ascending key order == ascending float order
}

// hist[d] = #{non-NaN x : key(x) has `prefix` in the bits above shift+8 and digit d at `shift`}; hist[256] = #NaN
__global__ void select_hist_kernel(const float *__restrict__ x, uint64_t len, uint32_t prefix, uint32_t prefix_mask, int shift,
                                   unsigned long long *__restrict__ hist)
{
    __shared__ unsigned int lh[257];
    for (int t = threadIdx.x; t < 257; t += blockDim.x) lh[t] = 0;
    __syncthreads();
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < len; e += (uint64_t)gridDim.x * blockDim.x) {
        const float v = x[e];
        if (v != v) { atomicAdd(&lh[256], 1u); continue; }
        const uint32_t k = order_key(v);
        if ((k & prefix_mask) == prefix) atomicAdd(&lh[(k >> shift) & 0xFFu], 1u);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < 257; t += blockDim.x)
        if (lh[t]) atomicAdd(&hist[t], (unsigned long long)lh[t]);
}

}  // namespace

namespace apd {
// numerics.rs:125-133 on a device array.  k = (len as f32 * perc) as usize is computed by the caller.
int device_select(apd_context *ctx, const float *d_x, uint64_t len, uint64_t k, float *value)
{
    int rc = APD_OK;
    unsigned long long *d_hist = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&d_hist, 257 * sizeof(unsigned long long)));
    uint32_t prefix = 0, mask = 0;
    uint64_t rank = k;
    unsigned long long h[257];
    const int blocks = (int)std::min<uint64_t>((len + 255) / 256, 4096);
    for (int pass = 0; pass < 4 && rc == APD_OK; ++pass) {
        const int shift = 24 - 8 * pass;
        hipError_t e = hipMemsetAsync(d_hist, 0, 257 * sizeof(unsigned long long), ctx->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(select_hist_kernel, dim3(std::max(blocks, 1)), dim3(256), 0, ctx->stream, d_x, len, prefix, mask, shift, d_hist);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(h, d_hist, sizeof(h), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); rc = APD_ERR_HIP; break; }
        if (pass == 0) {
            const uint64_t non_nan = len - h[256];                // numbers.len() after the NaN filter (:127-130)
            if (k >= non_nan) { rc = APD_ERR_INDEX; break; }      // numbers[n as usize] panics (:132)
        }
        uint64_t acc = 0;
        int d = 0;
        for (; d < 256; ++d) { if (rank < acc + h[d]) break; acc += h[d]; }
        if (d == 256) { rc = APD_ERR_INDEX; break; }
        rank -= acc;
        prefix |= (uint32_t)d << shift;
        mask |= 0xFFu << shift;
    }
    hipFree(d_hist);
    if (rc != APD_OK) return rc;
    const uint32_t u = (prefix & 0x80000000u) ? (prefix & 0x7FFFFFFFu) : ~prefix;   // invert order_key
    std::memcpy(value, &u, sizeof(float));
    return APD_OK;
}

uint64_t percentile_index(uint64_t len, float perc)
{
    const float nf = (float)len * perc;                           // numerics.rs:126 (f32 product of the unfiltered length)
    if (!(nf > 0.0f)) return 0;                                   // `as usize` saturates
    if (nf >= 18446744073709551616.0f) return UINT64_MAX;
    return (uint64_t)nf;
}
}  // namespace apd

namespace {

// ---------------------------------------------------------------------------------------- UPGMA

// Agent-scope relaxed accesses (sc1 on gfx950: the store goes through to the point every XCD sees, the load does not stop at this CU's
// L1 or this XCD's L2).  What one wavefront hands to another INSIDE a launch -- segment results, packed copies -- travels through these,
// so that the hand-over needs no agent-scope FENCE: a release fence writes the whole XCD's L2 back, and two thousand of them per
// merge (one per segment) were 40 % of the segment launch.
template <typename T> __device__ __forceinline__ void store_agent(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ T load_agent(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct Cand { float l; uint32_t idp, idq, sp, sq; };
// one segment of a long chain (see "the new cluster's row and column of S"): the predicted running sum at its start, the exponent
// es the maps were computed under (0: none), and for each of its four SUB-BLOCKS (the shares of lanes 0-15, 16-31, 32-47, 48-63) the
// map S -> S + (S odd ? a1 : a0) under es (a0 / a1) and under es + 1 (b0 / b1): when the running sum changes binade inside the
// segment, the commit walks the one sub-block that holds the change and goes on with the es + 1 maps -- a quarter of a segment
// re-walked per binade change instead of the whole of it.  A map that cannot be used holds kCap.
struct SegRes { float predicted; uint32_t es, pack, pad; uint32_t a0[4], a1[4], b0[4], b1[4]; };   // pack: offset of the segment's contiguous copy, or ~0; pad: see kExactSegment

__device__ __forceinline__ bool better(const Cand &a, const Cand &b)
{
    // strict '<' in ascending (id_p, id_q) iteration order (clustering.rs:184): first minimum wins; NaN never wins
    if (!(a.l < __builtin_inff())) return false;      // min_linkage starts at +INF: an infinite linkage is never taken
    if (a.l < b.l) return true;
    if (a.l == b.l) return (a.idp < b.idp) || (a.idp == b.idp && a.idq < b.idq);
    return false;
}

// Reductions of the select launch: a candidate is packed into ONE 64-bit key whose unsigned order is the order `better` defines,
// reduced inside the wavefront by shuffles and across wavefronts by one LDS atomic -- two barriers instead of the eleven of a tree.
// order_key() is monotone over the floats; +0.0 is added first so that -0.0 and +0.0 tie as they do under `<`.
__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long k)
{
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned int lo = (unsigned int)__shfl_xor((int)(unsigned int)k, o), hi = (unsigned int)__shfl_xor((int)(unsigned int)(k >> 32), o);
        const unsigned long long other = ((unsigned long long)hi << 32) | lo;
        k = other < k ? other : k;
    }
    return k;
}

// What the chain and segment launches need to know about the merge they serve, written by the select launch's bookkeeping as ONE
// 32-byte record: a single load instead of a chain of dependent ones (last_sp -> mstart / mcount -> ...).  sp == 0xFFFFFFFF: no merge
// is waiting for its chains (the loop has ended, or the degenerate (0, 0) merge was emitted).
struct MergeRec { uint32_t sp, sq, K, ms, nl, nb, t, pad; };   // new cluster's slot, dead slot, |Ck|, offset of Ck's lists, live slots, big clusters, n_ops

struct UpgmaState {
    const float *d;           // [n][n] raw distances as the caller laid them out (read only): threshold and the initial S
    const float *const *mat;  // device words: [0] the WORKING copy of d, [1] of its transpose -- rows and columns in `phys` order (the caller's
                              // order until the first defragmentation, see upgma_permute_kernel); [2] non-null once defragmented
    uint32_t *ppool;          // the member lists again, same offsets as `pool`, as PHYSICAL rows / columns of the working copies
    uint32_t *phys;           // [n] instance -> its physical row / column
    uint32_t *dsrc, *doff;    // [n] defragmentation scratch: new physical index -> old one; first new index of the i-th live cluster
    float *S;                 // [n][n] directed cluster sums (slot-indexed)
    uint32_t *pool;           // sorted member lists, appended per merge
    uint32_t *mstart, *mcount;// [n] list of the cluster held by a slot
    uint32_t *pool_used;
    uint32_t *last_sp;        // slot of the cluster created by the latest merge (0xFFFFFFFF: none)
    float *size;              // [n] member counts as f32 (the reference counts in f32, clustering.rs:154-166); 0: the slot is dead
    uint32_t *id;             // [n] cluster id held by the slot
    uint32_t *live;           // [n_live] slots still holding a root (order irrelevant); pos[slot] = its index in here
    uint32_t *pos;
    uint32_t *n_live;
    Cand *rbest;              // [n] cached best ordered pair of the row held by a slot
    float *rb_l;              // [n] its linkage alone (+INF for a dead slot): what the global arg-min streams over
    uint32_t *big;            // [n_big] live slots holding a cluster of two or more members (order irrelevant); bpos[slot] = index or ~0
    uint32_t *bpos;
    uint32_t *n_big;
    uint32_t *rscan;          // [n] 1: the cache of this row is stale and the slot is on the stale list
    uint32_t *stale;          // [n] slots whose row must be re-scanned by the next select launch
    uint32_t *n_stale;
    uint32_t *arrive;         // workgroups of the running select launch that have finished their rows
    uint32_t *r_pending;      // 1: R[.][last_sp] += R[.][last_sq] of the latest merge has not been applied yet
    uint32_t *last_sq;        // slot that died in the latest merge
    uint32_t *rec;            // MergeRec of the latest merge (32-byte aligned)
    uint32_t *items_total;    // segments made so far (wraps; the host looks at differences): does the segment launch have anything to do?
    uint32_t *whole_walks;    // long chains walked whole by the chain launch of a two-launch batch (see no_seg)
    uint32_t no_seg;          // 1: this launch belongs to a batch WITHOUT segment launches
    apd_cluster_op *ops;      // [n]
    uint32_t *n_ops;
    uint32_t *work;           // sum of the new clusters' member counts so far (wraps; the host looks at differences): when to defragment
    float *R;                 // [n][n] R[slot][x]: approximate sum of d[x][y] over the members y of the cluster in `slot` (a ROW per slot: a
                              // merge adds one row to another, 2 n contiguous floats, where a column per slot cost n scattered cache lines)
    uint32_t *item_start;     // [2 n] first work item (segment) of a segmented chain of the current merge
    uint32_t *item_chain;     // [max_items] chain of every work item
    uint32_t short_chain;     // chains up to this many elements are walked whole by one wavefront
    uint32_t *seg_done;       // [2 n] finished segments of a segmented chain
    float *packed;            // contiguous copies of the segments the commit pass is likely to re-walk (nullptr: off)
    uint32_t *pack_used;      // bump allocator of `packed`, reset per merge
    uint32_t pack_capacity;   // floats
    SegRes *seg;              // [max_items] segment results of the current merge
    uint32_t *n_items;
    uint32_t *done;           // set once the loop condition of clustering.rs:104 fails
    unsigned long long *dbg;  // APD_DEBUG_UPGMA_TIMING: [n][16] wall_clock64 stamps per merge (nullptr: off)
    float threshold;
    uint32_t n;
};

// Launch 1 of a merge.  Every workgroup: (a) its share of R[x][sp'] += R[x][sq'] of the PREVIOUS merge (R only predicts
// binades: nothing in this launch reads it); (b) row minima of the stale rows it draws from the compact stale list -- a
// direct scan over all n slots (S row, size, id: coalesced, independent loads; dead slots have size 0), no indirection
// through the live list.  The workgroup that arrives last then does what used to be a launch of its own: global arg-min over
// the cached row minima, merge_clusters (clustering.rs:134-141), the merged member list, the op record.
__global__ __launch_bounds__(1024) void upgma_select_kernel(UpgmaState st)
{
    __shared__ Cand win;
    __shared__ uint32_t is_last, kwin, lmin_key;
    __shared__ float kwin_l;
    __shared__ unsigned long long kmin;
    const uint32_t n = st.n;
    const unsigned long long t_entry = st.dbg ? wall_clock64() : 0ull;
    // every scalar this workgroup needs, loaded together (one memory round trip, not one per dependent step)
    const uint32_t finished = *st.done, pending = *st.r_pending, sp0 = *st.last_sp, sq0 = *st.last_sq, ns = *st.n_stale;
    const uint32_t first_stale = st.stale[min(blockIdx.x, n - 1)];       // meaningful if blockIdx.x < ns
    if (finished != 0u) return;
    if (pending != 0u) {
        // the R update is dealt from the far end of the grid: the stale rows start at workgroup 0
        float *rp = st.R + (uint64_t)sp0 * n;
        const float *rq = st.R + (uint64_t)sq0 * n;
        for (uint32_t x = (gridDim.x - 1u - blockIdx.x) * blockDim.x + threadIdx.x; x < n; x += gridDim.x * blockDim.x) rp[x] = rp[x] + rq[x];
    }
    for (uint32_t r = blockIdx.x; r < ns; r += gridDim.x) {
        const uint32_t sp = r == blockIdx.x ? first_stale : st.stale[r];
        Cand best{__builtin_inff(), 0xFFFFFFFFu, 0xFFFFFFFFu, sp, sp};
        const float size_p = st.size[sp];
        const uint32_t idp = st.id[sp];
        const float *row = st.S + (uint64_t)sp * n;
        // the row's best column, tracked as three scalars (linkage, id of the column's cluster, column slot): idp and sp are the
        // row's own.  Same order as `better`: smaller linkage, then smaller idq; +INF and NaN linkages never win.
        float bl = __builtin_inff();
        uint32_t bidq = 0xFFFFFFFFu, bsq = sp;
        auto consider = [&](uint32_t c, float v, float sz, uint32_t idq) __attribute__((always_inline)) {
            if (c >= n || c == sp || sz == 0.0f) return;                  // target_i != target_j (clustering.rs:182); dead slots
            const float l = v / (size_p * sz);                            // size_x * size_y (:169)
            const bool take = (l < __builtin_inff()) && (l < bl || (l == bl && idq < bidq));
            bl = take ? l : bl; bidq = take ? idq : bidq; bsq = take ? c : bsq;
        };
        if ((n & 3u) == 0u) {
            // 16-byte loads, every load of the row issued before the first use: one memory round trip per 16 K columns
            constexpr uint32_t kVec = 4;
            const float4 *row4 = reinterpret_cast<const float4 *>(row), *sz4 = reinterpret_cast<const float4 *>(st.size);
            const uint4 *id4 = reinterpret_cast<const uint4 *>(st.id);
            const uint32_t n4 = n >> 2;
            for (uint32_t q0 = threadIdx.x; q0 < n4; q0 += kVec * blockDim.x) {
                float4 v[kVec], z[kVec];
                uint4 i[kVec];
#pragma unroll
                for (uint32_t u = 0; u < kVec; ++u) { const uint32_t q = min(q0 + u * blockDim.x, n4 - 1); v[u] = row4[q]; z[u] = sz4[q]; i[u] = id4[q]; }
#pragma unroll
                for (uint32_t u = 0; u < kVec; ++u) {
                    const uint32_t q = q0 + u * blockDim.x;
                    if (q >= n4) continue;
                    consider(4 * q + 0, v[u].x, z[u].x, i[u].x); consider(4 * q + 1, v[u].y, z[u].y, i[u].y);
                    consider(4 * q + 2, v[u].z, z[u].z, i[u].z); consider(4 * q + 3, v[u].w, z[u].w, i[u].w);
                }
            }
        } else {
            constexpr uint32_t kIlp = 4;
            for (uint32_t c0 = threadIdx.x; c0 < n; c0 += kIlp * blockDim.x) {
                float v[kIlp], sz[kIlp];
                uint32_t idq[kIlp];
#pragma unroll
                for (uint32_t u = 0; u < kIlp; ++u) {
                    const uint32_t c = min(c0 + u * blockDim.x, n - 1);
                    v[u] = row[c]; sz[u] = st.size[c]; idq[u] = st.id[c];
                }
#pragma unroll
                for (uint32_t u = 0; u < kIlp; ++u) consider(c0 + u * blockDim.x, v[u], sz[u], idq[u]);
            }
        }
        // (linkage, idq) as one key: smaller linkage first, then smaller idq -- `better` within a row; idq is unique in a row, so
        // exactly one thread holds the winning key and publishes the column and the linkage that go with it
        if (threadIdx.x == 0) { kmin = ~0ull; kwin = sp; kwin_l = __builtin_inff(); }
        __syncthreads();
        const unsigned long long key = bl < __builtin_inff() ? ((unsigned long long)order_key(bl + 0.0f) << 32) | bidq : ~0ull;
        const unsigned long long wkey = wave_min_u64(key);
        if ((threadIdx.x & 63) == 0 && wkey != ~0ull) atomicMin(&kmin, wkey);
        __syncthreads();
        if (key != ~0ull && key == kmin) { kwin = bsq; kwin_l = bl; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const Cand found = kmin != ~0ull ? Cand{kwin_l, idp, (uint32_t)kmin, sp, kwin} : best;
            // (agent-scope stores: what the last workgroup to arrive reads of this one travels past this XCD's L2 by itself, see below)
            Cand *o = st.rbest + sp;
            store_agent(&o->l, found.l); store_agent(&o->idp, found.idp); store_agent(&o->idq, found.idq); store_agent(&o->sp, found.sp); store_agent(&o->sq, found.sq);
            store_agent(&st.rb_l[sp], found.l); store_agent(&st.rscan[sp], 0u);
        }
    }
    // ---- who is last?  (the classic fence + counter: every workgroup's writes above are visible to the one that sees the full count)
    const unsigned long long t_rows = st.dbg ? wall_clock64() : 0ull;
    // What the last workgroup reads of the others -- the row minima -- was written by thread 0 with agent-scope stores, so arriving
    // needs no agent-scope RELEASE fence (on a multi-XCD part that writes the XCD's L2 back: sixteen wavefronts doing it each was most
    // of this launch's 34 us at n = 4096, one per workgroup still 1.5 us): thread 0 waits for its own stores, then counts itself in.
    // Only the workgroup that finds itself last pays an acquire (its L1 / L2 may hold older copies of those lines).  The R updates
    // above are plain stores: nothing in this launch reads them, the end of the launch publishes them.
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        is_last = atomicAdd(st.arrive, 1u) == gridDim.x - 1u ? 1u : 0u;
        if (is_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (is_last == 0u) return;
    const unsigned long long t_last = st.dbg ? wall_clock64() : 0ull;
    if (threadIdx.x == 0) { *st.items_total += *st.n_items; *st.arrive = 0u; *st.n_stale = 0u; *st.r_pending = 0u; *st.n_items = 0u; *st.pack_used = 0u; }
    // global arg-min, two passes over the linkages alone (coalesced, independent loads): the smallest value, then `better` among
    // the rows that hold it (its tie rule needs their ids; almost always a single row)
    if (threadIdx.x == 0) { lmin_key = 0xFFFFFFFFu; kmin = ~0ull; win = Cand{__builtin_inff(), 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0}; }
    __syncthreads();
    float lmin = __builtin_inff();
    constexpr uint32_t kAhead = 8;                                        // loads in flight per thread
    float v0[kAhead];                                                     // the first 8192 rows' linkages stay in the registers for the second pass
#pragma unroll
    for (uint32_t u = 0; u < kAhead; ++u) v0[u] = st.rb_l[min(threadIdx.x + u * blockDim.x, n - 1)];   // a clamped repeat changes no minimum
#pragma unroll
    for (uint32_t u = 0; u < kAhead; ++u) lmin = __builtin_fminf(lmin, v0[u]);                  // NaN never wins (fminf drops it)
    for (uint32_t c0 = threadIdx.x + kAhead * blockDim.x; c0 < n; c0 += kAhead * blockDim.x) {
        float v[kAhead];
#pragma unroll
        for (uint32_t u = 0; u < kAhead; ++u) v[u] = st.rb_l[min(c0 + u * blockDim.x, n - 1)];
#pragma unroll
        for (uint32_t u = 0; u < kAhead; ++u) lmin = __builtin_fminf(lmin, v[u]);
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) lmin = __builtin_fminf(lmin, __shfl_xor(lmin, o));
    if ((threadIdx.x & 63) == 0 && lmin < __builtin_inff()) atomicMin(&lmin_key, order_key(lmin + 0.0f));
    __syncthreads();
    {   // invert order_key: the smallest linkage of the matrix, or +INF if no row has a finite one
        const uint32_t kk = lmin_key;
        lmin = kk == 0xFFFFFFFFu ? __builtin_inff() : __builtin_bit_cast(float, (kk & 0x80000000u) ? (kk & 0x7FFFFFFFu) : ~kk);
    }
    // ... then, among the rows that hold it, the lowest (idp, idq) -- tracked as scalars, and published by the thread that holds the
    // winner (no read-back of its record: one dependent round trip less).  (Carrying a whole Cand through
    // `if (better(cc, best)) best = cc` here was compiled into a partial assignment: (l, idp, idq) of one row with (sp, sq) of another
    // when a thread met two tied rows -- caught by the tie-laden matrices of tests/test_gpu_clustering.py.)
    uint32_t bidp = 0xFFFFFFFFu, bidq = 0xFFFFFFFFu, brow = 0xFFFFFFFFu, bsq2 = 0u;
    float bl2 = __builtin_inff();
    if (lmin < __builtin_inff())
        for (uint32_t c0 = threadIdx.x; c0 < n; c0 += kAhead * blockDim.x) {
            float v[kAhead];
#pragma unroll
            for (uint32_t u = 0; u < kAhead; ++u) v[u] = c0 == threadIdx.x ? v0[u] : st.rb_l[min(c0 + u * blockDim.x, n - 1)];   // second pass: registers, then L2 hits
#pragma unroll
            for (uint32_t u = 0; u < kAhead; ++u) {
                const uint32_t c = c0 + u * blockDim.x;
                if (c < n && v[u] == lmin) {
                    const Cand *o = st.rbest + c;
                    const uint32_t ip = o->idp, iq = o->idq, osq = o->sq;
                    const float ol = o->l;                                // (lmin up to the sign of a zero: the record's own bits are reported)
                    const bool take = ip < bidp || (ip == bidp && iq < bidq);
                    bidp = take ? ip : bidp; bidq = take ? iq : bidq; brow = take ? c : brow; bsq2 = take ? osq : bsq2; bl2 = take ? ol : bl2;
                }
            }
        }
    {   // (idp, idq) as one key; idp is unique per row, so exactly one thread holds the winning key and publishes its row
        const unsigned long long key = brow != 0xFFFFFFFFu ? ((unsigned long long)bidp << 32) | bidq : ~0ull;
        const unsigned long long wkey = wave_min_u64(key);
        if ((threadIdx.x & 63) == 0 && wkey != ~0ull) atomicMin(&kmin, wkey);
        __syncthreads();
        if (key != ~0ull && key == kmin) win = Cand{bl2, bidp, bidq, brow, bsq2};
        __syncthreads();
    }
    const unsigned long long t_argmin = st.dbg ? wall_clock64() : 0ull;
    const Cand w = win;
    const uint32_t t = *st.n_ops;
    const uint32_t k = n + t;                                             // parents.len() (:135)
    const uint32_t nl = *st.n_live;
    if (w.idp == 0xFFFFFFFFu) {
        // no linkage below +INF: the reference keeps min_merge = (0, 0) (clustering.rs:179) and merges it
        if (threadIdx.x == 0) {
            st.ops[t] = apd_cluster_op{0u, 0u, k, __builtin_inff(), (uint32_t)APD_SEQUENCE2SEQUENCE};
            *st.n_ops = t + 1;
            *st.last_sp = 0xFFFFFFFFu;
            st.rec[0] = 0xFFFFFFFFu;                                      // no chains to sum
            *st.done = 2;                                                 // INF < threshold is false: loop ends (:104)
        }
        return;
    }
    // new cluster k lives in slot sp; slot sq dies.  Everything the bookkeeping needs is loaded up front (independent loads: one
    // round trip instead of a dozen dependent ones in a single thread).
    const uint32_t msp = st.mstart[w.sp], msq = st.mstart[w.sq], cp = st.mcount[w.sp], cq = st.mcount[w.sq], used = *st.pool_used;
    const float zp = st.size[w.sp], zq = st.size[w.sq];
    const uint32_t at = st.pos[w.sq], tail = st.live[nl - 1];
    const uint32_t nb0 = *st.n_big, bps = st.bpos[w.sp], bq = st.bpos[w.sq], last_big = st.big[max(nb0, 1u) - 1u], work0 = *st.work;
    // merge the two sorted member lists into a fresh one
    {
        const uint32_t *lp = st.pool + msp, *lq = st.pool + msq, *pp = st.ppool + msp, *pq = st.ppool + msq;
        uint32_t *out = st.pool + used, *pout = st.ppool + used;          // the order is decided by the instance numbers; the physical list follows
        for (uint32_t i = threadIdx.x; i < cp + cq; i += blockDim.x) {
            const bool from_p = i < cp;
            const uint32_t v = from_p ? lp[i] : lq[i - cp], pv = from_p ? pp[i] : pq[i - cp];
            const uint32_t *other = from_p ? lq : lp;
            uint32_t lo = 0, hi = from_p ? cq : cp;                       // members are distinct: plain lower bound
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (other[mid] < v) lo = mid + 1; else hi = mid; }
            out[(from_p ? i : i - cp) + lo] = v; pout[(from_p ? i : i - cp) + lo] = pv;
        }
    }
    __syncthreads();                                                      // every wavefront has read what thread 0 is about to overwrite
    if (threadIdx.x == 0) {
        if (st.dbg) {
            unsigned long long *g = st.dbg + (uint64_t)t * 16;
            g[0] = t_entry; g[1] = t_rows; g[2] = t_last; g[3] = t_argmin; g[4] = wall_clock64(); g[5] = ns; g[6] = cp + cq;
        }
        uint32_t op;
        if (w.idp < n && w.idq < n) op = APD_SEQUENCE2SEQUENCE;           // clustering.rs:193-201
        else if (w.idp >= n && w.idq >= n) op = APD_CLUSTER2CLUSTER;
        else if (w.idp >= n && w.idq < n) op = APD_CLUSTER2SEQUENCE;
        else op = APD_SEQUENCE2CLUSTER;
        st.ops[t] = apd_cluster_op{w.idp, w.idq, k, w.l, op};
        *st.n_ops = t + 1;
        *st.work = work0 + cp + cq;
        st.size[w.sp] = zp + zq;
        st.size[w.sq] = 0.0f;                                             // dead: skipped by every scan
        st.rb_l[w.sq] = __builtin_inff();
        {   // the clusters of two or more members: sp joins (if new), sq leaves (if it was one)
            // (operands loaded up front with the others: the list's last entry is the one read there unless sp has just been appended)
            uint32_t nb = nb0;
            const bool joins = bps == 0xFFFFFFFFu;
            if (joins) { st.big[nb] = w.sp; st.bpos[w.sp] = nb; ++nb; }
            if (bq != 0xFFFFFFFFu) { const uint32_t tb = joins ? w.sp : last_big; st.big[bq] = tb; st.bpos[tb] = bq; st.bpos[w.sq] = 0xFFFFFFFFu; --nb; }
            *st.n_big = nb;
        }
        st.id[w.sp] = k;
        st.mstart[w.sp] = used; st.mcount[w.sp] = cp + cq; *st.pool_used = used + cp + cq;
        *st.last_sp = w.sp; *st.last_sq = w.sq;
        *st.r_pending = 1u;
        // drop sq from the live list (its order is irrelevant): the tail moves into its place
        st.live[at] = tail; st.pos[tail] = at;
        *st.n_live = nl - 1;
        st.rscan[w.sp] = 1;                                               // the new cluster's row is new
        st.stale[0] = w.sp;
        *st.n_stale = 1u;
        const bool ends = nl - 1 <= 1 || !(w.l < st.threshold);           // while n_clusters > 1 && distance < threshold (:104)
        if (ends) *st.done = 1;
        // the record the chain / segment launches read (two 16-byte stores); a merge that ends the loop leaves no chains to sum
        const uint32_t nb_now = *st.n_big;
        reinterpret_cast<uint4 *>(st.rec)[1] = uint4{nl - 1, nb_now, t + 1, 0u};
        reinterpret_cast<uint4 *>(st.rec)[0] = uint4{ends ? 0xFFFFFFFFu : w.sp, w.sq, cp + cq, used};
    }
}

// ---- the exact-order f32 sum, evaluated in parallel -------------------------------------------------------------------
// linkage() (clustering.rs:157-169) adds |Cp| * |Cq| distances into ONE f32 accumulator, x ascending, y ascending; every
// add rounds, so the order is part of the result and a tree sum would change bits (and with them which of two nearly
// tied cluster pairs merges first).  The chain is evaluated exactly, yet in parallel:
//
// While the running sum s stays inside one binade [2^e, 2^(e+1)), s = S * ulp with S an integer in [2^23, 2^24) and
// ulp = 2^(e-23).  Adding x >= 0 gives fl(s + x) = RNE((S + x / ulp)) * ulp: with x / ulp = X + f (X integer, 0 <= f < 1),
// S' = S + X + (f > 1/2) + (f == 1/2 and S + X odd).  So one element acts on S as  S -> S + A[S mod 2]  with two integer
// offsets (A[0], A[1]) that differ only on an exact half-way tie.  Such maps are closed under composition,
//      (F then G)[p] = F[p] + G[(p + F[p]) mod 2],
// and composition is associative: a lane composes its own run of consecutive elements, a 64-lane prefix scan composes the
// lanes in order, and S advances by up to 64 * kRun elements per step with integer arithmetic only.  The scan also tells
// WHERE the sum first reaches 2^24 (leaves the binade): everything before that lane is committed, that lane's elements
// are added one by one with real f32 adds (which handles the change of ulp, however many binades the sum jumps), and the
// walk resumes behind it in the new binade.  A sum of n like-sized terms changes binade ~log2(n) times, so almost every
// step is the parallel one.  Elements that are negative, NaN or -0.0 (penalties <= 0 can produce them), and sums that are
// zero, subnormal, negative or non-finite, take the literal path: 64 gathered values added one by one (v_readlane + add).
// The result is the reference's bit pattern in every case; tests/test_gpu_clustering.py compares bits against the CPU
// oracle on matrices up to N = 4500, including tie-laden integer matrices, +INF blocks and NaN entries.
namespace exact {

constexpr uint32_t kCap = 1u << 26;              // saturation of the offsets: anything >= 2^24 already means "left the binade"

struct Fn { uint32_t a0, a1; };                  // S -> S + (S odd ? a1 : a0)

__device__ __forceinline__ Fn compose(const Fn f, const Fn g)   // f first, then g
{
    Fn h;
    h.a0 = min(f.a0 + ((f.a0 & 1u) ? g.a1 : g.a0), kCap);
    h.a1 = min(f.a1 + ((f.a1 & 1u) ? g.a0 : g.a1), kCap);       // parity after f from an odd S is (1 + f.a1) mod 2
    return h;
}

// The map of one element x (raw bits xb of a finite, non-negative float) on the integer mantissa of a sum whose biased
// exponent is es (>= 1).  An element in the sum's own binade or above returns kCap: the sum leaves the binade.
__device__ __forceinline__ Fn element(uint32_t xb, uint32_t es)
{
    uint32_t ex = xb >> 23, mx = xb & 0x7FFFFFu;
    if (ex) mx |= 0x800000u; else ex = 1u;       // subnormal x: no hidden bit, exponent of the smallest normal
    Fn f;
    if (ex >= es) { f.a0 = f.a1 = (mx == 0u) ? 0u : kCap; return f; }   // x == +0.0 is the identity
    const uint32_t sh = min(es - ex, 25u);       // >= 25: x < ulp / 2, the add is absorbed
    const uint32_t X = mx >> sh, r = mx & ((1u << sh) - 1u), half = 1u << (sh - 1u);
    const uint32_t up = r > half ? 1u : 0u, tie = r == half ? 1u : 0u;
    f.a0 = X + (up | (tie & (X & 1u)));          // S even: S + X has the parity of X
    f.a1 = X + (up | (tie & (~X & 1u)));         // S odd
    return f;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Fn dpp_fn(const Fn v)  // lanes without a source (or outside ROW_MASK) get the identity map
{
    Fn r;
    r.a0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.a0, CTRL, ROW_MASK, 0xf, false);
    r.a1 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.a1, CTRL, ROW_MASK, 0xf, false);
    return r;
}

// inclusive prefix composition over the 64 lanes, lane order = element order
__device__ __forceinline__ Fn wave_scan(Fn v)
{
    v = compose(dpp_fn<0x111, 0xf>(v), v);       // row_shr:1
    v = compose(dpp_fn<0x112, 0xf>(v), v);       // row_shr:2
    v = compose(dpp_fn<0x114, 0xf>(v), v);       // row_shr:4
    v = compose(dpp_fn<0x118, 0xf>(v), v);       // row_shr:8   -> prefix inside each row of 16
    v = compose(dpp_fn<0x142, 0xa>(v), v);       // row_bcast:15 into rows 1, 3
    v = compose(dpp_fn<0x143, 0xc>(v), v);       // row_bcast:31 into rows 2, 3
    return v;
}

__device__ __forceinline__ float readlane_f(float v, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// Where a chain's elements come from: gathered from the distance matrix (element e = a * cy + b is d[lx[a] * n + ly[b]]),
// or from the packed copy a segment wavefront left behind (contiguous, so that re-walking a segment costs 1/16 of the
// cache lines of the gather).
// The working copies are reached through pointers that were themselves loaded from memory (st.mat[]): to the compiler they could point
// anywhere (FLAT loads, which count on both wait counters and so cannot be waited for one by one).  Global address space, said aloud:
typedef const float __attribute__((address_space(1))) *GlobalF;
struct GatherSrc {
    const float *m; uint64_t sx, sy;                                    // element (a, b) is m[lx[a] * sx + ly[b] * sy]
    const uint32_t *lx; uint32_t cx; const uint32_t *ly; uint32_t cy;
    template <int K>
    __device__ __forceinline__ void load_run(uint64_t first, uint64_t total, float (&x)[K]) const
    {
        uint32_t a = 0, b = 0;
        if (first < total) {
            if (total <= 0xFFFFFFFFull) { a = (uint32_t)first / cy; b = (uint32_t)first - a * cy; }          // wave-uniform: the cheap division
            else { a = (uint32_t)(first / cy); b = (uint32_t)(first - (uint64_t)a * cy); }
        }
        // Indices first (no memory), then every list entry, then every element: TWO dependent round trips per run whatever cy is.
        // (Written with a branch per element -- "next row?" -- the compiler waits for each load before the next: sixteen trips.)
        uint32_t aa[K], bb[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            aa[j] = min(a, cx - 1); bb[j] = b;                          // b < cy always; a runs past cx - 1 only behind `total`
            const bool wrap = ++b == cy;
            b = wrap ? 0u : b; a += wrap ? 1u : 0u;
        }
        uint32_t ra[K], cb[K];
#pragma unroll
        for (int j = 0; j < K; ++j) { ra[j] = lx[aa[j]]; cb[j] = ly[bb[j]]; }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const float v = ((GlobalF)m)[(uint64_t)ra[j] * sx + (uint64_t)cb[j] * sy];
            x[j] = first + j < total ? v : 0.0f;                        // padding: +0.0 is the identity
        }
    }
    __device__ __forceinline__ float load_one(uint64_t e, uint64_t total) const
    {
        uint32_t a, b;
        if (total <= 0xFFFFFFFFull) { a = (uint32_t)e / cy; b = (uint32_t)e - a * cy; }
        else { a = (uint32_t)(e / cy); b = (uint32_t)(e - (uint64_t)a * cy); }
        return ((GlobalF)m)[(uint64_t)lx[a] * sx + (uint64_t)ly[b] * sy];
    }
};
struct PackedSrc {
    const float *p;                                                     // p[e]: element e of the chain
    template <int K>
    __device__ __forceinline__ void load_run(uint64_t first, uint64_t total, float (&x)[K]) const
    {
#pragma unroll
        for (int j = 0; j < K; ++j) {                                   // written by another wavefront of this launch; branch-free, so the K loads
            const float v = load_agent(p + min<uint64_t>(first + j, total - 1));   // are in flight together (total >= 1 wherever a walk runs)
            x[j] = first + j < total ? v : 0.0f;
        }
    }
    __device__ __forceinline__ float load_one(uint64_t e, uint64_t) const { return load_agent(p + e); }
};

// Elements [pos, total) of a chain are added, in order, into the one f32 accumulator `s` -- by one wavefront; every lane
// returns the same value.  (pos = 0, s = +0.0, total = cx * cy is linkage()'s whole sum.)  kRun = consecutive elements per
// lane and step: 64 * kRun elements advance per step, kRun loads per lane are in flight at a time.
template <int kRun, typename Src>
__device__ float ordered_walk(const Src src, uint32_t lane, uint64_t pos, uint64_t total, float s)
{
    while (pos < total) {
        const uint32_t sb = __builtin_bit_cast(uint32_t, s);
        const uint32_t es = sb >> 23;                                   // sign + exponent: 1..254 <=> positive, normal, finite
        const bool fast = es >= 1u && es <= 254u;
        const bool tiny = es == 0u;                                     // +0.0 (every chain starts there) or a positive subnormal
        if ((fast || tiny || sb == 0x7F800000u) && total - pos > 64) {  // a remainder of <= 64 elements: one literal round is cheaper
            float x[kRun];                                              // this lane's run: elements pos + lane * kRun + j
            src.template load_run<kRun>(pos + (uint64_t)lane * kRun, total, x);
            bool bad = false;                                           // negative (incl. -0.0) or NaN: not a map on S
            Fn f{0u, 0u};
#pragma unroll
            for (int j = 0; j < kRun; ++j) {
                const uint32_t xb = __builtin_bit_cast(uint32_t, x[j]);
                bad |= xb > 0x7F800000u;
                f = compose(f, element(xb, es));
            }
            if (bad) f.a0 = f.a1 = kCap;                                // sends the walk into this lane's run, one real add at a time
            const uint64_t step = min<uint64_t>(total - pos, 64ull * kRun);
            if (!fast && !tiny) {
                // s == +INF stays +INF under non-negative finite or infinite terms; a NaN or a negative term (-INF) ends that
                const unsigned long long bm = __ballot(bad);
                if (bm == 0ull) { pos += step; continue; }
                const int L = __builtin_ctzll(bm);
#pragma unroll
                for (int j = 0; j < kRun; ++j) s = s + readlane_f(x[j], L);
                pos = min<uint64_t>(pos + (uint64_t)(L + 1) * kRun, total);
                continue;
            }
            // The loaded elements stay in registers while the sum climbs through binades: a lane whose run takes the sum out of its
            // binade is added with real f32 adds, and the lanes behind it are re-mapped under the new exponent and scanned again --
            // no reload.  (A sum that starts at zero changes binade ten times inside its first thousand elements: reloading after
            // each change made a short chain a dozen dependent memory round trips long.)
            uint32_t start = 0;                                          // lanes below `start` are consumed
            for (;;) {
                const uint32_t sb2 = __builtin_bit_cast(uint32_t, s), es2 = sb2 >> 23;
                if (es2 == 0u) {                                         // no integer map on a zero / subnormal sum: this lane's run by real adds,
                    const int L0 = (int)start;                           // then on with the lanes behind it -- a chain's start costs no load of its own
#pragma unroll
                    for (int j = 0; j < kRun; ++j) s = s + readlane_f(x[j], L0);
                    if (++start >= 64u) break;
                    continue;
                }
                if (es2 > 254u) break;                                   // negative, +INF or NaN now: the outer loop's other paths
                if (start != 0u) {                                       // re-map under the current exponent, consumed lanes as the identity
                    f = Fn{0u, 0u};
                    if (lane >= start) {
#pragma unroll
                        for (int j = 0; j < kRun; ++j) f = compose(f, element(__builtin_bit_cast(uint32_t, x[j]), es2));
                        if (bad) f.a0 = f.a1 = kCap;
                    }
                }
                const Fn pre = wave_scan(f);
                const uint32_t S = (sb2 & 0x7FFFFFu) | 0x800000u;
                const uint32_t adv = (S & 1u) ? pre.a1 : pre.a0;        // S advances by this much up to and including this lane
                const unsigned long long leaves = __ballot(S + adv >= (1u << 24));
                if (leaves == 0ull) {
                    const uint32_t Sn = S + (uint32_t)__builtin_amdgcn_readlane((int)adv, 63);
                    s = __builtin_bit_cast(float, (es2 << 23) | (Sn & 0x7FFFFFu));
                    start = 64u;
                    break;
                }
                const int L = __builtin_ctzll(leaves);                  // first lane whose run takes the sum out of the binade (>= start)
                if (L > 0 && (uint32_t)L > start) {
                    const uint32_t Sn = S + (uint32_t)__builtin_amdgcn_readlane((int)adv, L - 1);
                    s = __builtin_bit_cast(float, (es2 << 23) | (Sn & 0x7FFFFFu));
                }
#pragma unroll
                for (int j = 0; j < kRun; ++j) s = s + readlane_f(x[j], L);  // real adds: any rounding regime, any jump
                start = (uint32_t)L + 1u;
                if (start >= 64u) break;
            }
            pos = min<uint64_t>(pos + (uint64_t)start * kRun, total);
        } else {
            // literal path (s is zero, subnormal, negative or NaN, or the chain is about to end): 64 elements, one real add each
            const uint64_t e = pos + lane;
            const float v = e < total ? src.load_one(e, total) : 0.0f;
            const uint32_t cnt = (uint32_t)min<uint64_t>(64, total - pos);
            if (cnt == 64) {
#pragma unroll
                for (int t = 0; t < 64; ++t) s = s + readlane_f(v, t);
            } else {
                for (uint32_t t = 0; t < cnt; ++t) s = s + readlane_f(v, (int)t);
            }
            pos += cnt;
        }
    }
    return s;
}

// The maps of elements [begin, end) of the chain on the integer mantissa of a sum whose biased exponent is `es` (f) or `es + 1` (g)
// and stays so: every lane composes its own contiguous share (no cross-lane traffic until the ordered reduction at the end), the
// sixteen lanes of a DPP row compose into that sub-block's map.  {kCap, kCap} where an element cannot be expressed (negative, NaN, or
// large enough to leave the binade by itself).  The elements are also written to pack[0 .. end - begin) for whoever has to re-walk.
struct BlockMaps { Fn f[4], g[4]; };
__device__ BlockMaps segment_fn(const GatherSrc src, uint32_t lane, uint64_t begin, uint64_t end, uint32_t es, float *__restrict__ pack)
{
    const uint64_t len = end - begin, share = (len + 63) / 64;
    uint64_t e = begin + share * lane;
    const uint64_t stop = min<uint64_t>(e + share, end);
    Fn f{0u, 0u}, g{0u, 0u};
    bool bad = false;
    const bool up_ok = es + 1u <= 254u;                                  // es + 1 = 255 is +INF: no integer map there
    constexpr int kUnroll = 8;                                           // loads of 8 elements in flight per lane
    while (e < stop) {
        float x[kUnroll];
        src.load_run<kUnroll>(e, stop, x);
        const uint32_t cnt = (uint32_t)min<uint64_t>(kUnroll, stop - e);
#pragma unroll
        for (int j = 0; j < kUnroll; ++j) {
            const uint32_t xb = __builtin_bit_cast(uint32_t, x[j]);
            bad |= xb > 0x7F800000u;
            f = compose(f, element(xb, es));                             // padding (+0.0) is the identity
            g = compose(g, element(xb, es + 1u));
            if (pack && (uint32_t)j < cnt) store_agent(pack + (e - begin) + j, x[j]);
        }
        e += cnt;
    }
    if (bad) f.a0 = f.a1 = kCap;
    if (bad || !up_ok) g.a0 = g.a1 = kCap;
    // inclusive prefix inside each row of 16 lanes: lane 15 of a row holds the row's composition, in order
    auto row_scan = [](Fn v) __attribute__((always_inline)) {
        v = compose(dpp_fn<0x111, 0xf>(v), v);   // row_shr:1
        v = compose(dpp_fn<0x112, 0xf>(v), v);   // row_shr:2
        v = compose(dpp_fn<0x114, 0xf>(v), v);   // row_shr:4
        v = compose(dpp_fn<0x118, 0xf>(v), v);   // row_shr:8
        return v;
    };
    f = row_scan(f);
    g = row_scan(g);
    BlockMaps out;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        out.f[b].a0 = (uint32_t)__builtin_amdgcn_readlane((int)f.a0, 16 * b + 15); out.f[b].a1 = (uint32_t)__builtin_amdgcn_readlane((int)f.a1, 16 * b + 15);
        out.g[b].a0 = (uint32_t)__builtin_amdgcn_readlane((int)g.a0, 16 * b + 15); out.g[b].a1 = (uint32_t)__builtin_amdgcn_readlane((int)g.a1, 16 * b + 15);
    }
    return out;
}

}  // namespace exact

// ---- the new cluster's row and column of S -----------------------------------------------------------------------------
// After a merge, S[k][r] and S[r][k] are re-summed for every other live cluster r: 2 (n_live - 1) chains of |Ck| * |Cr|
// elements each.  The longest of them is the critical path of a merge (10^6 elements once two clusters of ~1000 members
// face each other), so a long chain is cut into SEGMENTS of whole rows (~kSegElems elements) that are evaluated by
// different wavefronts at the same time -- speculatively, because a segment's integer map (exact::segment_fn) needs the
// binade of the running sum at the segment's start.  That binade is predicted from R[x][slot], an approximate
// (any-order) sum of d[x][y] over the members y of the cluster in `slot`, kept up to date per merge: the prefix of the
// row sums estimates the running sum at every row boundary.  One wavefront per chain then walks the segments in order
// with the TRUE sum: a segment whose assumed exponent matches and whose map keeps the sum inside the binade is applied in
// O(1); any other segment (the first one, the ~log2(L) binade changes, a rare misprediction near a power of two, segments
// with negative or NaN entries) is re-walked element by element from the true sum by exact::ordered_walk.  The result is
// bit-identical to the sequential chain either way; the speculation only decides how fast it is reached.
namespace {

constexpr uint32_t kSegElems = 4096;             // target elements per segment (2048 .. 16384 measured: 4096 / 8192 within 2 % of each other, 1024 twice as slow)
constexpr uint32_t kShortChain = 2 * kSegElems;  // other chains up to this length are walked whole by one wavefront
constexpr uint32_t kLaneChain = 2048;            // a SINGLETON against a new cluster of up to this many members: one lane, the literal loop


struct Chain {
    const uint32_t *lx, *ly;  // members as instance numbers (rows of R); plx / ply: the same members as physical rows / columns
    const uint32_t *plx, *ply;
    uint32_t cx, cy;
    uint32_t slot_y;        // slot of the cluster the inner index runs over (its column of R predicts the row sums)
    uint32_t s, dir;        // the other cluster's slot; dir 0: S[sp][s], dir 1: S[s][sp]
    uint32_t rps, nseg;     // rows per segment, segments (0: skipped chain, 1: walked whole)
    bool lane_chain;        // a singleton's chain: summed by one lane of the chain launch's group wavefronts
};

// Chain id w = 2 * (slot of the other cluster) + direction: no trip through the live list to find out what a chain is.
__device__ __forceinline__ MergeRec load_rec(const UpgmaState &st)
{
    const uint4 a = reinterpret_cast<const uint4 *>(st.rec)[0], b = reinterpret_cast<const uint4 *>(st.rec)[1];
    return MergeRec{a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
}

__device__ __forceinline__ Chain get_chain(const UpgmaState &st, uint32_t w, const MergeRec &m)
{
    Chain c{};
    c.dir = w & 1u;
    c.s = w >> 1;
    if (c.s == m.sp) return c;                                           // nseg = 0
    const uint32_t ms_s = st.mstart[c.s], cnt_s = st.mcount[c.s];        // the new cluster's own list comes with the record
    const uint32_t mx = c.dir ? ms_s : m.ms, my = c.dir ? m.ms : ms_s;
    c.lx = st.pool + mx; c.ly = st.pool + my; c.plx = st.ppool + mx; c.ply = st.ppool + my;
    c.cx = c.dir ? cnt_s : m.K; c.cy = c.dir ? m.K : cnt_s;
    c.slot_y = c.dir ? m.sp : c.s;
    const uint64_t len = (uint64_t)c.cx * c.cy;
    if (len <= kLaneChain && (c.dir ? c.cx : c.cy) == 1u) { c.rps = c.cx; c.nseg = 1; c.lane_chain = true; return c; }
    if (len <= st.short_chain) { c.rps = c.cx; c.nseg = 1; return c; }
    c.rps = c.cy >= kSegElems ? 1u : kSegElems / c.cy;
    c.nseg = (c.cx + c.rps - 1) / c.rps;
    return c;
}

// S[sp][s] chains (dir 0) walk rows of d.  S[s][sp] chains (dir 1) walk, in the caller's layout, columns of d restricted to the new
// cluster -- one float per cache line, lines that no other chain uses -- so they read the transposed copy there, where the same
// elements lie in the new cluster's |Ck| rows.  Once the working copies are DEFRAGMENTED (a cluster's members are neighbouring
// rows and columns, upgma_permute_kernel) the inner index of both directions runs along a row of d and dir 1 reads d as well.
__device__ __forceinline__ exact::GatherSrc chain_src(const UpgmaState &st, const Chain &c)
{
    const float *d = st.mat[0], *dT = st.mat[1];
    if (c.dir == 0 || st.mat[2] != nullptr) return exact::GatherSrc{d, st.n, 1, c.plx, c.cx, c.ply, c.cy};
    return exact::GatherSrc{dT, 1, st.n, c.plx, c.cx, c.ply, c.cy};
}

// what lane 0 does with a finished sum (the tail of the former one-kernel update).  Its operands -- the row's cached best pair, the
// sizes and ids of the candidate -- do not depend on the sum: load_finish() requests them BEFORE the sum's own loads, so the end of a
// chain is stores only instead of two more dependent round trips on the launch's critical path.
struct FinishOps { Cand old; float sz_s, sz_sp; uint32_t id_s, id_sp; };
__device__ __forceinline__ FinishOps load_finish(const UpgmaState &st, uint32_t s, uint32_t dir, uint32_t sp)
{
    FinishOps f{};
    if (dir) { f.old = st.rbest[s]; f.sz_s = st.size[s]; f.sz_sp = st.size[sp]; f.id_s = st.id[s]; f.id_sp = st.id[sp]; }
    return f;
}
__device__ __forceinline__ void finish_chain(const UpgmaState &st, uint32_t s, uint32_t dir, const MergeRec &m, float acc, const FinishOps &f)
{
    if (dir) {
        st.S[(uint64_t)s * st.n + m.sp] = acc;
        // row s: its cached best pair survives unless it pointed at one of the two merged slots; the new entry may beat it
        if (f.old.sq == m.sp || f.old.sq == m.sq) {
            if (atomicExch(&st.rscan[s], 1u) == 0u) st.stale[atomicAdd(st.n_stale, 1u)] = s;   // re-scanned by the next select launch
        } else {
            const Cand cnd{acc / (f.sz_s * f.sz_sp), f.id_s, f.id_sp, s, m.sp};
            if (better(cnd, f.old)) { st.rbest[s] = cnd; st.rb_l[s] = cnd.l; }
        }
    } else st.S[(uint64_t)m.sp * st.n + s] = acc;
}

}  // namespace

// The predicted running sum at every segment start of a segmented chain, from the row sums in R (one wavefront).  R's update
// for the latest merge is applied by the NEXT select launch, so the new cluster's column is read as the sum of its two halves.
__device__ void predict_chain(const UpgmaState &st, const Chain &c, uint32_t sp, uint32_t sq, uint32_t lane, SegRes *res)
{
    const bool fresh = c.slot_y == sp;
    float base = 0.0f;
    for (uint32_t a0 = 0; a0 < c.cx; a0 += 64) {
        const uint32_t a = a0 + lane;
        float v = 0.0f;
        if (a < c.cx) {
            const uint32_t x = c.lx[a];
            v = fresh ? st.R[(uint64_t)sp * st.n + x] + st.R[(uint64_t)sq * st.n + x] : st.R[(uint64_t)c.slot_y * st.n + x];
        }
        float incl = v;                                                  // inclusive scan over the lanes (any rounding will do)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const float t = __shfl_up(incl, o); if ((int)lane >= o) incl += t; }
        if (a < c.cx && a % c.rps == 0) {
            const float before = base + (incl - v);                      // sum of the rows before row a
            res[a / c.rps].predicted = before;
        }
        base += __shfl(incl, 63);
    }
}

// Launch 2 of a merge: every chain of the new cluster's row and column.
//  * wavefronts [0, 2 ceil(n / 64)): SINGLETONS against the new cluster (most chains of a merge, |Ck| terms each): ONE LANE per
//    chain runs linkage()'s loop literally -- one f32 accumulator, members of the new cluster ascending -- so nothing about the
//    order needs proving.  The 64 lanes of a wavefront hold 64 neighbouring singletons in one direction.  In both directions the
//    matrix ROW of element i is member i of the NEW cluster (a row of d for S[sp][s], a row of the transposed copy for S[s][sp])
//    and the column is the singleton: the row sequence is wave-uniform (the member list goes through scalar loads), every load
//    instruction reads neighbouring floats of one row -- whole cache lines instead of one line per float -- and 32 loads are in
//    flight per lane.
//  * wavefronts behind them, one per chain w: a chain of up to kShortChain elements is walked whole (exact::ordered_walk) and
//    finished; a longer one reserves its nseg work items with one atomic add (their order among chains is irrelevant), records
//    which chain they belong to, and writes the predicted sums the segment launch needs.
__global__ __launch_bounds__(256) void upgma_chain_kernel(UpgmaState st)
{
    const uint32_t wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t group_waves = 2u * ((st.n + 63u) / 64u);
    // First round trip: the merge record, and with it the one list entry whose ADDRESS does not depend on the record (a singleton
    // group's live slot, a chain wavefront's big cluster) -- requested together.
    const bool grp = wid < group_waves;
    const uint32_t v = grp ? 0u : wid - group_waves, cidx = (wid >> 1) * 64u + lane;
    const uint32_t spec = grp ? st.live[min(cidx, st.n - 1u)] : st.big[min(v >> 1, st.n - 1u)];
    const MergeRec m = load_rec(st);
    const uint32_t sp = m.sp;
    if (sp == 0xFFFFFFFFu) return;
    auto stamp = [&]() __attribute__((always_inline)) {                  // tuning aid: when the last wavefront with work finished
        if (st.dbg && lane == 0) atomicMax(st.dbg + (uint64_t)(m.t - 1u) * 16 + 7, (unsigned long long)wall_clock64());
    };
    if (grp) {
        const uint32_t dir = wid & 1u, K = m.K;
        if ((wid >> 1) * 64u >= m.nl) return;
        // A singleton's slot holds the instance of the same number (slots are instance ids until they merge), so its chain needs no
        // member-list lookups.  Second round trip: is it a singleton, its physical column, what the end of the chain needs (the
        // row's cached best pair, the sizes and ids of its candidate) and the new cluster's member list; third: the elements.
        const uint32_t own = cidx < m.nl ? spec : sp;
        const uint32_t cnt_own = st.mcount[own];
        const uint64_t col = st.phys[own];
        const FinishOps fo = load_finish(st, own, dir, sp);
        const bool active = own != sp && K <= kLaneChain && cnt_own == 1u;
        if (__ballot(active) == 0ull) return;
        // element i is M[ck[i] * n + phys[own]]
        const uint32_t *ck = st.ppool + m.ms;                           // wave-uniform (physical rows)
        const exact::GlobalF M = (exact::GlobalF)st.mat[dir];          // dir 1: the transposed copy
        const uint64_t rs = st.n;
        float s = 0.0f;                                                  // distance = 0.0 (clustering.rs:154)
        if (K <= 4u) {                                                   // most merges: no point in issuing 32 loads for two members
            float x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = M[(uint64_t)ck[min((uint32_t)u, K - 1)] * rs + col];
#pragma unroll
            for (int u = 0; u < 4; ++u) if ((uint32_t)u < K) s = s + x[u];
        } else {
            constexpr int kAhead = 32;
            for (uint32_t i0 = 0; i0 < K; i0 += kAhead) {
                float x[kAhead];
#pragma unroll
                for (int u = 0; u < kAhead; ++u) {
                    const uint32_t r = ck[min(i0 + u, K - 1)];          // uniform address: a scalar load
                    x[u] = M[(uint64_t)r * rs + col];
                }
#pragma unroll
                for (int u = 0; u < kAhead; ++u) if (i0 + u < K) s = s + x[u];   // distance += d[x][y] (:162), in order
            }
        }
        if (active) finish_chain(st, own, dir, m, s, fo);
        stamp();
        return;
    }
    // chains [0, 2 n_big): against the clusters of two or more members; behind them, only when the NEW cluster is too large for one
    // lane per singleton (more than kLaneChain members), one chain per singleton and direction.  The grid holds one chain wavefront per
    // possible chain (2 n), so the loop below runs once; it is a loop so that a smaller grid stays correct (measured: no gain).
    const uint32_t nb2 = 2u * m.nb, n_chains = nb2 + (m.K > kLaneChain ? 2u * m.nl : 0u);
    const uint32_t chain_waves = ((gridDim.x * blockDim.x) >> 6) - group_waves;
    for (uint32_t vv = v; vv < n_chains; vv += chain_waves) {
    uint32_t w;
    if (vv < nb2) w = 2u * (vv == v ? spec : st.big[vv >> 1]) + (vv & 1u);
    else {
        const uint32_t s1 = st.live[(vv - nb2) >> 1];
        if (st.mcount[s1] != 1u) continue;                               // covered by the first range
        w = 2u * s1 + ((vv - nb2) & 1u);
    }
    if ((w >> 1) == sp) continue;
    const FinishOps fo = load_finish(st, w >> 1, w & 1u, sp);            // second round trip, together with get_chain's two loads
    const Chain c = get_chain(st, w, m);
    if (c.nseg == 0 || c.lane_chain) continue;
    if (c.nseg == 1) {
        const float acc = exact::ordered_walk<8>(chain_src(st, c), lane, 0, (uint64_t)c.cx * c.cy, 0.0f);
        if (lane == 0) finish_chain(st, c.s, c.dir, m, acc, fo);
        stamp();
        continue;
    }
    if (st.no_seg) {
        // a batch without segment launches (the host leaves them out while no merge needs them: a launch boundary costs ~4 us whatever
        // the launch does): a long chain is walked whole right here -- exact as ever, slow, and counted, so that the host goes back to
        // three launches for the next batch
        const float acc = exact::ordered_walk<8>(chain_src(st, c), lane, 0, (uint64_t)c.cx * c.cy, 0.0f);
        if (lane == 0) { finish_chain(st, c.s, c.dir, m, acc, fo); atomicAdd(st.whole_walks, 1u); }
        stamp();
        continue;
    }
    uint32_t base = 0;
    if (lane == 0) { base = atomicAdd(st.n_items, c.nseg); st.item_start[w] = base; st.seg_done[w] = 0u; }
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    // The chain's segments are items base .. base + nseg - 1; wavefront v of launch 3 takes item v.  (Until the end of round 4 every
    // segment also went onto the work list of ONE XCD, chosen by where in the source matrix it reads -- a locality hint that measured
    // as a non-effect against round-robin dealing and cost an atomic per segment on eight addresses, two thousand per merge.)
    for (uint32_t j = lane; j < c.nseg; j += 64) st.item_chain[base + j] = w;
    predict_chain(st, c, sp, m.sq, lane, st.seg + base);
    stamp();
    }
}

// Launch 3 of a merge: the segments of the long chains.  A segment gets its integer map under the predicted exponent
// (exact::segment_fn).  The wavefronts of an XCD take the items of that XCD's work list (filled by launch 2) in turn.  A segment is
// also copied to `packed`, contiguously, when the commit is likely to re-walk it: the predicted sum changes binade inside it or
// sits within 2^-10 of a power of two.
// The wavefront that finishes the LAST segment of a chain (atomic counter behind a fence) commits the chain: the true sum through
// the segments, in order.  The segments' maps are themselves composed by a 64-lane prefix scan (lane t = segment j0 + t, each under
// the map that fits the sum's current binade: its es map, its es + 1 map, or none): one scan carries the sum up to the first segment
// whose map does not keep it inside the binade -- a change of binade, or a misprediction -- and only THAT segment is taken apart:
// its four sub-block maps one by one, the sub-block that holds the change walked element by element (from the packed copy if the
// segment's wavefront made one); then the scan resumes behind it under the new exponent.  A chain of 71 segments changes binade
// five or six times behind its first segment: six scans and six walked sub-blocks instead of 284 dependent map applications.
// The FIRST segment starts from +0.0 exactly, so its wavefront has already walked it for real: kExactSegment, the sum in `pad`.
constexpr uint32_t kExactSegment = 0xFFFFFFFFu;

__device__ void commit_chain(const UpgmaState &st, const Chain &c, const MergeRec &m, uint32_t lane, const SegRes *res, uint32_t *rewalks = nullptr)
{
    const FinishOps fo = load_finish(st, c.s, c.dir, m.sp);              // requested with the first batch of segment results
    uint32_t n_rewalk = 0;
    const uint64_t total = (uint64_t)c.cx * c.cy, seg_len = (uint64_t)c.rps * c.cy;
    float s = 0.0f;
    for (uint32_t j0 = 0; j0 < c.nseg; j0 += 64) {
        // lane t holds segment j0 + t: its exponent, its packed copy, and the sub-block maps under es (a) and es + 1 (b)
        uint32_t m_es = 0u, m_pack = 0xFFFFFFFFu, m_exact = 0u, ma0[4], ma1[4], mb0[4], mb1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) ma0[q] = ma1[q] = mb0[q] = mb1[q] = exact::kCap;
        if (j0 + lane < c.nseg) {
            const SegRes *r = res + j0 + lane;
            m_es = load_agent(&r->es); m_pack = load_agent(&r->pack); m_exact = load_agent(&r->pad);
#pragma unroll
            for (int q = 0; q < 4; ++q) { ma0[q] = load_agent(&r->a0[q]); ma1[q] = load_agent(&r->a1[q]); mb0[q] = load_agent(&r->b0[q]); mb1[q] = load_agent(&r->b1[q]); }
        }
        const uint32_t cnt = min(64u, c.nseg - j0);
        // the whole-segment maps: the four sub-blocks composed, in order (kCap saturates: an unusable sub-block makes an unusable segment)
        exact::Fn F{ma0[0], ma1[0]}, G{mb0[0], mb1[0]};
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            // (a sub-block behind the end of a short last segment holds the identity)
            F = exact::compose(F, exact::Fn{ma0[q], ma1[q]}); G = exact::compose(G, exact::Fn{mb0[q], mb1[q]});
        }
        uint32_t t = 0;
        while (t < cnt) {
            const uint32_t sb = __builtin_bit_cast(uint32_t, s), e = sb >> 23;
            // lanes [t, cnt): the map that fits the current binade; consumed lanes and lanes behind the chain's end: the identity
            exact::Fn h{0u, 0u};
            if (lane >= t && lane < cnt) {
                const bool usable = m_es >= 1u && m_es <= 254u && e >= 1u && e <= 254u;
                h = (usable && e == m_es) ? F : (usable && e == m_es + 1u) ? G : exact::Fn{exact::kCap, exact::kCap};
            }
            const exact::Fn pre = exact::wave_scan(h);
            const uint32_t S = (sb & 0x7FFFFFu) | 0x800000u;
            const uint32_t adv = (S & 1u) ? pre.a1 : pre.a0;
            const unsigned long long leaves = __ballot(S + adv >= (1u << 24));
            if (leaves == 0ull) {                                         // (e is normal here: a zero / subnormal / non-finite sum makes every map kCap)
                const uint32_t Sn = S + (uint32_t)__builtin_amdgcn_readlane((int)adv, 63);
                s = __builtin_bit_cast(float, (e << 23) | (Sn & 0x7FFFFFu));
                break;
            }
            const uint32_t L = (uint32_t)__builtin_ctzll(leaves);         // >= t: the consumed lanes hold the identity
            if (L > t) {
                const uint32_t Sn = S + (uint32_t)__builtin_amdgcn_readlane((int)adv, (int)L - 1);
                s = __builtin_bit_cast(float, (e << 23) | (Sn & 0x7FFFFFu));
            }
            // ---- segment j0 + L taken apart
            const uint32_t es_j = (uint32_t)__builtin_amdgcn_readlane((int)m_es, (int)L);
            if (es_j == kExactSegment) {                                  // walked for real by its own wavefront (the chain's first segment)
                s = __builtin_bit_cast(float, (uint32_t)__builtin_amdgcn_readlane((int)m_exact, (int)L));
                t = L + 1u;
                continue;
            }
            const uint32_t pack_off = (uint32_t)__builtin_amdgcn_readlane((int)m_pack, (int)L);
            const uint64_t begin = (uint64_t)(j0 + L) * seg_len, end = min<uint64_t>(begin + seg_len, total);
            const uint64_t share = (end - begin + 63) / 64;               // as segment_fn dealt the segment to its lanes
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint64_t qb = begin + 16ull * q * share, qe = min<uint64_t>(begin + 16ull * (q + 1) * share, end);
                if (qb >= end) break;
                const uint32_t tb = __builtin_bit_cast(uint32_t, s), eq = tb >> 23;
                if (es_j >= 1u && es_j <= 254u && (eq == es_j || eq == es_j + 1u)) {
                    const bool up = eq != es_j;                           // the sum is one binade above the prediction: the es + 1 maps
                    const uint32_t a0 = (uint32_t)__builtin_amdgcn_readlane((int)(up ? mb0[q] : ma0[q]), (int)L);
                    const uint32_t a1 = (uint32_t)__builtin_amdgcn_readlane((int)(up ? mb1[q] : ma1[q]), (int)L);
                    const uint32_t Sq = (tb & 0x7FFFFFu) | 0x800000u;
                    const uint32_t Sn = Sq + ((Sq & 1u) ? a1 : a0);
                    if (Sn < (1u << 24)) { s = __builtin_bit_cast(float, (eq << 23) | (Sn & 0x7FFFFFu)); continue; }
                }
                // walk this sub-block from the true sum: from the packed copy if the segment's wavefront made one
                ++n_rewalk;
                if (pack_off != 0xFFFFFFFFu)                              // PackedSrc indexes by chain element: rebased to the segment
                    s = exact::ordered_walk<8>(exact::PackedSrc{st.packed + pack_off - begin}, lane, qb, qe, s);   // (16 per lane: the kernel spills)
                else
                    s = exact::ordered_walk<8>(chain_src(st, c), lane, qb, qe, s);
            }
            t = L + 1u;
        }
    }
    if (rewalks) *rewalks = n_rewalk;
    if (lane == 0) finish_chain(st, c.s, c.dir, m, s, fo);
}

__global__ __launch_bounds__(256) void upgma_segment_kernel(UpgmaState st)
{
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t v = blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = gridDim.x * 4u;
    // one round trip: the item count (zero for most merges: this launch is then over), the merge record, this wavefront's item's chain
    const uint32_t n_items = *st.n_items, w_first = st.item_chain[v];   // (item_chain holds max_items >= the grid's wavefronts entries)
    const MergeRec m = load_rec(st);
    const uint32_t sp = m.sp;
    if (sp == 0xFFFFFFFFu || n_items == 0u) return;
    for (uint32_t item = v; item < n_items; item += n_waves) {
        const uint32_t w = item == v ? w_first : st.item_chain[item];
        const Chain c = get_chain(st, w, m);
        const exact::GatherSrc src = chain_src(st, c);
        const uint32_t first = st.item_start[w], j = item - first;
        SegRes *res = st.seg + item;
        const uint32_t pb = __builtin_bit_cast(uint32_t, res->predicted), es = pb >> 23;
        exact::BlockMaps maps;
#pragma unroll
        for (int q = 0; q < 4; ++q) maps.f[q] = maps.g[q] = exact::Fn{exact::kCap, exact::kCap};
        uint32_t pack_off = 0xFFFFFFFFu, es_out = (es >= 1u && es <= 254u) ? es : 0u, exact_bits = 0u;
        if (j == 0u) {
            // the chain's first segment starts from +0.0, exactly: nothing to predict, it is walked for real right here
            const uint64_t end0 = min<uint64_t>((uint64_t)c.rps * c.cy, (uint64_t)c.cx * c.cy);
            exact_bits = __builtin_bit_cast(uint32_t, exact::ordered_walk<8>(src, lane, 0, end0, 0.0f));
            es_out = kExactSegment;
        } else if (es >= 1u && es <= 254u) {
            const uint64_t begin = (uint64_t)j * c.rps * c.cy, end = min<uint64_t>(begin + (uint64_t)c.rps * c.cy, (uint64_t)c.cx * c.cy);
            const uint32_t len = (uint32_t)(end - begin);
            // will the commit re-walk this segment?  (the next segment's prediction is this segment's predicted end)
            const uint32_t nb = j + 1 < c.nseg ? __builtin_bit_cast(uint32_t, res[1].predicted) : 0xFFFFFFFFu;
            const uint32_t frac = pb & 0x7FFFFFu, nfrac = nb & 0x7FFFFFu;
            const bool risky = (nb >> 23) != es || frac < (1u << 13) || nfrac > 0x7FFFFFu - (1u << 13);
            if (risky && st.packed) {
                if (lane == 0) {
                    pack_off = atomicAdd(st.pack_used, len);
                    if ((uint64_t)pack_off + len > st.pack_capacity) pack_off = 0xFFFFFFFFu;
                }
                pack_off = (uint32_t)__builtin_amdgcn_readfirstlane((int)pack_off);
            }
            maps = exact::segment_fn(src, lane, begin, end, es, pack_off != 0xFFFFFFFFu ? st.packed + pack_off : nullptr);
        }
        if (st.dbg && lane == 0 && item == 0u) st.dbg[(uint64_t)(m.t - 1u) * 16 + 10] = n_items;
        uint32_t finished = 0;
        if (lane < 4) {                                                  // lane q publishes sub-block q's maps (uniform values: any lane holds them)
            uint32_t a0 = maps.f[0].a0, a1 = maps.f[0].a1, b0 = maps.g[0].a0, b1 = maps.g[0].a1;
#pragma unroll
            for (int q = 1; q < 4; ++q) if ((int)lane == q) { a0 = maps.f[q].a0; a1 = maps.f[q].a1; b0 = maps.g[q].a0; b1 = maps.g[q].a1; }
            store_agent(&res->a0[lane], a0); store_agent(&res->a1[lane], a1); store_agent(&res->b0[lane], b0); store_agent(&res->b1[lane], b1);
            if (lane == 0) { store_agent(&res->es, es_out); store_agent(&res->pack, pack_off); store_agent(&res->pad, exact_bits); }
        }
        // this segment's map and packed copy (agent-scope stores of every lane) have been performed ...
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");           // (a wait for this wavefront's stores; no cache write-back)
        if (lane == 0) finished = atomicAdd(&st.seg_done[w], 1u) + 1u;   // ... then the count
        finished = (uint32_t)__builtin_amdgcn_readfirstlane((int)finished);
        if (finished == c.nseg) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const unsigned long long t0 = st.dbg ? wall_clock64() : 0ull;
            uint32_t rw = 0;
            commit_chain(st, c, m, lane, st.seg + first, &rw);          // reads the other wavefronts' results with agent-scope loads
            if (st.dbg && lane == 0) {                                   // tuning aid, stamped by the committing wavefronts only (a few dozen per merge)
                unsigned long long *g = st.dbg + (uint64_t)(m.t - 1u) * 16;
                const unsigned long long t1 = wall_clock64();
                atomicMax(g + 8, t1);                                    // end of the merge's last commit
                atomicMax(g + 9, t1 - t0);                               // the longest commit
                atomicMax(g + 11, ((unsigned long long)c.nseg << 32) | rw);   // ... and the longest chain, in segments, with its re-walks
                atomicMax(g + 12, t0);                                   // the start of the last commit to start
                atomicAdd(g + 13, t1 - t0); atomicAdd(g + 14, 1ull); atomicAdd(g + 15, (unsigned long long)rw);
            }
        }
    }
}

__global__ void upgma_init_S_kernel(UpgmaState st)
{
    const uint64_t nn = (uint64_t)st.n * st.n;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nn; e += (uint64_t)gridDim.x * blockDim.x)
    {
        const float v = st.d[e];
        st.S[e] = 0.0f + v;                                               // distance = 0.0 + d[x][y] (:154,162)
    }
}

// dT[y][x] = d[x][y], 32 x 32 tiles through LDS
__global__ __launch_bounds__(256) void upgma_transpose_kernel(const float *__restrict__ d, float *__restrict__ dT, uint32_t n)
{
    __shared__ float tile[32][33];
    const uint32_t tiles = (n + 31) / 32;
    for (uint64_t t = blockIdx.x; t < (uint64_t)tiles * tiles; t += gridDim.x) {
        const uint32_t ty = (uint32_t)(t / tiles), tx = (uint32_t)(t % tiles), lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
        for (uint32_t r = ly; r < 32; r += 8) {
            const uint32_t y = ty * 32 + r, x = tx * 32 + lx;
            tile[r][lx] = (y < n && x < n) ? d[(uint64_t)y * n + x] : 0.0f;
        }
        __syncthreads();
        for (uint32_t r = ly; r < 32; r += 8) {
            const uint32_t y = tx * 32 + r, x = ty * 32 + lx;
            if (y < n && x < n) dT[(uint64_t)y * n + x] = tile[lx][r];
        }
        __syncthreads();
    }
}

// ---- defragmentation of the working copies ------------------------------------------------------------------------------
// The chains of a merge gather |Ck| x n elements of d, and in the caller's layout the members of a cluster are scattered over the
// rows and columns: one useful float per cache line fetched, and at n = 16384 the two chain launches of a merge are bound by that
// (8.5 M gathered lines in ~110 us).  Every few hundred merges the host has the working copies re-laid out so that the members of
// every live cluster are NEIGHBOURS, in member-list order: rows and columns are permuted alike, the member lists keep their
// instance numbers (the order of summation is the reference's, by instance number) and get their physical twins rewritten.  Which
// elements are added, and in which order, does not change -- only where they lie.  A cluster merged after the last
// defragmentation reads two interleaved contiguous runs instead of one.
//   step 1 (one workgroup): first new index of every live cluster, in live-list order (a prefix sum of the member counts)
__global__ __launch_bounds__(1024) void upgma_defrag_offsets_kernel(UpgmaState st)
{
    __shared__ uint32_t wsum[16], carry;
    const uint32_t nl = *st.n_live, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0u;
    __syncthreads();
    for (uint32_t i0 = 0; i0 < nl; i0 += 1024) {
        const uint32_t i = i0 + threadIdx.x;
        const uint32_t c = i < nl ? st.mcount[st.live[i]] : 0u;
        uint32_t incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const uint32_t t = (uint32_t)__shfl_up((int)incl, o); if ((int)lane >= o) incl += t; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t before = carry;
        for (uint32_t w = 0; w < wave; ++w) before += wsum[w];
        if (i < nl) st.doff[i] = before + incl - c;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
        __syncthreads();
    }
}
//   step 2 (one wavefront per live cluster): where every new index comes from, the rewritten physical lists, instance -> index
__global__ __launch_bounds__(256) void upgma_defrag_lists_kernel(UpgmaState st)
{
    const uint32_t nl = *st.n_live, lane = threadIdx.x & 63;
    for (uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; i < nl; i += (gridDim.x * blockDim.x) >> 6) {
        const uint32_t s = st.live[i], m0 = st.mstart[s], cnt = st.mcount[s], off = st.doff[i];
        for (uint32_t j = lane; j < cnt; j += 64) {
            st.dsrc[off + j] = st.ppool[m0 + j];
            st.ppool[m0 + j] = off + j;
            st.phys[st.pool[m0 + j]] = off + j;
        }
    }
}
//   step 3, per matrix: out[i][j] = in[src[i]][src[j]].  One workgroup per row: the source row is staged in LDS (coalesced in, gathered
//   from LDS, coalesced out) while it fits; read from memory directly beyond that (n > 32768).
__global__ __launch_bounds__(1024) void upgma_permute_kernel(const float *__restrict__ in, float *__restrict__ out, const uint32_t *__restrict__ src,
                                                             uint32_t n, int staged)
{
    extern __shared__ float rowbuf[];
    for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {
        const float *row = in + (uint64_t)src[i] * n;
        float *o = out + (uint64_t)i * n;
        if (staged) {
            __syncthreads();                                              // the previous row's readers are done
            for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) rowbuf[j] = row[j];
            __syncthreads();
            for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) o[j] = rowbuf[src[j]];
        } else
            for (uint32_t j = threadIdx.x; j < n; j += blockDim.x) o[j] = row[src[j]];
    }
}

__global__ void upgma_init_kernel(UpgmaState st)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < st.n) {                                                       // parents = [0..n) (:88-91); every row minimum is still to be found
        st.size[i] = 1.0f; st.id[i] = i; st.live[i] = i; st.pos[i] = i; st.pool[i] = i; st.ppool[i] = i; st.phys[i] = i; st.mstart[i] = i; st.mcount[i] = 1;
        st.rscan[i] = 1; st.stale[i] = i; st.rb_l[i] = __builtin_inff(); st.bpos[i] = 0xFFFFFFFFu;
    }
    if (i == 0) {
        *st.n_live = st.n;
        *st.n_big = 0;
        *st.n_stale = st.n; *st.arrive = 0; *st.r_pending = 0; *st.n_items = 0; *st.pack_used = 0; *st.last_sq = 0;
        *st.n_ops = 0; *st.work = 0; *st.items_total = 0; *st.whole_walks = 0;
        *st.pool_used = st.n;
        *st.last_sp = 0xFFFFFFFFu;
        st.rec[0] = 0xFFFFFFFFu;
        *st.done = (st.n > 1 && 0.0f < st.threshold) ? 0u : 1u;           // distance starts at 0.0 (:103)
    }
}

}  // namespace

extern "C" int apd_percentile(apd_context *ctx, const float *x, uint64_t len, float perc, int x_on_device, float *value)
{
    if (!ctx || !value || (len && !x)) return APD_ERR_INVALID_ARG;
    HIP_TRY(ctx, apd::bind_device(ctx));
    const uint64_t k = percentile_index(len, perc);
    if (len == 0 || k >= len) return APD_ERR_INDEX;                       // numbers[n] out of range panics (numerics.rs:132)
    const float *d_x = x;
    float *d_tmp = nullptr;
    if (!x_on_device) {
        HIP_TRY(ctx, hipMalloc((void **)&d_tmp, len * sizeof(float)));
        hipError_t e = hipMemcpyAsync(d_tmp, x, len * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) { hipFree(d_tmp); ctx->last_error = hipGetErrorString(e); return APD_ERR_HIP; }
        d_x = d_tmp;
    }
    const int rc = device_select(ctx, d_x, len, k, value);
    if (d_tmp) hipFree(d_tmp);
    return rc;
}

extern "C" int apd_clustering(apd_context *ctx, const float *distances, int distances_on_device, uint32_t n, float perc,
                              apd_cluster_op *ops, uint32_t *n_ops, uint32_t *roots, uint32_t *n_roots, float *threshold)
{
    if (!ctx || !n_ops || !n_roots || (n && (!distances || !ops || !roots))) return APD_ERR_INVALID_ARG;
    HIP_TRY(ctx, apd::bind_device(ctx));
    *n_ops = 0; *n_roots = 0;
    const uint64_t nn = (uint64_t)n * n;
    const uint64_t k = percentile_index(nn, perc);
    if (nn == 0 || k >= nn) return APD_ERR_INDEX;                         // percentile of an empty / too short vector panics

    UpgmaState st{};
    st.n = n;
    char *pool = nullptr;
    const size_t bytes_S = nn * sizeof(float), bytes_f = (size_t)n * sizeof(float), bytes_u = (size_t)n * sizeof(uint32_t);
    const size_t bytes_lists = ((size_t)n * (n + 1) / 2 + n) * sizeof(uint32_t);    // every merged list is appended once (twice: instance numbers, physical indices)
    const size_t bytes_d = distances_on_device ? 0 : bytes_S;
    // segments of one merge: sum over chains of ceil(rows / rows-per-segment) <= 2 (sum of chain lengths) / kSegElems + chains,
    // and the chains of one merge hold 2 |Ck| (n - |Ck|) <= n^2 / 2 elements
    const size_t max_items = (size_t)(nn / kSegElems) + 2 * (size_t)n + 64;
    const size_t bytes_items = (2 * (size_t)n + 2) * sizeof(uint32_t), bytes_seg = max_items * sizeof(SegRes), bytes_ichain = std::max<size_t>(max_items, 4 * 8192 + 256) * sizeof(uint32_t)   /* (every wavefront of the segment grid reads its entry) */;
    // packed copies of the segments the commit pass is likely to re-walk (a few per chain); when it is full, they are gathered again
    const size_t bytes_packed = (size_t)std::min<uint64_t>(nn / 8 + 65536, 1ull << 30) * sizeof(float);
    // one allocation, every array on a 256-byte boundary (the row scans read S rows, sizes and ids with 16-byte loads)
    size_t off = 0;
    auto carve = [&](size_t bytes) { off = (off + 255) & ~(size_t)255; const size_t at = off; off += bytes; return at; };
    const size_t o_S = carve(bytes_S), o_R = carve(bytes_S), o_d = carve(bytes_d), o_lists = carve(bytes_lists), o_plists = carve(bytes_lists), o_size = carve(bytes_f),
                 o_id = carve(bytes_u), o_live = carve(bytes_u), o_mstart = carve(bytes_u), o_mcount = carve(bytes_u), o_rscan = carve(bytes_u),
                 o_pos = carve(bytes_u), o_big = carve(bytes_u), o_bpos = carve(bytes_u), o_rbl = carve(bytes_f), o_stale = carve(bytes_u),
                 o_rbest = carve((size_t)n * sizeof(Cand)), o_ops = carve((size_t)n * sizeof(apd_cluster_op)), o_seg = carve(bytes_seg),
                 o_istart = carve(bytes_items), o_sdone = carve(bytes_items), o_ichain = carve(bytes_ichain), o_packed = carve(bytes_packed),
                 o_T = carve(bytes_S), o_phys = carve(bytes_u), o_dsrc = carve(bytes_u), o_doff = carve(bytes_u), o_words = carve(256);
    HIP_TRY(ctx, hipMalloc((void **)&pool, off));
    st.S = (float *)(pool + o_S);
    st.R = (float *)(pool + o_R);
    float *d_copy = (float *)(pool + o_d);
    st.pool = (uint32_t *)(pool + o_lists);
    st.ppool = (uint32_t *)(pool + o_plists);
    st.phys = (uint32_t *)(pool + o_phys); st.dsrc = (uint32_t *)(pool + o_dsrc); st.doff = (uint32_t *)(pool + o_doff);
    st.size = (float *)(pool + o_size);
    st.id = (uint32_t *)(pool + o_id);
    st.live = (uint32_t *)(pool + o_live);
    st.mstart = (uint32_t *)(pool + o_mstart);
    st.mcount = (uint32_t *)(pool + o_mcount);
    st.rscan = (uint32_t *)(pool + o_rscan);
    st.pos = (uint32_t *)(pool + o_pos);
    st.big = (uint32_t *)(pool + o_big);
    st.bpos = (uint32_t *)(pool + o_bpos);
    st.rb_l = (float *)(pool + o_rbl);
    st.stale = (uint32_t *)(pool + o_stale);
    st.rbest = (Cand *)(pool + o_rbest);
    st.ops = (apd_cluster_op *)(pool + o_ops);
    st.seg = (SegRes *)(pool + o_seg);
    st.item_start = (uint32_t *)(pool + o_istart);
    st.seg_done = (uint32_t *)(pool + o_sdone);
    st.item_chain = (uint32_t *)(pool + o_ichain);
    st.packed = (float *)(pool + o_packed);
    st.pack_capacity = (uint32_t)(bytes_packed / sizeof(float));
    float *d_T = (float *)(pool + o_T);
    const float **d_mat = (const float **)(pool + o_words + 128);          // three device words behind the counters
    st.mat = d_mat;
    st.n_live = (uint32_t *)(pool + o_words); st.n_ops = st.n_live + 1; st.done = st.n_live + 2; st.work = st.n_live + 3;   // host_state reads these four
    st.pool_used = st.n_live + 12; st.last_sp = st.n_live + 4; st.last_sq = st.n_live + 5; st.n_items = st.n_live + 6;
    st.pack_used = st.n_live + 7; st.n_stale = st.n_live + 8; st.arrive = st.n_live + 9; st.r_pending = st.n_live + 10;
    st.n_big = st.n_live + 11; st.rec = st.n_live + 24;   // (words 24..31: 32-byte aligned)
    st.items_total = st.n_live + 13; st.whole_walks = st.n_live + 14;
    float *spare = nullptr;                                               // two more n x n buffers: the defragmented copies rotate through d_T and these
    auto fail = [&](int rc) { hipFree(pool); if (spare) hipFree(spare); if (st.dbg) hipFree(st.dbg); return rc; };
    if (distances_on_device) st.d = distances;
    else {
        hipError_t e0 = hipMemcpyAsync(d_copy, distances, bytes_S, hipMemcpyHostToDevice, ctx->stream);
        if (e0 != hipSuccess) { ctx->last_error = hipGetErrorString(e0); return fail(APD_ERR_HIP); }
        st.d = d_copy;
    }
    hipError_t e = hipSuccess;

    float thr = 0.0f;
    int rc = device_select(ctx, st.d, nn, k, &thr);                       // clustering.rs:101
    if (rc != APD_OK) return fail(rc);
    st.threshold = thr;
    if (threshold) *threshold = thr;

    hipLaunchKernelGGL(upgma_init_S_kernel, dim3((unsigned)std::min<uint64_t>((nn + 255) / 256, 8192)), dim3(256), 0, ctx->stream, st);
    hipLaunchKernelGGL(upgma_init_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, st);
    hipLaunchKernelGGL(upgma_transpose_kernel, dim3((unsigned)std::min<uint64_t>((uint64_t)((n + 31) / 32) * ((n + 31) / 32), 16384)), dim3(256), 0,
                       ctx->stream, st.d, d_T, n);
    // R[slot y][x] = d[x][y] while every cluster is a singleton (slot y holds instance y): the transpose
    e = hipMemcpyAsync(st.R, d_T, bytes_S, hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return fail(APD_ERR_HIP); }
    // the working copies start out as the caller's matrix and its transpose
    const float *h_mat[3] = {st.d, d_T, nullptr};
    e = hipMemcpyAsync((void *)d_mat, h_mat, sizeof(h_mat), hipMemcpyHostToDevice, ctx->stream);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return fail(APD_ERR_HIP); }
    // Defragmentation (see upgma_permute_kernel) pays when the merges since the last one gathered many elements: the host asks for one
    // between two batches once the new clusters of `defrag_period` or more merges sum to 8 n members (a permutation moves 4 n^2 floats
    // at streaming speed; those merges gathered >= 16 n^2 of them one cache line apiece).  Measured at n = 16384 (round 4): a matrix whose
    // clusters grow to hundreds of members, 5.34 -> 4.00 s with anything from 128 to 512 merges between two; 64 blobs (18 members per
    // new cluster on average), 0.86 s without against 0.88 - 0.93 s on a fixed schedule -- hence the work criterion.
    // APD_UPGMA_DEFRAG = least number of merges between two (0: never).  Without the two spare buffers the loop simply runs without.
    uint32_t defrag_period = n >= 2048 ? 128u : 0u;
    if (const char *v = std::getenv("APD_UPGMA_DEFRAG")) defrag_period = (uint32_t)std::max(0, std::atoi(v));
    const bool defrag_forced = std::getenv("APD_UPGMA_DEFRAG_ALWAYS") != nullptr;   // tests: every `defrag_period` merges, whatever the work
    if (defrag_period && hipMalloc((void **)&spare, 2 * bytes_S) != hipSuccess) { (void)hipGetLastError(); spare = nullptr; defrag_period = 0; }
    std::vector<float *> free_bufs;                                       // buffers not holding a working copy right now
    if (spare) { free_bufs.push_back(spare + nn); free_bufs.push_back(spare); }
    const bool permute_staged = (size_t)n * sizeof(float) <= 128 * 1024;
    if (defrag_period && permute_staged &&
        hipFuncSetAttribute((const void *)upgma_permute_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)n * sizeof(float))) != hipSuccess) {
        (void)hipGetLastError();
        hipFree(spare); spare = nullptr; defrag_period = 0;
    }
    uint32_t defrag_at = 0, defrag_work = 0, n_defrag = 0;                // n_ops and work counter at the latest defragmentation
    auto defragment = [&]() {
        hipLaunchKernelGGL(upgma_defrag_offsets_kernel, dim3(1), dim3(1024), 0, ctx->stream, st);
        hipLaunchKernelGGL(upgma_defrag_lists_kernel, dim3(std::min<uint32_t>((n + 3) / 4, 4096u)), dim3(256), 0, ctx->stream, st);
        const size_t lds = permute_staged ? (size_t)n * sizeof(float) : 0;
        for (int m = 0; m < 2; ++m) {                                     // d, then its transpose: the same permutation of rows and columns
            float *out = free_bufs.back();
            free_bufs.pop_back();
            hipLaunchKernelGGL(upgma_permute_kernel, dim3(std::min<uint32_t>(n, 2048u)), dim3(1024), lds, ctx->stream, h_mat[m], out, st.dsrc, n, (int)permute_staged);
            if (h_mat[m] != st.d) free_bufs.push_back(const_cast<float *>(h_mat[m]));   // (the caller's matrix is never written: first round, m = 0)
            h_mat[m] = out;
        }
        h_mat[2] = h_mat[0];                                              // "defragmented": any non-null word
        return hipMemcpyAsync((void *)d_mat, h_mat, sizeof(h_mat), hipMemcpyHostToDevice, ctx->stream);
    };
    uint32_t host_state[16] = {n, 0, 0, 0};                               // n_live, n_ops, done, work; [13] segments made, [14] long chains walked whole
    // The merge loop is launch-bound (three short dependent launches per merge): a batch of merges is captured once into a
    // hipGraph and replayed until the device-side `done` flag rises; kernels launched after that return immediately.
    const uint32_t batch = 64;
    // stale rows per merge: a handful (the first launch loops over all n); every workgroup costs an arrival, 8 .. 64 measured alike
    const uint32_t select_blocks = std::min<uint32_t>(std::max<uint32_t>(n / 128u, 1u), 64u);
    // Grid sizes (round 4, same-box A/B): one resident round of workgroups that take chains / items in turn (768 chain, 512 segment
    // workgroups) instead of one wavefront per possible chain / item changes nothing at n = 4096 (0.116 vs 0.115 s: a launch that finds
    // nothing to do costs its ~4 us boundary whatever its grid) and LOSES at n = 16384 (blobs 0.83 -> 0.93 s, chain proxy 2.79 -> 3.10 s):
    // a wavefront that walks several chains or segments one after the other is what the critical path of a large merge is made of.
    const uint32_t chain_waves = 2 * ((n + 63) / 64) + 2 * n;             // singleton groups, then one wavefront per (other cluster, direction)
    const uint32_t chain_blocks = (chain_waves + 3) / 4;
    // one wavefront per item up to 32768 items (a wavefront that commits a chain must not hold other items back), grid-stride beyond
    uint32_t segment_blocks = (std::min((2 * n + 3) / 4, 8192u) + 63u) / 64u * 64u;   // groups of 32 wavefronts per XCD
    if (const char *v = std::getenv("APD_UPGMA_SEGMENT_BLOCKS")) segment_blocks = (uint32_t)std::max(64, std::atoi(v)) / 64u * 64u;   // tuning
    const bool debug_timing = std::getenv("APD_DEBUG_UPGMA_TIMING") != nullptr;   // tuning aid: phase stamps of every select launch
    if (debug_timing && hipMalloc((void **)&st.dbg, (size_t)n * 16 * sizeof(unsigned long long)) == hipSuccess)
        (void)hipMemsetAsync(st.dbg, 0, (size_t)n * 16 * sizeof(unsigned long long), ctx->stream);
    // Two captured batches: [select, chain, segment] x 64 and [select, chain] x 64.  The second one is replayed while the batch before
    // made no segment at all (most of a dendrogram at n = 4096: the segment launch then only costs its boundary, ~4.5 us per merge);
    // its chain launch walks a long chain whole if one turns up after all, and the host returns to three launches for the next batch.
    hipGraph_t graph[2] = {nullptr, nullptr};
    hipGraphExec_t exec[2] = {nullptr, nullptr};
    auto enqueue_batch = [&](int two) {
        UpgmaState sb = st;
        sb.no_seg = two ? 1u : 0u;
        for (uint32_t b = 0; b < batch; ++b) {
            hipLaunchKernelGGL(upgma_select_kernel, dim3(select_blocks), dim3(1024), 0, ctx->stream, sb);
            hipLaunchKernelGGL(upgma_chain_kernel, dim3(chain_blocks), dim3(256), 0, ctx->stream, sb);
            if (!two) hipLaunchKernelGGL(upgma_segment_kernel, dim3(segment_blocks), dim3(256), 0, ctx->stream, sb);
        }
    };
    auto drop_graph = [&]() {
        for (int g = 0; g < 2; ++g) { if (exec[g]) hipGraphExecDestroy(exec[g]); if (graph[g]) hipGraphDestroy(graph[g]); exec[g] = nullptr; graph[g] = nullptr; }
    };
    e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return fail(APD_ERR_HIP); }
    // the legacy default stream cannot be captured: then the batch is enqueued directly
    st.short_chain = kShortChain;
    if (const char *v = std::getenv("APD_UPGMA_SHORT_CHAIN")) st.short_chain = (uint32_t)std::max(64, std::atoi(v));   // tuning
    bool use_graph = ctx->stream != nullptr && std::getenv("APD_UPGMA_NO_GRAPH") == nullptr &&   // (the env: plain launches, for profilers that choke on graphs)
                     hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (use_graph) {
        enqueue_batch(0);
        use_graph = hipStreamEndCapture(ctx->stream, &graph[0]) == hipSuccess && hipGraphInstantiate(&exec[0], graph[0], nullptr, nullptr, 0) == hipSuccess;
        if (use_graph && hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            enqueue_batch(1);
            use_graph = hipStreamEndCapture(ctx->stream, &graph[1]) == hipSuccess && hipGraphInstantiate(&exec[1], graph[1], nullptr, nullptr, 0) == hipSuccess;
        } else use_graph = false;
        if (!use_graph) drop_graph();
    }
    (void)hipGetLastError();
    // APD_UPGMA_TWO_LAUNCH: 0 = always three launches per merge, 2 = always two (tests: every long chain walked whole), default = by batch
    int two_policy = 1;
    if (const char *v = std::getenv("APD_UPGMA_TWO_LAUNCH")) two_policy = std::atoi(v);
    int two = two_policy == 2 ? 1 : 0;
    uint32_t items_before = 0, whole_before = 0, n_two = 0, n_batches = 0;
    const bool debug = std::getenv("APD_DEBUG_UPGMA") != nullptr;

    uint32_t ops_before = 0;
    while (true) {
        if (use_graph) e = hipGraphLaunch(exec[two], ctx->stream);
        else { enqueue_batch(two); e = hipGetLastError(); }
        ++n_batches; n_two += (uint32_t)two;
        if (e == hipSuccess) e = hipMemcpyAsync(host_state, st.n_live, sizeof(host_state), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); drop_graph(); return fail(APD_ERR_HIP); }
        if (debug) std::fprintf(stderr, "[apd] upgma: graph %d, live %u, ops %u, done %u\n", (int)use_graph, host_state[0], host_state[1], host_state[2]);
        if (host_state[2] != 0) break;
        // every batch performs `batch` merges unless the loop condition failed: a batch without progress would spin forever
        if (host_state[1] < ops_before + batch) {
            ctx->last_error = "UPGMA made no progress in a batch of merges (n_ops " + std::to_string(host_state[1]) + " after " + std::to_string(ops_before) + ")";
            drop_graph();
            return fail(APD_ERR_HIP);
        }
        ops_before = host_state[1];
        if (two_policy == 1) {
            if (!two) two = host_state[13] == items_before ? 1 : 0;      // a whole batch without a segment: leave the segment launches out
            else if (host_state[14] != whole_before) two = 0;            // a long chain turned up: three launches again
        }
        items_before = host_state[13]; whole_before = host_state[14];
        if (defrag_period && host_state[0] > 2 && host_state[1] - defrag_at >= defrag_period &&
            (defrag_forced || (uint64_t)(host_state[3] - defrag_work) >= 8ull * n)) {
            e = defragment();
            if (e == hipSuccess) e = hipGetLastError();
            if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); drop_graph(); return fail(APD_ERR_HIP); }
            defrag_at = host_state[1]; defrag_work = host_state[3]; ++n_defrag;
        }
    }
    drop_graph();
    if (debug || debug_timing) std::fprintf(stderr, "[apd] upgma: %u defragmentations (period %u merges); %u of %u batches without segment launches, %u long chains walked whole\n",
                                            n_defrag, defrag_period, n_two, n_batches, host_state[14]);
    const uint32_t cnt = host_state[1];
    if (st.dbg) {
        std::vector<unsigned long long> g((size_t)n * 16);
        if (hipMemcpy(g.data(), st.dbg, g.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) == hipSuccess && cnt > 1) {
            double ph[4] = {0, 0, 0, 0}, stale = 0, merged = 0, chain = 0, seg = 0, gap = 0, maps = 0, items = 0, longest = 0, rewalk = 0, cstart = 0, csum = 0, ccnt = 0, rwsum = 0;
            double stamped = 0;
            for (uint32_t t = 1; t + 1 < cnt; ++t) {                      // stamps: 100 MHz; merge 0 scans every row, the last merge ends the loop
                const unsigned long long *q = &g[(size_t)t * 16];
                stamped += 1;
                for (int k = 0; k < 4; ++k) ph[k] += (double)(q[k + 1] - q[k]) * 0.01;
                stale += (double)q[5]; merged += (double)q[6];
                chain += (double)(q[7] - q[4]) * 0.01;                    // end of select's bookkeeping -> last wavefront of the chain launch
                seg += (double)(std::max(q[8], q[7]) - q[7]) * 0.01;      // -> last wavefront of the segment launch that had work
                gap += (double)(g[(size_t)(t + 1) * 16] - std::max(q[8], q[7])) * 0.01;
                maps += (double)q[9] * 0.01; items += (double)q[10]; longest += (double)(q[11] >> 32); rewalk += (double)(q[11] & 0xFFFFFFFFull);
                if (q[12] > q[7]) cstart += (double)(q[12] - q[7]) * 0.01;
                csum += (double)q[13] * 0.01; ccnt += (double)q[14]; rwsum += (double)q[15];
            }
            const double m = std::max(1.0, stamped);
            std::fprintf(stderr, "[apd] upgma us per merge: select [rows %.2f | arrive %.2f | argmin %.2f | lists %.2f] chain launch %.2f, segment launch %.2f, "
                                 "(longest commit %.2f; %.1f segments per merge, longest chain %.1f, %.1f of its sub-blocks walked) to the next select's entry %.2f ; stale rows %.1f, merged list %.1f members\n",
                         ph[0] / m, ph[1] / m, ph[2] / m, ph[3] / m, chain / m, seg / m, maps / m, items / m, longest / m, rewalk / m, gap / m, stale / m, merged / m);
            std::fprintf(stderr, "[apd] upgma commits: %.1f per merge, mean %.2f us each, %.2f sub-block walks each; the last one starts %.2f us after the chain launch's end\n",
                         ccnt / m, csum / std::max(1.0, ccnt), rwsum / std::max(1.0, ccnt), cstart / m);
        }
        hipFree(st.dbg);
    }
    std::vector<uint32_t> ids(n), live(host_state[0]);
    e = hipMemcpyAsync(ops, st.ops, (size_t)cnt * sizeof(apd_cluster_op), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ids.data(), st.id, bytes_u, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && !live.empty()) e = hipMemcpyAsync(live.data(), st.live, live.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return fail(APD_ERR_HIP); }
    // dendrogram.clusters() (clustering.rs:109,146-148): the root ids, ascending
    std::vector<uint32_t> r;
    for (uint32_t s : live) r.push_back(ids[s]);
    if (host_state[2] == 2 && cnt > 0) {
        // the degenerate (0, 0) merge re-parents INSTANCE 0 onto the new node (clustering.rs:136-138 with p = q = 0):
        // its old cluster keeps its other members, instance 0 alone now roots at `into`
        const uint32_t k_new = ops[cnt - 1].into;
        bool zero_was_singleton_root = false;
        for (uint32_t &v : r) if (v == 0) { v = k_new; zero_was_singleton_root = true; }
        if (!zero_was_singleton_root) r.push_back(k_new);
    }
    std::sort(r.begin(), r.end());
    for (size_t i = 0; i < r.size(); ++i) roots[i] = r[i];
    *n_roots = (uint32_t)r.size();
    *n_ops = cnt;
    hipFree(pool);
    if (spare) hipFree(spare);
    return APD_OK;
}

// clustering.rs:40-76: replay the merge list into leaf lists.  Pure bookkeeping on <= 2n ids, host side.
extern "C" int apd_cluster_sets(const apd_cluster_op *ops, uint32_t n_ops, const uint32_t *roots, uint32_t n_roots,
                                uint32_t n, uint32_t *members, uint32_t *set_off, uint32_t *n_sets)
{
    if ((n_ops && !ops) || (n_roots && !roots) || !members || !set_off || !n_sets) return APD_ERR_INVALID_ARG;
    const uint64_t ids = (uint64_t)n + n_ops + 2;
    std::vector<std::vector<uint32_t>> results(ids);
    std::vector<char> present(ids, 0);
    for (uint32_t t = 0; t < n_ops; ++t) {
        const uint32_t i = ops[t].merge_i, j = ops[t].merge_j, k = ops[t].into;
        if (k >= ids) return APD_ERR_INVALID_ARG;
        std::vector<uint32_t> cluster;
        if (i < ids && present[i]) cluster.insert(cluster.end(), results[i].begin(), results[i].end());   // :48-49
        else cluster.push_back(i);                                                                         // :51
        if (j < ids && present[j]) cluster.insert(cluster.end(), results[j].begin(), results[j].end());   // :53-54
        else cluster.push_back(j);                                                                         // :56
        results[k] = std::move(cluster);                                                                   // :58
        present[k] = 1;
    }
    uint32_t ns = 0, pos = 0;
    set_off[0] = 0;
    for (uint32_t r = 0; r < n_roots; ++r) {                                                               // :61
        const uint32_t id = roots[r];
        if (id < ids && present[id]) {
            for (uint32_t v : results[id])
                if (v < n) { if (pos >= n + n_ops + 2) return APD_ERR_INVALID_ARG; members[pos++] = v; }   // :65-69
            set_off[++ns] = pos;
        }                                                                                                  // else: "Cluster not found", :71
    }
    *n_sets = ns;
    return APD_OK;
}

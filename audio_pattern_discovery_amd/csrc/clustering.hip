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
//   by rounding noise -- is bit-identical to the literal algorithm.  Each merge is three launches: row minima (a
//   cached best ordered pair per live row; only rows whose cache went stale re-read their row of S), one workgroup
//   that reduces the row minima, applies merge_clusters and merges the two member lists, and one pass that rebuilds
//   the new cluster's row and column and patches the other rows' caches.  Exact ties resolve to the lowest (id_p, id_q): what the reference does when its HashSet happens
//   to iterate in ascending order (clustering.rs:180-187); any other order is equally "reference".
#include <algorithm>
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

struct Cand { float l; uint32_t idp, idq, sp, sq; };

__device__ __forceinline__ bool better(const Cand &a, const Cand &b)
{
    // strict '<' in ascending (id_p, id_q) iteration order (clustering.rs:184): first minimum wins; NaN never wins
    if (!(a.l < __builtin_inff())) return false;      // min_linkage starts at +INF: an infinite linkage is never taken
    if (a.l < b.l) return true;
    if (a.l == b.l) return (a.idp < b.idp) || (a.idp == b.idp && a.idq < b.idq);
    return false;
}

struct UpgmaState {
    const float *d;           // [n][n] raw distances (read only)
    float *S;                 // [n][n] directed cluster sums (slot-indexed)
    uint32_t *pool;           // sorted member lists, appended per merge
    uint32_t *mstart, *mcount;// [n] list of the cluster held by a slot
    uint32_t *pool_used;
    uint32_t *last_sp;        // slot of the cluster created by the latest merge (0xFFFFFFFF: none)
    float *size;              // [n] member counts as f32 (the reference counts in f32, clustering.rs:154-166)
    uint32_t *id;             // [n] cluster id held by the slot
    uint32_t *live;           // [n_live] slots still holding a root, ascending slot order is irrelevant
    uint32_t *n_live;
    Cand *rbest;              // [n] cached best ordered pair of the row held by a slot
    uint32_t *rscan;          // [n] 1: the cache of this row is stale, rescan it
    uint32_t *last_sq;        // slot that died in the latest merge
    apd_cluster_op *ops;      // [n]
    uint32_t *n_ops;
    uint32_t *done;           // set once the loop condition of clustering.rs:104 fails
    float threshold;
    uint32_t n;
};

// Row minima: one workgroup per live row whose cached best pair is stale (new cluster, or its best column just
// merged/died); the others keep their cache.  A full scan reads one row of S: the per-merge HBM traffic is
// (stale rows) x 4c bytes instead of 4c^2.
__global__ __launch_bounds__(256) void upgma_rowmin_kernel(UpgmaState st)
{
    __shared__ Cand red[256];
    if (*st.done != 0) return;
    const uint32_t nl = *st.n_live;
    for (uint32_t r = blockIdx.x; r < nl; r += gridDim.x) {
        const uint32_t sp = st.live[r];
        if (st.rscan[sp] == 0) continue;                                  // block-uniform
        Cand best{__builtin_inff(), 0xFFFFFFFFu, 0xFFFFFFFFu, sp, sp};
        const float size_p = st.size[sp];
        const uint32_t idp = st.id[sp];
        const float *row = st.S + (uint64_t)sp * st.n;
        for (uint32_t c = threadIdx.x; c < nl; c += blockDim.x) {
            const uint32_t sq = st.live[c];
            if (sq == sp) continue;                                       // target_i != target_j (clustering.rs:182)
            const float denom = size_p * st.size[sq];                     // size_x * size_y (:169)
            const Cand cnd{row[sq] / denom, idp, st.id[sq], sp, sq};
            if (better(cnd, best)) best = cnd;
        }
        red[threadIdx.x] = best;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s && better(red[threadIdx.x + s], red[threadIdx.x])) red[threadIdx.x] = red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) { st.rbest[sp] = red[0]; st.rscan[sp] = 0; }
        __syncthreads();
    }
}

// One workgroup: final arg-min, merge_clusters (clustering.rs:134-141), sums of the new cluster, the op record.
__global__ __launch_bounds__(1024) void upgma_merge_kernel(UpgmaState st)
{
    __shared__ Cand red[1024];
    __shared__ Cand win;
    if (*st.done != 0) return;
    Cand best{__builtin_inff(), 0xFFFFFFFFu, 0xFFFFFFFFu, 0, 0};
    {
        const uint32_t nl0 = *st.n_live;
        for (uint32_t c = threadIdx.x; c < nl0; c += blockDim.x) { const Cand cc = st.rbest[st.live[c]]; if (better(cc, best)) best = cc; }
    }
    red[threadIdx.x] = best;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s && better(red[threadIdx.x + s], red[threadIdx.x])) red[threadIdx.x] = red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) win = red[0];
    __syncthreads();
    const Cand w = win;
    const uint32_t t = *st.n_ops;
    const uint32_t k = st.n + t;                                          // parents.len() (:135)
    const uint32_t nl = *st.n_live;
    if (w.idp == 0xFFFFFFFFu) {
        // no linkage below +INF: the reference keeps min_merge = (0, 0) (clustering.rs:179) and merges it
        if (threadIdx.x == 0) {
            st.ops[t] = apd_cluster_op{0u, 0u, k, __builtin_inff(), (uint32_t)APD_SEQUENCE2SEQUENCE};
            *st.n_ops = t + 1;
            *st.last_sp = 0xFFFFFFFFu;
            *st.done = 2;                                                 // INF < threshold is false: loop ends (:104)
        }
        return;
    }
    // new cluster k lives in slot sp; slot sq dies.  Merge the two sorted member lists into a fresh one.
    {
        const uint32_t *lp = st.pool + st.mstart[w.sp], *lq = st.pool + st.mstart[w.sq];
        const uint32_t cp = st.mcount[w.sp], cq = st.mcount[w.sq];
        uint32_t *out = st.pool + *st.pool_used;
        for (uint32_t i = threadIdx.x; i < cp + cq; i += blockDim.x) {
            const bool from_p = i < cp;
            const uint32_t v = from_p ? lp[i] : lq[i - cp];
            const uint32_t *other = from_p ? lq : lp;
            uint32_t lo = 0, hi = from_p ? cq : cp;                       // members are distinct: plain lower bound
            while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (other[mid] < v) lo = mid + 1; else hi = mid; }
            out[(from_p ? i : i - cp) + lo] = v;
        }
    }
    // drop sq from the live list (its order is irrelevant): every thread scans a stride, the finder swaps in the tail
    {
        const uint32_t tail = st.live[nl - 1];
        __syncthreads();
        for (uint32_t c = threadIdx.x; c < nl; c += blockDim.x)
            if (st.live[c] == w.sq) st.live[c] = tail;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t op;
        const uint32_t n = st.n;
        if (w.idp < n && w.idq < n) op = APD_SEQUENCE2SEQUENCE;           // clustering.rs:193-201
        else if (w.idp >= n && w.idq >= n) op = APD_CLUSTER2CLUSTER;
        else if (w.idp >= n && w.idq < n) op = APD_CLUSTER2SEQUENCE;
        else op = APD_SEQUENCE2CLUSTER;
        st.ops[t] = apd_cluster_op{w.idp, w.idq, k, w.l, op};
        *st.n_ops = t + 1;
        st.size[w.sp] = st.size[w.sp] + st.size[w.sq];
        st.id[w.sp] = k;
        const uint32_t used = *st.pool_used, cnt = st.mcount[w.sp] + st.mcount[w.sq];
        st.mstart[w.sp] = used; st.mcount[w.sp] = cnt; *st.pool_used = used + cnt;
        *st.last_sp = w.sp; *st.last_sq = w.sq;
        st.rscan[w.sp] = 1;                                               // the new cluster's row is new
        *st.n_live = nl - 1;
        if (nl - 1 <= 1 || !(w.l < st.threshold)) *st.done = 1;           // while n_clusters > 1 && distance < threshold (:104)
    }
}

// Row and column of the cluster created by the latest merge, summed exactly as linkage() does
// (clustering.rs:157-169): x ascending, y ascending, ONE f32 accumulator -- the order is the contract, so the sum
// cannot be a tree.  One wavefront per (other cluster, direction): the 64 lanes gather 64 distances at a time, then
// the accumulator walks them in order (v_readlane + add), ~100x faster than one thread chasing dependent loads.
__global__ __launch_bounds__(256) void upgma_update_kernel(UpgmaState st)
{
    const uint32_t sp = *st.last_sp;
    if (sp == 0xFFFFFFFFu || *st.done != 0) return;                      // nothing merged, or no further arg-min will run
    const uint32_t nl = *st.n_live;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const uint32_t c = wave >> 1, dir = wave & 1u;
    if (c >= nl) return;
    const uint32_t s = st.live[c];
    if (s == sp) return;
    const uint32_t *lx = st.pool + st.mstart[dir ? s : sp], *ly = st.pool + st.mstart[dir ? sp : s];
    const uint32_t cx = st.mcount[dir ? s : sp], cy = st.mcount[dir ? sp : s];
    float acc = 0.0f;
    const uint64_t total = (uint64_t)cx * cy;
    if (total < (1ull << 32)) {
        // the (x, y) pairs in linkage()'s order are one linear sequence e = a * cy + b: 64 at a time are gathered (lane = e & 63),
        // kPre chunks ahead of the adds so that the gather latency hides behind them; a chunk is padded with +0.0, which the
        // accumulator absorbs exactly, so every chunk is the same 64 unrolled readlane + add steps.
        constexpr int kPre = 8;
        const uint32_t tot = (uint32_t)total;
        auto fetch = [&](uint32_t e0) __attribute__((always_inline)) -> float {
            const uint32_t e = e0 + lane;
            if (e0 >= tot || e >= tot) return 0.0f;
            const uint32_t a = e / cy, b = e - a * cy;
            return st.d[(uint64_t)lx[a] * st.n + ly[b]];
        };
        float buf[kPre];
#pragma unroll
        for (int k = 0; k < kPre; ++k) buf[k] = fetch((uint32_t)k * 64u);
        for (uint64_t e0 = 0; e0 < total; e0 += 64ull * kPre) {
#pragma unroll
            for (int k = 0; k < kPre; ++k) {
                const uint64_t ek = e0 + 64ull * k;
                if (ek < total) {                                        // wave-uniform
                    const float v = buf[k];
                    const uint64_t nx = ek + 64ull * kPre;
                    buf[k] = (nx < total) ? fetch((uint32_t)nx) : 0.0f;
#pragma unroll
                    for (int t = 0; t < 64; ++t)
                        acc = acc + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), t));
                }
            }
        }
    } else {
        for (uint32_t a = 0; a < cx; ++a) {
            const float *row = st.d + (uint64_t)lx[a] * st.n;
            for (uint32_t b0 = 0; b0 < cy; b0 += 64) {
                const uint32_t cnt = min(64u, cy - b0);
                const float v = (lane < cnt) ? row[ly[b0 + lane]] : 0.0f;
                for (uint32_t t = 0; t < cnt; ++t)
                    acc = acc + __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), t));
            }
        }
    }
    if (lane != 0) return;
    if (dir) {
        st.S[(uint64_t)s * st.n + sp] = acc;
        // row s: its cached best pair survives unless it pointed at one of the two merged slots; the new entry may beat it
        const Cand old = st.rbest[s];
        if (old.sq == sp || old.sq == *st.last_sq) st.rscan[s] = 1;
        else {
            const Cand cnd{acc / (st.size[s] * st.size[sp]), st.id[s], st.id[sp], s, sp};
            if (better(cnd, old)) st.rbest[s] = cnd;
        }
    } else st.S[(uint64_t)sp * st.n + s] = acc;
}

__global__ void upgma_init_S_kernel(UpgmaState st)
{
    const uint64_t nn = (uint64_t)st.n * st.n;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nn; e += (uint64_t)gridDim.x * blockDim.x)
        st.S[e] = 0.0f + st.d[e];                                         // distance = 0.0 + d[x][y] (:154,162)
}

__global__ void upgma_init_kernel(UpgmaState st)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < st.n) { st.size[i] = 1.0f; st.id[i] = i; st.live[i] = i; st.pool[i] = i; st.mstart[i] = i; st.mcount[i] = 1; st.rscan[i] = 1; }   // parents = [0..n) (:88-91)
    if (i == 0) {
        *st.n_live = st.n;
        *st.n_ops = 0;
        *st.pool_used = st.n;
        *st.last_sp = 0xFFFFFFFFu;
        *st.done = (st.n > 1 && 0.0f < st.threshold) ? 0u : 1u;           // distance starts at 0.0 (:103)
    }
}

}  // namespace

extern "C" int apd_percentile(apd_context *ctx, const float *x, uint64_t len, float perc, int x_on_device, float *value)
{
    if (!ctx || !value || (len && !x)) return APD_ERR_INVALID_ARG;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    *n_ops = 0; *n_roots = 0;
    const uint64_t nn = (uint64_t)n * n;
    const uint64_t k = percentile_index(nn, perc);
    if (nn == 0 || k >= nn) return APD_ERR_INDEX;                         // percentile of an empty / too short vector panics

    UpgmaState st{};
    st.n = n;
    const int n_blocks = (int)std::min<uint32_t>(n, 4096);
    char *pool = nullptr;
    const size_t bytes_S = nn * sizeof(float), bytes_f = (size_t)n * sizeof(float), bytes_u = (size_t)n * sizeof(uint32_t);
    const size_t bytes_lists = ((size_t)n * (n + 1) / 2 + n) * sizeof(uint32_t);    // every merged list is appended once
    const size_t bytes_d = distances_on_device ? 0 : bytes_S;
    const size_t total = bytes_S + bytes_d + bytes_lists + bytes_f + 5 * bytes_u + (size_t)n * sizeof(Cand) +
                         (size_t)n * sizeof(apd_cluster_op) + 256;
    HIP_TRY(ctx, hipMalloc((void **)&pool, total));
    size_t off = 0;
    st.S = (float *)(pool + off); off += bytes_S;
    float *d_copy = (float *)(pool + off); off += bytes_d;
    st.pool = (uint32_t *)(pool + off); off += bytes_lists;
    st.size = (float *)(pool + off); off += bytes_f;
    st.id = (uint32_t *)(pool + off); off += bytes_u;
    st.live = (uint32_t *)(pool + off); off += bytes_u;
    st.mstart = (uint32_t *)(pool + off); off += bytes_u;
    st.mcount = (uint32_t *)(pool + off); off += bytes_u;
    st.rscan = (uint32_t *)(pool + off); off += bytes_u;
    st.rbest = (Cand *)(pool + off); off += (size_t)n * sizeof(Cand);
    st.ops = (apd_cluster_op *)(pool + off); off += (size_t)n * sizeof(apd_cluster_op);
    off = (off + 63) & ~(size_t)63;
    st.n_live = (uint32_t *)(pool + off); st.n_ops = st.n_live + 1; st.done = st.n_live + 2;
    st.pool_used = st.n_live + 3; st.last_sp = st.n_live + 4; st.last_sq = st.n_live + 5;
    auto fail = [&](int rc) { hipFree(pool); return rc; };
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
    uint32_t host_state[3] = {n, 0, 0};                                   // n_live, n_ops, done
    // The merge loop is launch-bound (three short dependent launches per merge): a batch of merges is captured once into a
    // hipGraph and replayed until the device-side `done` flag rises; kernels launched after that return immediately.
    const uint32_t batch = 64;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    auto enqueue_batch = [&]() {
        for (uint32_t b = 0; b < batch; ++b) {
            hipLaunchKernelGGL(upgma_rowmin_kernel, dim3(n_blocks), dim3(256), 0, ctx->stream, st);
            hipLaunchKernelGGL(upgma_merge_kernel, dim3(1), dim3(1024), 0, ctx->stream, st);
            hipLaunchKernelGGL(upgma_update_kernel, dim3((2 * n + 3) / 4), dim3(256), 0, ctx->stream, st);
        }
    };
    auto drop_graph = [&]() { if (exec) hipGraphExecDestroy(exec); if (graph) hipGraphDestroy(graph); exec = nullptr; graph = nullptr; };
    e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return fail(APD_ERR_HIP); }
    // the legacy default stream cannot be captured: then the batch is enqueued directly
    bool use_graph = ctx->stream != nullptr && hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
    if (use_graph) {
        enqueue_batch();
        use_graph = hipStreamEndCapture(ctx->stream, &graph) == hipSuccess && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
        if (!use_graph) drop_graph();
    }
    (void)hipGetLastError();
    while (true) {
        if (use_graph) e = hipGraphLaunch(exec, ctx->stream);
        else { enqueue_batch(); e = hipGetLastError(); }
        if (e == hipSuccess) e = hipMemcpyAsync(host_state, st.n_live, sizeof(host_state), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); drop_graph(); return fail(APD_ERR_HIP); }
        if (host_state[2] != 0) break;
    }
    drop_graph();
    const uint32_t cnt = host_state[1];
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

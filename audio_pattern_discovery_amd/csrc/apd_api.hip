// C ABI of libapd_hip.so (include/apd.h): context, resident batches, tile sharding and the
// host-side plumbing around the alignment kernels.  No torch, no CPU fallback.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>

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

int ensure_ws(apd_context *ctx, void **p, size_t *have, size_t need)
{
    APD_AFFINITY(ctx, "workspace allocation");
    if (*have >= need && *p) return APD_OK;
    if (*p) { HIP_TRY(ctx, hipFree(*p)); *p = nullptr; *have = 0; }
    HIP_TRY(ctx, hipMalloc(p, need));
    *have = need;
    return APD_OK;
}

uint32_t tiles_side(uint32_t n_seq) { return (n_seq + kTile - 1) / kTile; }

void rank_tile_list(uint32_t n_seq, uint32_t rank, uint32_t world, std::vector<uint2> &out)
{
    out.clear();
    const uint32_t side = tiles_side(n_seq);
    uint64_t g = 0;
    for (uint32_t ta = 0; ta < side; ++ta)
        for (uint32_t tb = ta; tb < side; ++tb, ++g)
            if (g % world == rank) out.push_back(make_uint2(ta, tb));
}

// Resident / tiling order: position p holds sequence order[p]; longest first, equal lengths by ascending index.  Pure
// function of the lengths, so every rank derives the same order.
void length_order(const uint64_t *offsets, uint32_t n_seq, std::vector<uint32_t> &order)
{
    order.resize(n_seq);
    for (uint32_t s = 0; s < n_seq; ++s) order[s] = s;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        return offsets[a + 1] - offsets[a] > offsets[b + 1] - offsets[b];
    });
}

// cells visited by alignments.rs:174-175 for lengths (n, m) and half-width w:
// #{(i,j) in [1,n]x[1,m] : -w <= j-i <= w-1}
uint64_t tri_count(uint64_t n, uint64_t m, uint64_t k)   // #{(i,j): j - i >= k}, k >= 0
{
    if (m <= k) return 0;
    const uint64_t q = m - k, l = std::min(n, q);
    return l * q - l * (l - 1) / 2;
}
uint64_t band_cells(uint64_t n, uint64_t m, uint64_t w)
{
    return n * m - tri_count(n, m, w) - tri_count(m, n, w + 1);
}

}  // namespace

namespace apd {
thread_local const apd_context *tl_bound_context = nullptr;
bool affinity_debug()
{
    static const bool on = [] { const char *v = std::getenv("APD_DEBUG_AFFINITY"); return v && v[0] && v[0] != '0'; }();
    return on;
}
hipError_t bind_device(const apd_context *ctx)
{
    tl_bound_context = ctx;
    return hipSetDevice(ctx->device);
}
bool affinity_ok(const apd_context *ctx, const char *where)
{
    int dev = -1;
    const bool ok = tl_bound_context == ctx && hipGetDevice(&dev) == hipSuccess && dev == ctx->device;
    if (!ok) {
        const_cast<apd_context *>(ctx)->last_error = std::string("APD_DEBUG_AFFINITY: ") + where + " on a thread bound to " +
            (tl_bound_context == ctx ? "this context but HIP device " + std::to_string(dev) :
             tl_bound_context ? "another context (device " + std::to_string(tl_bound_context->device) + ")" : "no context") +
            ", expected device " + std::to_string(ctx->device);
        std::fprintf(stderr, "[apd] %s\n", ctx->last_error.c_str());
    }
    return ok;
}
}  // namespace apd

// ------------------------------------------------------------------------------------ context

extern "C" const char *apd_status_string(int s)
{
    switch (s) {
        case APD_OK: return "ok";
        case APD_ERR_INVALID_ARG: return "invalid argument";
        case APD_ERR_NO_DEVICE: return "no gfx950 HIP device";
        case APD_ERR_HIP: return "HIP runtime error";
        case APD_ERR_OOM: return "out of device memory";
        case APD_ERR_EMPTY_SEQUENCE: return "zero-length sequence (undefined in the reference, alignments.rs:120)";
        case APD_ERR_BAND_TOO_WIDE: return "warping band too wide for one wavefront";
        case APD_ERR_INDEX: return "percentile index out of range (the reference panics, numerics.rs:132)";
        case APD_ERR_UNSUPPORTED: return "unsupported";
        case APD_ERR_INCOMPLETE: return "a pair score was never written (launch cut short or skipped): NaN left in the output";
        case APD_ERR_COMM: return "RCCL error";
        default: return "unknown status";
    }
}

extern "C" int apd_create(int device, apd_context **out)
{
    if (!out) return APD_ERR_INVALID_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return APD_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return APD_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return APD_ERR_NO_DEVICE;   // kernels are built for gfx950 only
    apd_context *ctx = new (std::nothrow) apd_context();
    if (!ctx) return APD_ERR_OOM;
    ctx->device = device;
    if (bind_device(ctx) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return APD_ERR_HIP;
    }
    ctx->own_stream = true;
    if (hipMalloc((void **)&ctx->d_status, 64) != hipSuccess || hipMemset(ctx->d_status, 0, 64) != hipSuccess) {
        hipStreamDestroy(ctx->stream);
        delete ctx;
        return APD_ERR_OOM;
    }
    hipEventCreate(&ctx->ev0);
    hipEventCreate(&ctx->ev1);
    hipEventCreateWithFlags(&ctx->fork, hipEventDisableTiming);
    for (int k = 0; k < apd_context::kSideStreams; ++k) {
        hipStreamCreateWithFlags(&ctx->side[k], hipStreamNonBlocking);
        hipEventCreateWithFlags(&ctx->side_done[k], hipEventDisableTiming);
    }
    *out = ctx;
    return APD_OK;
}

static void release_batch_device_memory(apd_batch *b);

extern "C" int apd_destroy(apd_context *ctx)
{
    if (!ctx) return APD_ERR_INVALID_ARG;
    bind_device(ctx);
    hipStreamSynchronize(ctx->stream);
    if (ctx->pair_batch) { apd_batch_destroy(ctx->pair_batch); ctx->pair_batch = nullptr; }   // apd_align_pair's own (nobody else holds it)
    for (apd_batch *b : ctx->batches) { release_batch_device_memory(b); b->ctx = nullptr; }   // orphans: see apd_batch_destroy
    ctx->batches.clear();
    for (apd_comm *c : ctx->comms) apd::orphan_comm(c);
    ctx->comms.clear();
    for (apd_encoder *e : ctx->encoders) apd::orphan_encoder(e);
    ctx->encoders.clear();
    for (apd_cepstrum_plan *cp : ctx->cepstrum_plans) apd::orphan_cepstrum_plan(cp);
    ctx->cepstrum_plans.clear();
    for (void *p : ctx->buffers) hipFree(p);                              // apd_device_alloc'ed and never freed
    ctx->buffers.clear();
    if (ctx->ws_tiles) hipFree(ctx->ws_tiles);
    if (ctx->ws_slab) hipFree(ctx->ws_slab);
    if (ctx->ws_misc) hipFree(ctx->ws_misc);
    if (ctx->ws_gather) hipFree(ctx->ws_gather);
    if (ctx->d_status) hipFree(ctx->d_status);
    if (ctx->ev0) hipEventDestroy(ctx->ev0);
    if (ctx->ev1) hipEventDestroy(ctx->ev1);
    if (ctx->fork) hipEventDestroy(ctx->fork);
    for (int k = 0; k < apd_context::kSideStreams; ++k) {
        if (ctx->side[k]) { hipStreamSynchronize(ctx->side[k]); hipStreamDestroy(ctx->side[k]); }
        if (ctx->side_done[k]) hipEventDestroy(ctx->side_done[k]);
    }
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return APD_OK;
}

extern "C" int apd_set_stream(apd_context *ctx, void *hip_stream)
{
    if (!ctx) return APD_ERR_INVALID_ARG;
    if (ctx->own_stream && ctx->stream) { hipStreamSynchronize(ctx->stream); hipStreamDestroy(ctx->stream); }
    ctx->stream = (hipStream_t)hip_stream;
    ctx->own_stream = false;
    return APD_OK;
}

// Waits for the stream and reads the sticky device status word back; a raised bit is reported once and cleared.
static int sync_and_report(apd_context *ctx)
{
    uint32_t st = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&st, ctx->d_status, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (st & 1u) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(st), ctx->stream));
        ctx->last_error = "unpack met a pair score that no alignment kernel wrote (NaN poison survived): launch cut short or skipped";
        return APD_ERR_INCOMPLETE;
    }
    return APD_OK;
}

extern "C" int apd_synchronize(apd_context *ctx)
{
    if (!ctx) return APD_ERR_INVALID_ARG;
    HIP_TRY(ctx, bind_device(ctx));
    return sync_and_report(ctx);
}

extern "C" int apd_debug_affinity_probe(apd_context *bound, apd_context *checked, int *enabled)
{
    if (!bound || !checked) return APD_ERR_INVALID_ARG;
    if (enabled) *enabled = apd::affinity_debug() ? 1 : 0;
    HIP_TRY(bound, bind_device(bound));
    APD_AFFINITY(checked, "affinity probe");                              // what every allocation / event / launch in the library does
    return APD_OK;
}

extern "C" int apd_stream_busy(apd_context *ctx, int *busy)
{
    if (!ctx || !busy) return APD_ERR_INVALID_ARG;
    HIP_TRY(ctx, bind_device(ctx));
    const hipError_t e = hipStreamQuery(ctx->stream);
    if (e != hipSuccess && e != hipErrorNotReady) { ctx->last_error = std::string("hipStreamQuery: ") + hipGetErrorString(e); return APD_ERR_HIP; }
    *busy = e == hipErrorNotReady ? 1 : 0;
    return APD_OK;
}

extern "C" int apd_set_fault_injection(apd_context *ctx, uint32_t drop_tiles)
{
    if (!ctx) return APD_ERR_INVALID_ARG;
    ctx->drop_tiles = drop_tiles;
    return APD_OK;
}

extern "C" const char *apd_last_error(apd_context *ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

extern "C" int apd_set_timing(apd_context *ctx, int enabled)
{
    if (!ctx) return APD_ERR_INVALID_ARG;
    ctx->timing = enabled != 0;
    ctx->timed = false;
    return APD_OK;
}

extern "C" float apd_last_kernel_ms(apd_context *ctx)
{
    if (!ctx || !ctx->timed) return -1.0f;
    float ms = -1.0f;
    if (hipEventSynchronize(ctx->ev1) != hipSuccess) return -1.0f;
    if (hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1) != hipSuccess) return -1.0f;
    return ms;
}

extern "C" int apd_set_variant(apd_context *ctx, int variant)
{
    if (!ctx) return APD_ERR_INVALID_ARG;
    ctx->variant = variant;
    return APD_OK;
}

extern "C" int apd_set_distance_mode(apd_context *ctx, int mode, float tau)
{
    if (!ctx || mode < 0 || mode > 2 || !(tau >= 0.0f)) return APD_ERR_INVALID_ARG;
    ctx->distance_mode = mode;
    if (tau > 0.0f) ctx->tau = tau;
    return APD_OK;
}

extern "C" int apd_selftest(apd_context *ctx)
{
    if (!ctx) return APD_ERR_INVALID_ARG;
    HIP_TRY(ctx, bind_device(ctx));
    int rc = ensure_ws(ctx, &ctx->ws_misc, &ctx->ws_misc_bytes, 256);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemsetAsync(ctx->ws_misc, 0, sizeof(int), ctx->stream));
    HIP_TRY(ctx, launch_selftest((int *)ctx->ws_misc, ctx->stream));
    int ok = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&ok, ctx->ws_misc, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (!ok) { ctx->last_error = "DPP wave_shr/wave_shl self-test failed"; return APD_ERR_HIP; }
    return APD_OK;
}

extern "C" int apd_selftest_sqrt(apd_context *ctx, uint32_t first_bits, uint64_t count, uint64_t *mismatches, uint32_t *first_mismatch,
                                 uint64_t *raw_ulp_hist)
{
    if (!ctx || !mismatches || count > (1ull << 32)) return APD_ERR_INVALID_ARG;
    HIP_TRY(ctx, bind_device(ctx));
    int rc = ensure_ws(ctx, &ctx->ws_misc, &ctx->ws_misc_bytes, 256);
    if (rc) return rc;
    unsigned long long h[7] = {0, ~0ull, 0, 0, 0, 0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(ctx->ws_misc, h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));                       // h is a stack buffer
    if (count) HIP_TRY(ctx, launch_sqrt_sweep(first_bits, count, (unsigned long long *)ctx->ws_misc, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(h, ctx->ws_misc, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *mismatches = h[0];
    if (first_mismatch) *first_mismatch = h[0] ? (uint32_t)(h[1] - 1ull) : 0u;
    if (raw_ulp_hist) for (int k = 0; k < 5; ++k) raw_ulp_hist[k] = h[2 + k];
    return APD_OK;
}

// --------------------------------------------------------------------- Discovery::alignment_params

extern "C" int apd_discovery_alignment_params(const apd_align_config *cfg, uint64_t n_size, apd_alignment_params *out)
{
    if (!cfg || !out) return APD_ERR_INVALID_ARG;
    const float p = cfg->warping_band_percentage * (float)n_size;        // discovery.rs:40
    uint64_t band;
    if (!(p > 0.0f)) band = 0;
    else if (p >= 18446744073709551616.0f) band = UINT64_MAX;
    else band = (uint64_t)p;
    out->warping_band = band;
    out->insertion_penalty = cfg->insertion_penalty;                     // :41
    out->match_penalty = cfg->match_penalty;                             // :42
    out->deletion_penalty = cfg->deletion_penalty;                       // :43
    return APD_OK;
}

// ------------------------------------------------------------------------------------- batch

extern "C" int apd_batch_create(apd_context *ctx, const float *frames, const uint64_t *offsets, uint32_t n_seq,
                                uint32_t dim, int frames_on_device, apd_batch **out)
{
    if (!ctx || !offsets || !out || dim == 0 || dim > 1024) return APD_ERR_INVALID_ARG;
    *out = nullptr;
    HIP_TRY(ctx, bind_device(ctx));
    const uint64_t total = offsets[n_seq];
    if (total > 0 && !frames) return APD_ERR_INVALID_ARG;
    if (total + 2ull * n_seq >= (1ull << 32) || offsets[0] != 0) return APD_ERR_INVALID_ARG;
    apd_batch *b = new (std::nothrow) apd_batch();
    if (!b) return APD_ERR_OOM;
    // resident frames carry kernel_dim(dim) >= dim components (zero fill: distances unchanged, dtw_common.h), then the squared norm
    b->ctx = ctx; b->n_seq = n_seq; b->src_dim = dim; b->dim = kernel_dim(dim); b->dpad = (b->dim + 4) & ~3u; b->total_frames = total;
    b->min_len = 0xFFFFFFFFu; b->max_len = 0;
    for (uint32_t s = 0; s < n_seq; ++s) {
        if (offsets[s + 1] < offsets[s]) { delete b; return APD_ERR_INVALID_ARG; }
        const uint32_t len = (uint32_t)(offsets[s + 1] - offsets[s]);
        b->min_len = std::min(b->min_len, len);
        b->max_len = std::max(b->max_len, len);
    }
    if (n_seq == 0) b->min_len = 0;
    // Resident order: longest sequence first (length_order), so that the 16 sequences of a tile row have like lengths --
    // one kernel geometry fits the whole tile -- and the most expensive tiles of a launch start first.
    length_order(offsets, n_seq, b->order);
    b->offsets.assign(n_seq + 1, 0);                                     // offsets of the RESIDENT order
    // device metadata, one allocation and one copy: [seq_off | src_off | order | flags]
    const size_t m1 = (size_t)n_seq + 1;
    b->h_meta.assign(4 * m1 + 4, 0u);
    uint32_t *off32 = b->h_meta.data(), *src32 = off32 + m1, *ord32 = src32 + m1;
    for (uint32_t p = 0; p < n_seq; ++p) {
        b->offsets[p + 1] = b->offsets[p] + (offsets[b->order[p] + 1] - offsets[b->order[p]]);
        src32[p] = (uint32_t)offsets[b->order[p]];
        ord32[p] = b->order[p];
    }
    for (uint32_t p = 0; p <= n_seq; ++p) off32[p] = (uint32_t)b->offsets[p] + 2 * p;   // two sentinel frames behind every sequence
    auto fail = [&](int rc) { apd_batch_destroy(b); return rc; };
    const uint64_t padded_frames = total + 2ull * n_seq;
    const size_t padded_bytes = std::max<size_t>((size_t)padded_frames * b->dpad * sizeof(float), 16);
    b->frames_bytes = padded_bytes < 0xFFFFFE00ull ? (uint32_t)padded_bytes : 0u;
    if (hipMalloc((void **)&b->d_frames, padded_bytes) != hipSuccess) return fail(APD_ERR_OOM);
    if (hipMalloc((void **)&b->d_meta, b->h_meta.size() * sizeof(uint32_t)) != hipSuccess) return fail(APD_ERR_OOM);
    b->d_seq_off = b->d_meta; b->d_src_off = b->d_meta + m1; b->d_order = b->d_meta + 2 * m1; b->d_flags = b->d_meta + 3 * m1;
    b->d_seq_nmax = reinterpret_cast<float *>(b->d_meta + 3 * m1 + 4);
    if (hipMemcpyAsync(b->d_meta, b->h_meta.data(), b->h_meta.size() * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
        return fail(APD_ERR_HIP);                                         // h_meta lives as long as the batch: no sync needed
    ctx->batches.insert(b);
    const int rc_fill = apd_batch_refill(ctx, b, frames, frames_on_device);
    if (rc_fill != APD_OK) return fail(rc_fill);
    *out = b;
    return APD_OK;
}

extern "C" int apd_batch_refill(apd_context *ctx, apd_batch *b, const float *frames, int frames_on_device)
{
    if (!ctx || !b || b->ctx != ctx) return APD_ERR_INVALID_ARG;
    const uint64_t total = b->total_frames, padded_frames = total + 2ull * b->n_seq;
    if (total > 0 && !frames) return APD_ERR_INVALID_ARG;
    HIP_TRY(ctx, bind_device(ctx));
    b->nonfinite = -1;
    HIP_TRY(ctx, hipMemsetAsync(b->d_flags, 0, (4 + (size_t)b->n_seq + 1) * sizeof(uint32_t), ctx->stream));   // flags and the per-sequence norm maxima
    if (padded_frames > 0) {
        const float *d_src = frames;
        float *d_tmp = nullptr;
        const uint32_t dim = b->src_dim;
        if (!frames_on_device && total > 0) {
            if (hipMalloc((void **)&d_tmp, (size_t)total * dim * sizeof(float)) != hipSuccess) return APD_ERR_OOM;
            if (hipMemcpyAsync(d_tmp, frames, (size_t)total * dim * sizeof(float), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) {
                hipFree(d_tmp);
                return APD_ERR_HIP;
            }
            d_src = d_tmp;
        }
        hipError_t e = launch_pad(d_src, b->d_frames, b->d_seq_off, b->d_src_off, b->n_seq, padded_frames, dim, b->dim, b->dpad, b->d_flags, b->d_seq_nmax, ctx->stream);
        if (d_tmp) { hipStreamSynchronize(ctx->stream); hipFree(d_tmp); }
        if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return APD_ERR_HIP; }
    }
    return APD_OK;
}

// The repack kernel's verdict on the frames, read back once per fill (4 bytes, one stream sync).
static int batch_nonfinite(apd_context *ctx, const apd_batch *b, bool *out)
{
    if (b->nonfinite < 0) {
        uint32_t f = 0;
        HIP_TRY(ctx, hipMemcpyAsync(&f, b->d_flags, sizeof(f), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        b->nonfinite = f ? 1 : 0;
    }
    *out = b->nonfinite == 1;
    return APD_OK;
}

extern "C" int apd_batch_nonfinite(apd_context *ctx, const apd_batch *b, int *nonfinite)
{
    if (!ctx || !b || !nonfinite || b->ctx != ctx) return APD_ERR_INVALID_ARG;
    HIP_TRY(ctx, bind_device(ctx));
    bool nf = false;
    const int rc = batch_nonfinite(ctx, b, &nf);
    *nonfinite = nf ? 1 : 0;
    return rc;
}

static void release_batch_device_memory(apd_batch *b)
{
    for (auto &kv : b->tile_cache) if (kv.second.d_tiles) hipFree(kv.second.d_tiles);
    b->tile_cache.clear();
    if (b->d_frames) hipFree(b->d_frames);
    if (b->d_meta) hipFree(b->d_meta);
    b->d_frames = nullptr; b->d_meta = nullptr; b->d_seq_off = nullptr; b->d_src_off = nullptr; b->d_order = nullptr; b->d_flags = nullptr; b->d_seq_nmax = nullptr;
}

extern "C" int apd_batch_destroy(apd_batch *b)
{
    if (!b) return APD_ERR_INVALID_ARG;
    if (b->ctx) {                                                        // else: orphaned by apd_destroy, device memory already released
        bind_device(b->ctx);
        hipStreamSynchronize(b->ctx->stream);
        release_batch_device_memory(b);
        b->ctx->batches.erase(b);
    }
    delete b;
    return APD_OK;
}

extern "C" uint32_t apd_batch_len(const apd_batch *b) { return b ? b->n_seq : 0; }

// ------------------------------------------------------------------------------------- tiles

extern "C" uint32_t apd_tile_size(void) { return kTile; }

extern "C" uint64_t apd_num_tiles(uint32_t n_seq)
{
    const uint64_t side = tiles_side(n_seq);
    return side * (side + 1) / 2;
}

extern "C" uint64_t apd_rank_tiles(uint32_t n_seq, uint32_t rank, uint32_t world)
{
    if (world == 0 || rank >= world) return 0;
    const uint64_t t = apd_num_tiles(n_seq);
    return t / world + (rank < t % world ? 1 : 0);
}

extern "C" uint64_t apd_slab_floats(uint32_t n_seq, uint32_t world)
{
    if (world == 0) return 0;
    const uint64_t t = apd_num_tiles(n_seq);
    return ((t + world - 1) / world) * 2 * kSlotsPerTile;
}

extern "C" int apd_rank_tile_list(uint32_t n_seq, uint32_t rank, uint32_t world, uint32_t *tile_ab, uint64_t capacity,
                                  uint64_t *n_tiles)
{
    if (world == 0 || rank >= world || !n_tiles) return APD_ERR_INVALID_ARG;
    std::vector<uint2> tiles;
    rank_tile_list(n_seq, rank, world, tiles);
    *n_tiles = tiles.size();
    if (tile_ab) {
        if (capacity < tiles.size()) return APD_ERR_INVALID_ARG;
        for (size_t t = 0; t < tiles.size(); ++t) { tile_ab[2 * t] = tiles[t].x; tile_ab[2 * t + 1] = tiles[t].y; }
    }
    return APD_OK;
}

extern "C" int apd_length_order(const uint64_t *offsets, uint32_t n_seq, uint32_t *order)
{
    if (!offsets || (n_seq && !order)) return APD_ERR_INVALID_ARG;
    std::vector<uint32_t> o;
    length_order(offsets, n_seq, o);
    std::copy(o.begin(), o.end(), order);
    return APD_OK;
}

extern "C" int apd_unpack_tiles_host(const uint64_t *offsets, uint32_t n_seq, uint32_t world, const float *gathered, float *out)
{
    if (!offsets || world == 0 || (n_seq && (!gathered || !out))) return APD_ERR_INVALID_ARG;
    std::vector<uint32_t> order;
    length_order(offsets, n_seq, order);
    const uint64_t slab = apd_slab_floats(n_seq, world);
    std::memset(out, 0, (size_t)n_seq * n_seq * sizeof(float));                 // alignments.rs:21-23
    const uint32_t side = tiles_side(n_seq);
    uint64_t g = 0;
    for (uint32_t ta = 0; ta < side; ++ta)
        for (uint32_t tb = ta; tb < side; ++tb, ++g) {
            const float *t = gathered + (g % world) * slab + (g / world) * 2 * kSlotsPerTile;
            for (uint32_t sa = 0; sa < kTile; ++sa)
                for (uint32_t sb = 0; sb < kTile; ++sb) {
                    const uint32_t pa = ta * kTile + sa, pb = tb * kTile + sb;   // positions in the resident order
                    if (pa < pb && pb < n_seq) {
                        const uint32_t a = order[pa], b = order[pb];
                        out[(uint64_t)a * n_seq + b] = t[sa * kTile + sb];
                        out[(uint64_t)b * n_seq + a] = t[kSlotsPerTile + sa * kTile + sb];
                    }
                }
        }
    return APD_OK;
}

static int check_lengths(const apd_batch *b)
{
    if (b->n_seq > 0 && b->min_len == 0) return APD_ERR_EMPTY_SEQUENCE;
    return APD_OK;
}

// Which kernel sweeps which tile.  Tiles are grouped by the kernel geometry their widest pair needs (w is bounded per tile
// from the lengths of its 32 sequences), one launch per group: a few long or unequal sequences do not force every pair
// onto a wide kernel.  The plan (device tile list + classes) is cached in the batch.
static int build_tile_plan(apd_context *ctx, const apd_batch *batch, const BandSpec &band, uint32_t rank, uint32_t world,
                           bool fast_ok, bool uniform_pen, bool fast_shift, apd_batch::TilePlan &plan_out)
{
    apd_batch::TilePlan plan;                                             // built locally, published only when complete
    std::vector<uint2> tiles;
    rank_tile_list(batch->n_seq, rank, world, tiles);
    // per tile-row (16 sequences) min / max length
    const uint32_t side = tiles_side(batch->n_seq);
    std::vector<uint32_t> lo(side, 0xFFFFFFFFu), hi(side, 0);
    for (uint32_t s = 0; s < batch->n_seq; ++s) {
        const uint32_t len = (uint32_t)(batch->offsets[s + 1] - batch->offsets[s]);
        lo[s / kTile] = std::min(lo[s / kTile], len);
        hi[s / kTile] = std::max(hi[s / kTile], len);
    }
    std::map<int, std::vector<uint4>> groups;
    std::map<int, uint32_t> wmax, nmax;
    for (uint32_t t = 0; t < tiles.size(); ++t) {
        const uint32_t mx = std::max(hi[tiles[t].x], hi[tiles[t].y]), mn = std::min(lo[tiles[t].x], lo[tiles[t].y]);
        const uint32_t band_ub = band.use_explicit ? band.explicit_band : host_band_from_pct(band.pct, mx);
        const uint32_t w = std::max(std::min(band_ub, mx), mx - mn) + 2;   // >= w of every pair of the tile
        int key = fast_ok ? pick_geometry_key(2 * w + 1, batch->dim, ctx->variant, uniform_pen, fast_shift) : 0;
        // the band binds nowhere in this tile (band >= longest - 3 for its longest sequence, hence for all) and the penalties
        // are equal: both ordered scores are one number, swept over column strips (dtw_full.h).  Not for very short columns,
        // where four pairs per wavefront in band form keep more lanes busy.
        const uint32_t cols = std::min(hi[tiles[t].x], hi[tiles[t].y]);
        const bool never_binds = mx >= 3 && std::min(band_ub, mx) >= mx - 3;
        if (fast_ok && uniform_pen && never_binds && (cols >= 49 || (ctx->variant >= 20000 && ctx->variant < 30000))) {
            const int fk = pick_full_key(cols > 0 ? cols - 1 : 0, mx, batch->dim, ctx->variant);
            if (fk != 0) key = fk;
        } else if (fast_ok && mx >= 3 && cols >= 49 &&                          // (any penalties: unequal ones take the literal select)
                   (2ull * w + 1 >= cols ||                                   // band at least as wide as the short side
                    key == 0 ||                                               // no band-form kernel fits: anything beats the generic one
                    ctx->variant >= 30000)) {
            // the band binds, but is wider than the short side of the tile's pairs (w grows with |n - m|, alignments.rs:173):
            // in band coordinates most offsets of such a pair lie outside it; column strips with masked band edges fit
            const int bk = pick_banded_strip_key(cols - 1, mx, batch->dim, ctx->variant);
            if (bk != 0) key = bk;
        }
        groups[key].push_back(make_uint4(tiles[t].x, tiles[t].y, t, 0));
        wmax[key] = std::max(wmax[key], w);
        nmax[key] = std::max(nmax[key], mx);
    }
    // Every class is a launch of its own, and launches of one stream do not overlap: a full-matrix class of a few dozen
    // tiles would run at a fraction of the machine.  Small classes (all of them, in a small batch) move to the geometry
    // that is best for the full-matrix tiles as a whole.
    if (ctx->variant == 0) {
        auto dims_of = [&](const uint4 &t, uint32_t *cols, uint32_t *rows) {
            *rows = std::max(hi[t.x], hi[t.y]);
            const uint32_t c = std::min(hi[t.x], hi[t.y]);
            *cols = c > 0 ? c - 1 : 0;
        };
        for (int family = 20000; family <= 30000; family += 10000) {      // one DP (band never binds) / two DPs (banded)
            auto in_family = [&](int k) { return k >= family && k < family + 10000; };
            size_t n_fam = 0;
            for (auto &g : groups) if (in_family(g.first)) n_fam += g.second.size();
            if (n_fam == 0) continue;
            const int max_ppw = n_fam * kSlotsPerTile < 8192 ? 1 : 4;    // too few pairs to fill the GPU: one wavefront each
            int global_key = 0;
            double global_cost = INFINITY;
            for (int ppw = 1; ppw <= max_ppw; ppw *= 2)
                for (int cw = 5; cw <= max_strip_columns(batch->dim); cw += 2) {
                    const int k = family + ppw * 100 + cw;
                    double total = 0.0;
                    for (auto &g : groups) {
                        if (!in_family(g.first)) continue;
                        for (const uint4 &t : g.second) { uint32_t c, r; dims_of(t, &c, &r); total += full_key_cost(c, r, batch->dim, k); }
                    }
                    if (total < global_cost) { global_cost = total; global_key = k; }
                }
            const size_t min_class = n_fam < 2048 ? n_fam + 1 : 256;
            if (global_key == 0) continue;
            std::vector<int> small;
            for (auto &g : groups) if (in_family(g.first) && g.first != global_key && g.second.size() < min_class) small.push_back(g.first);
            for (int k : small) {
                std::vector<uint4> &from = groups[k], &to = groups[global_key];
                to.insert(to.end(), from.begin(), from.end());
                wmax[global_key] = std::max(wmax[global_key], wmax[k]);
                nmax[global_key] = std::max(nmax[global_key], nmax[k]);
                groups.erase(k);
                wmax.erase(k);
                nmax.erase(k);
            }
            std::sort(groups[global_key].begin(), groups[global_key].end(), [](const uint4 &a, const uint4 &b) { return a.z < b.z; });
        }
    }
    std::vector<uint4> flat;
    for (auto &g : groups) {
        plan.classes.push_back(apd_batch::TileClass{g.first, (uint32_t)flat.size(), (uint32_t)g.second.size(), wmax[g.first], nmax[g.first]});
        flat.insert(flat.end(), g.second.begin(), g.second.end());
    }
    if (std::getenv("APD_DEBUG_PLAN"))                                  // tuning aid: which kernel geometry got how many tiles
        for (const apd_batch::TileClass &tc : plan.classes)
            std::fprintf(stderr, "[apd] rank %u/%u: geometry %d: %u tiles, w_max %u, n_max %u\n", rank, world, tc.geom_key, tc.count, tc.w_max, tc.n_max);
    HIP_TRY(ctx, hipMalloc((void **)&plan.d_tiles, std::max<size_t>(flat.size(), 1) * sizeof(uint4)));
    if (!flat.empty()) {
        hipError_t e = hipMemcpyAsync(plan.d_tiles, flat.data(), flat.size() * sizeof(uint4), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {                                            // never leave a half-made plan in the cache
            hipFree(plan.d_tiles);
            ctx->last_error = std::string("tile plan upload: ") + hipGetErrorString(e);
            return APD_ERR_HIP;
        }
    }
    plan_out = std::move(plan);
    return APD_OK;
}

static int align_tiles_impl(apd_context *ctx, const apd_batch *batch, const BandSpec &band, uint32_t rank,
                            uint32_t world, float *d_slab)
{
    if (!ctx || !batch || !d_slab || world == 0 || rank >= world || batch->ctx != ctx) return APD_ERR_INVALID_ARG;
    int rc = check_lengths(batch);
    if (rc) return rc;
    HIP_TRY(ctx, bind_device(ctx));
    const bool pens_ok = (band.ins > 0.0f) && (band.del > 0.0f) && (band.mat > 0.0f) && (band.ins < INFINITY) &&
                         (band.del < INFINITY) && (band.mat < INFINITY);   // the systolic kernel needs pen * INF = INF
    // A NaN / infinite feature anywhere in the batch means: literal kernel only (the fast kernels' selects and sentinels assume
    // finite features).  The repack kernel leaves that verdict in batch->d_flags[0]; it is consumed ON THE DEVICE: the fast
    // kernels return at once when it is set, and a fallback launch of the generic kernel over the same tiles returns at once when
    // it is clear (`device_select`).  No host round trip, nothing blocks: the call only enqueues.  The one exception: a band so
    // wide that the generic kernel cannot hold it in LDS -- then the host reads the flag (a stream synchronisation), as the fast
    // kernels are the only ones that can run.
    const bool static_fast = pens_ok && batch->frames_bytes != 0;
    bool nonfinite = false, device_select = false;
    if (static_fast) {
        const uint32_t band_ub = band.use_explicit ? band.explicit_band : host_band_from_pct(band.pct, batch->max_len);
        const uint32_t w_all = std::max(std::min(band_ub, batch->max_len), batch->max_len - batch->min_len) + 2;   // >= w of every pair
        device_select = generic_fallback_fits(w_all) && apd_num_tiles(batch->n_seq) * kSlotsPerTile <= 0xFFFFFFFFull &&
                        !std::getenv("APD_HOST_SELECT");                  // (the env: the old host-side choice, for A/B timing)
        if (!device_select) {
            rc = batch_nonfinite(ctx, batch, &nonfinite);
            if (rc) return rc;
        }
    }
    const bool fast_ok = static_fast && !nonfinite;
    // strict mode: distances computed operation for operation as numerics.rs:114-120 in every kernel family.  The band kernels
    // keep the fast select with unit penalties (it picks the reference's predecessor for every non-NaN input: dtw_systolic.h, <.., true,
    // false>) and take the literal comparison chain with any others; the strip kernels go through their literal-select path, which
    // is why `uniform_pen` is cleared here -- it steers the strip / wide families only (launch_align_chunk, dtw_generic.hip).
    const bool strict = ctx->distance_mode == 2;
    const bool uniform_pen = (band.ins == band.del) && (band.del == band.mat) && !strict;
    char keybuf[160];
    uint32_t pct_bits;
    std::memcpy(&pct_bits, &band.pct, sizeof(pct_bits));
    // the band kernel's hybrid form with unit penalties moves its column window with one DPP instruction per register at any
    // group size (dtw_systolic.h, MASKED_FETCH): 8- and 32-lane groups cost no more than 16- and 64-lane ones there
    const bool fast_shift = band.ins == 1.0f && band.del == 1.0f && band.mat == 1.0f && ctx->distance_mode == 1 && batch->dim >= 8;
    std::snprintf(keybuf, sizeof(keybuf), "%u/%u/%08x/%u/%d/%d/%d/%d/%d", rank, world, pct_bits, band.explicit_band, band.use_explicit,
                  ctx->variant, (int)fast_ok, (int)uniform_pen, (int)fast_shift);   // everything the choice of kernels depends on
    auto cached = batch->tile_cache.find(keybuf);
    if (cached == batch->tile_cache.end()) {
        apd_batch::TilePlan fresh;
        rc = build_tile_plan(ctx, batch, band, rank, world, fast_ok, uniform_pen, fast_shift, fresh);
        if (rc) return rc;
        cached = batch->tile_cache.emplace(keybuf, std::move(fresh)).first;
    }
    const apd_batch::TilePlan &plan = cached->second;
    // Poison: every score slot of the rank's slab starts as NaN, so a pair that no kernel writes (a launch cut short, a
    // skipped class) reaches the matrix as NaN and raises APD_ERR_INCOMPLETE in the unpack -- never a stale or zero distance.
    APD_AFFINITY(ctx, "alignment launches");
    HIP_TRY(ctx, hipMemsetAsync(d_slab, 0xFF, apd_slab_floats(batch->n_seq, world) * sizeof(float), ctx->stream));
    AlignLaunch L{};
    L.d_frames = batch->d_frames; L.frames_bytes = batch->frames_bytes; L.d_seq_off = batch->d_seq_off; L.d_seq_nmax = batch->d_seq_nmax;
    L.n_seq = batch->n_seq; L.dim = batch->dim; L.dpad = batch->dpad; L.band = band; L.d_slab = d_slab;
    L.variant = ctx->variant;
    L.hybrid = ctx->distance_mode == 1; L.strict = strict; L.tau = ctx->tau;
    L.d_nonfinite = device_select ? batch->d_flags : nullptr;
    if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    // Classes are independent (disjoint tiles, disjoint slab regions): with more than one, their launches are spread over
    // side streams forked from and joined back into the context's stream, so that a class of a few tiles does not hold the
    // GPU alone (a ragged banded corpus splits into a dozen geometries).
    const bool fan_out = plan.classes.size() > 1 && ctx->side[0] != nullptr;
    if (fan_out) {
        HIP_TRY(ctx, hipEventRecord(ctx->fork, ctx->stream));
        for (int k = 0; k < apd_context::kSideStreams; ++k) HIP_TRY(ctx, hipStreamWaitEvent(ctx->side[k], ctx->fork, 0));
    }
    int rc_launch = APD_OK;
    size_t ci = 0;
    for (const apd_batch::TileClass &tc : plan.classes) {
        L.d_tiles = plan.d_tiles + tc.first; L.n_tiles = tc.count; L.w_max = tc.w_max; L.n_max = tc.n_max;
        if (ctx->drop_tiles) L.n_tiles -= std::min(L.n_tiles, ctx->drop_tiles);   // fault injection (apd_set_fault_injection)
        int status = APD_OK;
        hipStream_t s = fan_out ? ctx->side[ci++ % apd_context::kSideStreams] : ctx->stream;
        hipError_t e = launch_align(L, tc.geom_key, s, ctx->last_error, &status);
        if (e != hipSuccess) { ctx->last_error = std::string("launch_align: ") + hipGetErrorString(e); rc_launch = APD_ERR_HIP; break; }
        if (status != APD_OK) { rc_launch = status; break; }
    }
    if (fan_out)                                                        // always join, also after a failed launch
        for (int k = 0; k < apd_context::kSideStreams; ++k) {
            HIP_TRY(ctx, hipEventRecord(ctx->side_done[k], ctx->side[k]));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->side_done[k], 0));
        }
    if (rc_launch != APD_OK) return rc_launch;
    if (device_select) {
        // the literal kernel behind the fast ones, idle unless the flag is raised -- over the classes a FAST kernel took (a class
        // that already runs the literal kernel, geometry 0, ignores the flag and must not be aligned twice); neighbouring classes
        // are contiguous in the tile list and share a launch
        size_t k = 0;
        while (k < plan.classes.size()) {
            if (plan.classes[k].geom_key == 0) { ++k; continue; }
            uint32_t first = plan.classes[k].first, total = 0, w_max = 0, n_max = 0;
            for (; k < plan.classes.size() && plan.classes[k].geom_key != 0 && plan.classes[k].first == first + total; ++k) {
                total += plan.classes[k].count; w_max = std::max(w_max, plan.classes[k].w_max); n_max = std::max(n_max, plan.classes[k].n_max);
            }
            L.d_tiles = plan.d_tiles + first; L.n_tiles = total; L.w_max = w_max; L.n_max = n_max;
            if (ctx->drop_tiles) L.n_tiles -= std::min(L.n_tiles, ctx->drop_tiles);
            bool fits = true;
            const hipError_t e = launch_generic_fallback(L, ctx->stream, &fits);
            if (e != hipSuccess || !fits) { ctx->last_error = std::string("launch_generic_fallback: ") + (fits ? hipGetErrorString(e) : "band too wide"); return APD_ERR_HIP; }
        }
    }
    if (ctx->timing) { HIP_TRY(ctx, hipEventRecord(ctx->ev1, ctx->stream)); ctx->timed = true; }
    return APD_OK;
}

static BandSpec band_from_cfg(const apd_align_config *cfg)
{
    BandSpec b{};
    b.pct = cfg->warping_band_percentage; b.use_explicit = 0; b.explicit_band = 0;
    b.ins = cfg->insertion_penalty; b.del = cfg->deletion_penalty; b.mat = cfg->match_penalty;
    return b;
}

extern "C" int apd_align_tiles_async(apd_context *ctx, const apd_batch *batch, const apd_align_config *cfg,
                                     uint32_t rank, uint32_t world, float *d_slab)
{
    if (!cfg) return APD_ERR_INVALID_ARG;
    return align_tiles_impl(ctx, batch, band_from_cfg(cfg), rank, world, d_slab);
}

extern "C" int apd_unpack_tiles_async(apd_context *ctx, const apd_batch *batch, uint32_t world, const float *d_gathered,
                                      float *d_out)
{
    if (!ctx || !batch || batch->ctx != ctx || !d_gathered || !d_out || world == 0) return APD_ERR_INVALID_ARG;
    const uint32_t n_seq = batch->n_seq;
    HIP_TRY(ctx, bind_device(ctx));
    APD_AFFINITY(ctx, "unpack launch");
    // the unpack writes every entry, the zero diagonal (alignments.rs:21-23) included; anything it fails to write stays NaN
    HIP_TRY(ctx, hipMemsetAsync(d_out, 0xFF, (size_t)n_seq * n_seq * sizeof(float), ctx->stream));
    HIP_TRY(ctx, launch_unpack(d_gathered, d_out, batch->d_order, n_seq, world, apd_slab_floats(n_seq, world), batch->d_flags,
                               ctx->d_status, ctx->stream));
    return APD_OK;
}

static int align_all_device_impl(apd_context *ctx, const apd_batch *batch, const BandSpec &band, float *d_out)
{
    if (!ctx || !batch || !d_out) return APD_ERR_INVALID_ARG;
    const size_t slab_bytes = std::max<size_t>(apd_slab_floats(batch->n_seq, 1) * sizeof(float), 16);
    int rc = ensure_ws(ctx, &ctx->ws_slab, &ctx->ws_slab_bytes, slab_bytes);
    if (rc) return rc;
    rc = align_tiles_impl(ctx, batch, band, 0, 1, (float *)ctx->ws_slab);
    if (rc) return rc;
    return apd_unpack_tiles_async(ctx, batch, 1, (const float *)ctx->ws_slab, d_out);
}

extern "C" int apd_align_all_device_async(apd_context *ctx, const apd_batch *batch, const apd_align_config *cfg,
                                          float *d_out)
{
    if (!cfg) return APD_ERR_INVALID_ARG;
    return align_all_device_impl(ctx, batch, band_from_cfg(cfg), d_out);
}

extern "C" int apd_align_all(apd_context *ctx, const apd_batch *batch, const apd_align_config *cfg, float *out)
{
    if (!ctx || !batch || !cfg || (!out && batch->n_seq)) return APD_ERR_INVALID_ARG;
    if (batch->n_seq == 0) return APD_OK;
    HIP_TRY(ctx, bind_device(ctx));
    const size_t bytes = (size_t)batch->n_seq * batch->n_seq * sizeof(float);
    float *d_out = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&d_out, bytes));
    int rc = align_all_device_impl(ctx, batch, band_from_cfg(cfg), d_out);
    if (rc == APD_OK) {
        hipError_t e = hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); rc = APD_ERR_HIP; }
        else rc = sync_and_report(ctx);                                   // APD_ERR_INCOMPLETE: `out` holds NaN where no score was written
    }
    hipFree(d_out);
    return rc;
}

extern "C" int apd_align_pair(apd_context *ctx, const float *x, uint64_t n, const float *y, uint64_t m, uint32_t dim,
                              const apd_alignment_params *params, float *score)
{
    if (!ctx || !params || !score || dim == 0) return APD_ERR_INVALID_ARG;
    if (n == 0 && m == 0) { *score = INFINITY; return APD_OK; }           // alignments.rs:117-118
    if (n == 0 || m == 0) return APD_ERR_EMPTY_SEQUENCE;                  // usize underflow at :120
    if (!x || !y) return APD_ERR_INVALID_ARG;
    std::vector<float> frames((n + m) * (size_t)dim);
    std::memcpy(frames.data(), x, n * (size_t)dim * sizeof(float));
    std::memcpy(frames.data() + n * (size_t)dim, y, m * (size_t)dim * sizeof(float));
    const uint64_t offsets[3] = {0, n, n + m};
    // A host that loops over Alignment::construct_alignment (alignments.rs:165) mostly aligns pairs of the SAME lengths (fixed
    // windows): the context keeps the last pair's two-sequence batch and refills it (no allocation, no plan rebuild) when the next
    // pair has the same (n, m, dim); any other shape replaces it.
    int rc = APD_OK;
    if (ctx->pair_batch && (ctx->pair_n != n || ctx->pair_m != m || ctx->pair_dim != dim)) { apd_batch_destroy(ctx->pair_batch); ctx->pair_batch = nullptr; }
    if (!ctx->pair_batch) {
        rc = apd_batch_create(ctx, frames.data(), offsets, 2, dim, 0, &ctx->pair_batch);
        if (rc) { ctx->pair_batch = nullptr; return rc; }
        ctx->pair_n = n; ctx->pair_m = m; ctx->pair_dim = dim;
    } else {
        rc = apd_batch_refill(ctx, ctx->pair_batch, frames.data(), 0);   // host frames: the refill has consumed them when it returns
        if (rc) return rc;
    }
    apd_batch *b = ctx->pair_batch;
    BandSpec band{};
    band.use_explicit = 1;
    band.explicit_band = params->warping_band > 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)params->warping_band;
    band.ins = params->insertion_penalty; band.del = params->deletion_penalty; band.mat = params->match_penalty;
    rc = ensure_ws(ctx, &ctx->ws_misc, &ctx->ws_misc_bytes, 256);
    if (rc) return rc;
    float *d_out = (float *)ctx->ws_misc;
    rc = align_all_device_impl(ctx, b, band, d_out);
    float host[4] = {0, 0, 0, 0};
    if (rc == APD_OK) {
        hipError_t e = hipMemcpyAsync(host, d_out, sizeof(host), hipMemcpyDeviceToHost, ctx->stream);
        if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); rc = APD_ERR_HIP; }
        else rc = sync_and_report(ctx);
    }
    if (rc == APD_OK) *score = host[1];                                   // out[0*2+1] = score(x, y)
    return rc;
}

// ------------------------------------------------------------------------------- work accounting

extern "C" int apd_align_work(const uint64_t *offsets, uint32_t n_seq, uint32_t dim, const apd_align_config *cfg,
                              uint32_t rank, uint32_t world, uint64_t *pairs, uint64_t *cells, uint64_t *alg_bytes)
{
    if (!offsets || !cfg || world == 0 || rank >= world) return APD_ERR_INVALID_ARG;
    const BandSpec band = band_from_cfg(cfg);
    std::vector<uint2> tiles;
    rank_tile_list(n_seq, rank, world, tiles);
    std::vector<uint32_t> order;
    length_order(offsets, n_seq, order);                                  // tiles are cut from the resident order
    uint64_t np = 0, nc = 0, nb = 0;
    for (const uint2 &t : tiles)
        for (uint32_t sa = 0; sa < kTile; ++sa)
            for (uint32_t sb = 0; sb < kTile; ++sb) {
                const uint32_t pa = t.x * kTile + sa, pb = t.y * kTile + sb;
                if (!(pa < pb && pb < n_seq)) continue;
                const uint32_t a = order[pa], b = order[pb];
                const uint64_t n = offsets[a + 1] - offsets[a], m = offsets[b + 1] - offsets[b];
                if (n == 0 || m == 0) return APD_ERR_EMPTY_SEQUENCE;
                const uint64_t mx = std::max(n, m), gap = mx - std::min(n, m);
                uint64_t bnd = host_band_from_pct(band.pct, (uint32_t)mx);
                const uint64_t w = std::max(bnd, gap) + 2;                // alignments.rs:173 (unclamped: same cell set)
                np += 2;                                                  // both ordered pairs (alignments.rs:50-51)
                nc += band_cells(n, m, w) + band_cells(m, n, w);
                nb += 2 * (4ull * dim * (n + m) + 4);
            }
    if (pairs) *pairs = np;
    if (cells) *cells = nc;
    if (alg_bytes) *alg_bytes = nb;
    return APD_OK;
}

// ------------------------------------------------------------------------------- device buffers

extern "C" int apd_device_alloc(apd_context *ctx, uint64_t bytes, void **d_ptr)
{
    if (!ctx || !d_ptr) return APD_ERR_INVALID_ARG;
    *d_ptr = nullptr;
    HIP_TRY(ctx, bind_device(ctx));
    HIP_TRY(ctx, hipMalloc(d_ptr, std::max<size_t>((size_t)bytes, 16)));
    ctx->buffers.insert(*d_ptr);
    return APD_OK;
}

extern "C" int apd_device_free(apd_context *ctx, void *d_ptr)
{
    if (!ctx) return APD_ERR_INVALID_ARG;
    if (!d_ptr) return APD_OK;
    HIP_TRY(ctx, bind_device(ctx));
    if (ctx->buffers.erase(d_ptr) == 0) return APD_ERR_INVALID_ARG;       // not a buffer of this context (or freed twice)
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));                      // nothing queued may still use it
    HIP_TRY(ctx, hipFree(d_ptr));
    return APD_OK;
}

extern "C" int apd_copy_to_device(apd_context *ctx, void *d_dst, const void *src, uint64_t bytes)
{
    if (!ctx || (bytes && (!d_dst || !src))) return APD_ERR_INVALID_ARG;
    if (bytes == 0) return APD_OK;
    HIP_TRY(ctx, bind_device(ctx));
    HIP_TRY(ctx, hipMemcpyAsync(d_dst, src, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return APD_OK;
}

extern "C" int apd_copy_to_host(apd_context *ctx, void *dst, const void *d_src, uint64_t bytes)
{
    if (!ctx || (bytes && (!dst || !d_src))) return APD_ERR_INVALID_ARG;
    if (bytes == 0) return APD_OK;
    HIP_TRY(ctx, bind_device(ctx));
    HIP_TRY(ctx, hipMemcpyAsync(dst, d_src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return APD_OK;
}

extern "C" int apd_device_fill(apd_context *ctx, void *d_dst, int byte_value, uint64_t bytes)
{
    if (!ctx || (bytes && !d_dst)) return APD_ERR_INVALID_ARG;
    if (bytes == 0) return APD_OK;
    HIP_TRY(ctx, bind_device(ctx));
    HIP_TRY(ctx, hipMemsetAsync(d_dst, byte_value, (size_t)bytes, ctx->stream));
    return APD_OK;
}

// Multi-GPU alignment behind the C ABI: RCCL communicators owned by the library, ONE all-gather of the equal-sized pair-tile
// slabs over xGMI (include/apd.h, "the same over several GPUs").  Replaces AlignmentWorkers::align_all's thread fan-out
// (reference src/alignments.rs:31-67: `alignment_workers` threads over row blocks, joined at :64-66) for N GPUs.
// No torch, no MPI: the 128-byte unique id travels over whatever channel the host has.
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "apd_internal.h"

static_assert(sizeof(ncclUniqueId) == APD_COMM_ID_BYTES, "apd.h promises 128-byte communicator ids");

struct apd_comm {
    ncclComm_t comm = nullptr;
    apd_context *ctx = nullptr;       // the context (device, stream) this end of the communicator belongs to
    uint32_t rank = 0, world = 1;
};

namespace {

#define HIP_TRY(ctx, call)                                                             \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e_);     \
            return e_ == hipErrorOutOfMemory ? APD_ERR_OOM : APD_ERR_HIP;              \
        }                                                                              \
    } while (0)

#define NCCL_TRY(ctx, call)                                                            \
    do {                                                                               \
        ncclResult_t r_ = (call);                                                      \
        if (r_ != ncclSuccess) {                                                       \
            if (ctx) (ctx)->last_error = std::string(#call) + ": " + ncclGetErrorString(r_); \
            return APD_ERR_COMM;                                                       \
        }                                                                              \
    } while (0)

int ensure_gather(apd_context *ctx, size_t need)
{
    if (ctx->ws_gather && ctx->ws_gather_bytes >= need) return APD_OK;
    if (ctx->ws_gather) { HIP_TRY(ctx, hipFree(ctx->ws_gather)); ctx->ws_gather = nullptr; ctx->ws_gather_bytes = 0; }
    HIP_TRY(ctx, hipMalloc(&ctx->ws_gather, need));
    ctx->ws_gather_bytes = need;
    return APD_OK;
}

}  // namespace

extern "C" int apd_comm_unique_id(void *id_bytes)
{
    if (!id_bytes) return APD_ERR_INVALID_ARG;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return APD_ERR_COMM;
    std::memcpy(id_bytes, &id, sizeof(id));
    return APD_OK;
}

extern "C" int apd_comm_create(apd_context *ctx, const void *id_bytes, uint32_t rank, uint32_t world, apd_comm **out)
{
    if (!ctx || !id_bytes || !out || world == 0 || rank >= world) return APD_ERR_INVALID_ARG;
    *out = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    apd_comm *c = new (std::nothrow) apd_comm();
    if (!c) return APD_ERR_OOM;
    c->ctx = ctx; c->rank = rank; c->world = world;
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof(id));
    const ncclResult_t r = ncclCommInitRank(&c->comm, (int)world, id, (int)rank);   // collective over all `world` ranks
    if (r != ncclSuccess) {
        ctx->last_error = std::string("ncclCommInitRank: ") + ncclGetErrorString(r);
        delete c;
        return APD_ERR_COMM;
    }
    ctx->comms.insert(c);
    *out = c;
    return APD_OK;
}

namespace apd {
void orphan_comm(apd_comm *c)                                            // the context is going away (apd_destroy)
{
    if (c->comm) ncclCommDestroy(c->comm);
    c->comm = nullptr;
    c->ctx = nullptr;
}
}  // namespace apd

extern "C" int apd_comm_destroy(apd_comm *c)
{
    if (!c) return APD_ERR_INVALID_ARG;
    if (c->ctx) {                                                        // else: orphaned by apd_destroy, RCCL side already gone
        hipSetDevice(c->ctx->device);
        hipStreamSynchronize(c->ctx->stream);
        if (c->comm) ncclCommDestroy(c->comm);
        c->ctx->comms.erase(c);
    }
    delete c;
    return APD_OK;
}

extern "C" int apd_comm_count(const apd_comm *c, uint32_t *world)
{
    if (!c || !world || !c->comm) return APD_ERR_INVALID_ARG;
    int n = 0;
    NCCL_TRY(c->ctx, ncclCommCount(c->comm, &n));
    *world = (uint32_t)n;
    return APD_OK;
}

extern "C" int apd_comm_rank(const apd_comm *c, uint32_t *rank)
{
    if (!c || !rank || !c->comm) return APD_ERR_INVALID_ARG;
    int r = 0;
    NCCL_TRY(c->ctx, ncclCommUserRank(c->comm, &r));
    *rank = (uint32_t)r;
    return APD_OK;
}

extern "C" int apd_all_gather_async(apd_context *ctx, apd_comm *c, const float *d_send, float *d_recv, uint64_t count)
{
    if (!ctx || !c || c->ctx != ctx || (count && (!d_send || !d_recv))) return APD_ERR_INVALID_ARG;
    if (count == 0) return APD_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    NCCL_TRY(ctx, ncclAllGather(d_send, d_recv, (size_t)count, ncclFloat, c->comm, ctx->stream));
    return APD_OK;
}

// This rank's pair tiles -> its slab, in place inside the gather buffer (sendbuff == recvbuff + rank * count is RCCL's
// in-place all-gather) -> ONE ncclAllGather on the context's stream -> unpack into the n x n matrix.
extern "C" int apd_align_all_sharded_async(apd_context *ctx, apd_comm *c, const apd_batch *batch, const apd_align_config *cfg,
                                           float *d_out)
{
    if (!ctx || !batch || !cfg || (!d_out && batch->n_seq)) return APD_ERR_INVALID_ARG;
    if (!c) return apd_align_all_device_async(ctx, batch, cfg, d_out);
    if (c->ctx != ctx) return APD_ERR_INVALID_ARG;
    if (batch->n_seq == 0) return APD_OK;
    const uint64_t slab = apd_slab_floats(batch->n_seq, c->world);
    int rc = ensure_gather(ctx, std::max<size_t>((size_t)slab * c->world * sizeof(float), 16));
    if (rc) return rc;
    float *gathered = (float *)ctx->ws_gather, *mine = gathered + (size_t)slab * c->rank;
    rc = apd_align_tiles_async(ctx, batch, cfg, c->rank, c->world, mine);
    if (rc) return rc;
    NCCL_TRY(ctx, ncclAllGather(mine, gathered, (size_t)slab, ncclFloat, c->comm, ctx->stream));
    return apd_unpack_tiles_async(ctx, batch, c->world, gathered, d_out);
}

// One process, n_devices GPUs: ncclCommInitAll, one context and one resident copy of the batch per device, every device
// aligns its share of the pair tiles on its own stream, one grouped all-gather, unpack on devices[0].
extern "C" int apd_align_all_multi(const int *devices, uint32_t n_devices, const float *frames, const uint64_t *offsets, uint32_t n_seq,
                                   uint32_t dim, const apd_align_config *cfg, float *out, uint32_t *ranks_seen)
{
    if (!devices || n_devices == 0 || !offsets || !cfg || (n_seq && !out)) return APD_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < n_devices; ++i)
        for (uint32_t j = 0; j < i; ++j)
            if (devices[i] == devices[j]) return APD_ERR_INVALID_ARG;        // RCCL refuses two ranks on one device
    std::vector<apd_context *> ctx(n_devices, nullptr);
    std::vector<apd_batch *> batch(n_devices, nullptr);
    std::vector<ncclComm_t> comms(n_devices, nullptr);
    float *d_out = nullptr;
    bool comms_live = false;
    int rc = APD_OK;
    auto cleanup = [&]() {
        for (uint32_t i = 0; i < n_devices; ++i) if (ctx[i]) { hipSetDevice(ctx[i]->device); hipStreamSynchronize(ctx[i]->stream); }
        if (comms_live) for (uint32_t i = 0; i < n_devices; ++i) if (comms[i]) ncclCommDestroy(comms[i]);
        if (d_out) { hipSetDevice(ctx[0]->device); hipFree(d_out); }
        for (uint32_t i = 0; i < n_devices; ++i) {
            if (batch[i]) apd_batch_destroy(batch[i]);
            if (ctx[i]) apd_destroy(ctx[i]);
        }
    };
    for (uint32_t i = 0; i < n_devices && rc == APD_OK; ++i) rc = apd_create(devices[i], &ctx[i]);
    if (rc == APD_OK && ncclCommInitAll(comms.data(), (int)n_devices, devices) != ncclSuccess) rc = APD_ERR_COMM;
    else if (rc == APD_OK) comms_live = true;
    if (rc == APD_OK && ranks_seen) {
        int cnt = 0;
        if (ncclCommCount(comms[0], &cnt) != ncclSuccess) rc = APD_ERR_COMM;
        *ranks_seen = (uint32_t)cnt;
    }
    // every device holds the whole corpus (<= 2.1 GB at cfg 5): any pair tile can be aligned anywhere.  One host thread per
    // device uploads the batch and enqueues that device's pair tiles: the uploads run over eight PCIe links at once instead of
    // one after the other (cfg 3: 218 MB per device, a fifth of a device's share of the alignment time each).
    const uint64_t slab = n_seq ? apd_slab_floats(n_seq, n_devices) : 0;
    const size_t gather_bytes = std::max<size_t>((size_t)slab * n_devices * sizeof(float), 16);
    if (rc == APD_OK) {
        std::vector<int> drc(n_devices, APD_OK);
        auto per_device = [&](uint32_t i) {
            int r = apd_batch_create(ctx[i], frames, offsets, n_seq, dim, 0, &batch[i]);
            if (r == APD_OK && n_seq > 0) {
                r = hipSetDevice(ctx[i]->device) == hipSuccess ? APD_OK : APD_ERR_HIP;
                if (r == APD_OK) r = ensure_gather(ctx[i], gather_bytes);
                if (r == APD_OK) r = apd_align_tiles_async(ctx[i], batch[i], cfg, i, n_devices, (float *)ctx[i]->ws_gather + (size_t)slab * i);
            }
            drc[i] = r;
        };
        std::vector<std::thread> workers;
        for (uint32_t i = 1; i < n_devices; ++i) workers.emplace_back(per_device, i);
        per_device(0);
        for (std::thread &t : workers) t.join();
        for (uint32_t i = 0; i < n_devices; ++i)
            if (drc[i] != APD_OK && rc == APD_OK) { rc = drc[i]; if (i != 0) ctx[0]->last_error = "device " + std::to_string(devices[i]) + ": " + ctx[i]->last_error; }
    }
    if (rc == APD_OK && n_seq > 0) {
        if (rc == APD_OK) {
            ncclResult_t r = ncclGroupStart();
            for (uint32_t i = 0; i < n_devices && r == ncclSuccess; ++i) {
                hipSetDevice(ctx[i]->device);
                float *g = (float *)ctx[i]->ws_gather;
                r = ncclAllGather(g + (size_t)slab * i, g, (size_t)slab, ncclFloat, comms[i], ctx[i]->stream);
            }
            const ncclResult_t r2 = ncclGroupEnd();
            if (r != ncclSuccess || r2 != ncclSuccess) {
                ctx[0]->last_error = std::string("ncclAllGather: ") + ncclGetErrorString(r != ncclSuccess ? r : r2);
                rc = APD_ERR_COMM;
            }
        }
        if (rc == APD_OK) {
            hipSetDevice(ctx[0]->device);
            const size_t bytes = (size_t)n_seq * n_seq * sizeof(float);
            if (hipMalloc((void **)&d_out, bytes) != hipSuccess) rc = APD_ERR_OOM;
            if (rc == APD_OK) rc = apd_unpack_tiles_async(ctx[0], batch[0], n_devices, (const float *)ctx[0]->ws_gather, d_out);
            if (rc == APD_OK && hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, ctx[0]->stream) != hipSuccess) rc = APD_ERR_HIP;
            if (rc == APD_OK) rc = apd_synchronize(ctx[0]);               // also reports APD_ERR_INCOMPLETE
        }
        for (uint32_t i = 1; i < n_devices; ++i) { const int r = apd_synchronize(ctx[i]); if (rc == APD_OK) rc = r; }
    }
    cleanup();
    return rc;
}

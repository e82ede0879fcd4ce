// Multi-GPU alignment behind the C ABI: RCCL communicators owned by the library, ONE all-gather of the equal-sized pair-tile
// slabs over xGMI (include/apd.h, "the same over several GPUs").  Replaces AlignmentWorkers::align_all's thread fan-out
// (reference src/alignments.rs:31-67: `alignment_workers` threads over row blocks, joined at :64-66) for N GPUs.
// No torch, no MPI: the 128-byte unique id travels over whatever channel the host has.
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <set>
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
    APD_AFFINITY(ctx, "gather workspace allocation");
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
    HIP_TRY(ctx, apd::bind_device(ctx));
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
        apd::bind_device(c->ctx);
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
    HIP_TRY(ctx, apd::bind_device(ctx));
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

// ---------------------------------------------------------------------------------------------------------------------
// One process, n_devices GPUs, as a PERSISTENT handle (include/apd.h "as a persistent handle"): contexts, communicators,
// worker threads, gather workspaces and resident batches are made once; an align_all pays kernels + one all-gather + unpack.

namespace {

// One host thread per device, alive as long as the handle: the counterpart of the reference's `alignment_workers` threads
// (alignments.rs:35-41).  run(f) executes f(i) on worker i for every device at once and returns when all are done.
class WorkerPool {
public:
    explicit WorkerPool(uint32_t n) : slots_(n > 1 ? n : 0)
    {
        try {
            for (uint32_t i = 0; i < slots_.size(); ++i) threads_.emplace_back([this, i] { loop(i); });
        } catch (...) {                                                   // a thread that cannot be started (EAGAIN, a process limit):
            stop();                                                       // the ones already running are joined, not destroyed joinable
            throw;
        }
    }
    ~WorkerPool() { stop(); }
    void run(uint32_t n, const std::function<int(uint32_t)> &f, std::vector<int> &rc)
    {
        rc.assign(n, APD_OK);
        if (slots_.empty()) {                                             // one device: no thread at all
            for (uint32_t i = 0; i < n; ++i) rc[i] = f(i);
            return;
        }
        {
            std::lock_guard<std::mutex> g(mu_);
            task_ = &f; rc_ = &rc; pending_ = (uint32_t)slots_.size(); ++generation_;
        }
        cv_work_.notify_all();
        std::unique_lock<std::mutex> g(mu_);
        cv_done_.wait(g, [this] { return pending_ == 0; });
        task_ = nullptr; rc_ = nullptr;
    }

private:
    void stop()
    {
        {
            std::lock_guard<std::mutex> g(mu_);
            quit_ = true;
        }
        cv_work_.notify_all();
        for (std::thread &t : threads_) if (t.joinable()) t.join();
        threads_.clear();
    }
    void loop(uint32_t i)
    {
        uint64_t seen = 0;
        for (;;) {
            const std::function<int(uint32_t)> *f;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_work_.wait(g, [&] { return quit_ || generation_ != seen; });
                if (quit_) return;
                seen = generation_;
                f = task_;
            }
            int r;
            try { r = (*f)(i); } catch (...) { r = APD_ERR_OOM; }          // nothing may unwind out of a worker
            {
                std::lock_guard<std::mutex> g(mu_);
                (*rc_)[i] = r;
                if (--pending_ == 0) cv_done_.notify_all();
            }
        }
    }
    std::vector<int> slots_;
    std::vector<std::thread> threads_;
    std::mutex mu_;
    std::condition_variable cv_work_, cv_done_;
    const std::function<int(uint32_t)> *task_ = nullptr;
    std::vector<int> *rc_ = nullptr;
    uint32_t pending_ = 0;
    uint64_t generation_ = 0;
    bool quit_ = false;
};

}  // namespace

struct apd_multi {
    std::vector<int> devices;
    std::vector<apd_context *> ctx;
    std::vector<ncclComm_t> comms;                 // empty in the peer-copy fallback
    std::vector<hipEvent_t> slab_ready;            // peer-copy fallback: device i's slab is complete (recorded on its stream)
    hipEvent_t gathered = nullptr;                 //                     devices[0] has copied every slab of the last call
    bool gathered_valid = false;
    std::string collective, last_error;
    std::set<apd_multi_batch *> batches;
    float *d_result = nullptr;                     // matrix owned by the handle (apd_multi_align_all_async with d_out == NULL)
    size_t result_bytes = 0;
    const float *last_result = nullptr;
    WorkerPool *pool = nullptr;
};

struct apd_multi_batch {
    apd_multi *multi = nullptr;
    std::vector<apd_batch *> per_device;
    uint32_t n_seq = 0, dim = 0;
};

namespace {

uint32_t nd(const apd_multi *m) { return (uint32_t)m->devices.size(); }

// rc[] of a fan-out -> one status; the first failing device's text goes to the handle
int merge_status(apd_multi *m, const std::vector<int> &rc)
{
    for (uint32_t i = 0; i < rc.size(); ++i)
        if (rc[i] != APD_OK) {
            m->last_error = "device " + std::to_string(m->devices[i]) + ": " + apd_status_string(rc[i]) +
                            (m->ctx[i]->last_error.empty() ? "" : " (" + m->ctx[i]->last_error + ")");
            return rc[i];
        }
    return APD_OK;
}

int multi_fail(apd_multi *m, int rc, const std::string &what)
{
    m->last_error = what;
    return rc;
}

}  // namespace

static int apd_multi_create_impl(const int *devices, uint32_t n_devices, apd_multi **out)
{
    if (!devices || n_devices == 0 || !out) return APD_ERR_INVALID_ARG;
    *out = nullptr;
    const char *force = std::getenv("APD_MULTI_COLLECTIVE");
    const bool force_peer = force && std::strcmp(force, "peer") == 0;
    // RCCL refuses two ranks on one device.  The peer-copy collective does not care: with it forced, a device may be named more
    // than once -- several "ranks" (contexts, streams, worker threads, tile shares, slabs) on one GPU.  That is how the N > 1
    // machinery of this handle is rehearsed on a box with a single GPU (tests/test_gpu_multi.py); it is of no use in production.
    for (uint32_t i = 0; i < n_devices && !force_peer; ++i)
        for (uint32_t j = 0; j < i; ++j)
            if (devices[i] == devices[j]) return APD_ERR_INVALID_ARG;
    apd_multi *m = new (std::nothrow) apd_multi();
    if (!m) return APD_ERR_OOM;
    m->devices.assign(devices, devices + n_devices);
    m->ctx.assign(n_devices, nullptr);
    int rc = APD_OK;
    for (uint32_t i = 0; i < n_devices && rc == APD_OK; ++i) rc = apd_create(devices[i], &m->ctx[i]);
    if (rc != APD_OK) { apd_multi_destroy(m); return rc; }
    std::string why;
    if (force_peer) why = "APD_MULTI_COLLECTIVE=peer";
    else {
        m->comms.assign(n_devices, nullptr);
        const ncclResult_t r = ncclCommInitAll(m->comms.data(), (int)n_devices, devices);
        if (r != ncclSuccess) {
            why = std::string("ncclCommInitAll: ") + ncclGetErrorString(r);
            m->comms.clear();
        }
    }
    if (!m->comms.empty()) {
        int version = 0;
        ncclGetVersion(&version);
        m->collective = "rccl: ncclCommInitAll over " + std::to_string(n_devices) + " device(s), one grouped ncclAllGather per align_all (RCCL " +
                        std::to_string(version) + ")";
    } else {
        m->collective = "peer-copy fallback: " + why + "; slabs gathered onto devices[0] with hipMemcpyPeerAsync";
        m->slab_ready.assign(n_devices, nullptr);
        for (uint32_t i = 0; i < n_devices; ++i) {
            apd::bind_device(m->ctx[i]);
            if (hipEventCreateWithFlags(&m->slab_ready[i], hipEventDisableTiming) != hipSuccess) rc = APD_ERR_HIP;
        }
        apd::bind_device(m->ctx[0]);
        if (hipEventCreateWithFlags(&m->gathered, hipEventDisableTiming) != hipSuccess) rc = APD_ERR_HIP;
        if (rc != APD_OK) { apd_multi_destroy(m); return rc; }
    }
    try {
        m->pool = new WorkerPool(n_devices);
    } catch (const std::bad_alloc &) {
        apd_multi_destroy(m);                                             // contexts, communicators, events: nothing leaks
        return APD_ERR_OOM;
    } catch (...) {                                                       // std::system_error: a worker thread could not be started
        apd_multi_destroy(m);
        return APD_ERR_HIP;
    }
    *out = m;
    return APD_OK;
}

extern "C" int apd_multi_destroy(apd_multi *m)
{
    if (!m) return APD_ERR_INVALID_ARG;
    for (apd_context *c : m->ctx) if (c) { apd::bind_device(c); hipStreamSynchronize(c->stream); }
    delete m->pool;                                                       // joins the workers
    m->pool = nullptr;
    while (!m->batches.empty()) apd_multi_batch_destroy(*m->batches.begin());
    for (ncclComm_t c : m->comms) if (c) ncclCommDestroy(c);
    for (uint32_t i = 0; i < m->slab_ready.size(); ++i) if (m->slab_ready[i] && m->ctx[i]) { apd::bind_device(m->ctx[i]); hipEventDestroy(m->slab_ready[i]); }
    if (m->gathered && m->ctx[0]) { apd::bind_device(m->ctx[0]); hipEventDestroy(m->gathered); }
    if (m->d_result && m->ctx[0]) { apd::bind_device(m->ctx[0]); hipFree(m->d_result); }
    for (apd_context *c : m->ctx) if (c) apd_destroy(c);
    delete m;
    return APD_OK;
}

extern "C" uint32_t apd_multi_size(const apd_multi *m) { return m ? nd(m) : 0; }

extern "C" int apd_multi_ranks_seen(const apd_multi *m, uint32_t *ranks)
{
    if (!m || !ranks) return APD_ERR_INVALID_ARG;
    if (m->comms.empty()) { *ranks = nd(m); return APD_OK; }
    int cnt = 0;
    if (ncclCommCount(m->comms[0], &cnt) != ncclSuccess) return APD_ERR_COMM;
    *ranks = (uint32_t)cnt;
    return APD_OK;
}

extern "C" const char *apd_multi_collective(const apd_multi *m) { return m ? m->collective.c_str() : "null handle"; }
extern "C" const char *apd_multi_last_error(const apd_multi *m) { return m ? m->last_error.c_str() : "null handle"; }
extern "C" apd_context *apd_multi_context(apd_multi *m, uint32_t i) { return (m && i < nd(m)) ? m->ctx[i] : nullptr; }
extern "C" const float *apd_multi_result(const apd_multi *m) { return m ? m->last_result : nullptr; }

static int apd_multi_batch_create_impl(apd_multi *m, const float *frames, const float *const *d_frames, const uint64_t *offsets,
                                      uint32_t n_seq, uint32_t dim, apd_multi_batch **out)
{
    if (!m || !offsets || !out) return APD_ERR_INVALID_ARG;
    *out = nullptr;
    if (offsets[n_seq] > 0 && !frames && !d_frames) return APD_ERR_INVALID_ARG;
    apd_multi_batch *mb = new (std::nothrow) apd_multi_batch();
    if (!mb) return APD_ERR_OOM;
    mb->multi = m; mb->n_seq = n_seq; mb->dim = dim;
    mb->per_device.assign(nd(m), nullptr);
    // every device holds the whole corpus (<= 2.1 GB at cfg 5): any pair tile can be aligned anywhere.  The workers upload
    // concurrently, one PCIe link each.
    std::vector<int> rc;
    m->pool->run(nd(m), [&](uint32_t i) {
        return apd_batch_create(m->ctx[i], d_frames ? d_frames[i] : frames, offsets, n_seq, dim, d_frames ? 1 : 0, &mb->per_device[i]);
    }, rc);
    const int st = merge_status(m, rc);
    m->batches.insert(mb);
    if (st != APD_OK) { apd_multi_batch_destroy(mb); return st; }
    *out = mb;
    return APD_OK;
}

static int apd_multi_batch_refill_impl(apd_multi *m, apd_multi_batch *mb, const float *frames, const float *const *d_frames)
{
    if (!m || !mb || mb->multi != m) return APD_ERR_INVALID_ARG;
    if (!frames && !d_frames && mb->n_seq) return APD_ERR_INVALID_ARG;
    std::vector<int> rc;
    m->pool->run(nd(m), [&](uint32_t i) {
        return apd_batch_refill(m->ctx[i], mb->per_device[i], d_frames ? d_frames[i] : frames, d_frames ? 1 : 0);
    }, rc);
    return merge_status(m, rc);
}

extern "C" int apd_multi_batch_destroy(apd_multi_batch *mb)
{
    if (!mb) return APD_ERR_INVALID_ARG;
    for (apd_batch *b : mb->per_device) if (b) apd_batch_destroy(b);
    if (mb->multi) mb->multi->batches.erase(mb);
    delete mb;
    return APD_OK;
}

static int apd_multi_align_all_async_impl(apd_multi *m, const apd_multi_batch *mb, const apd_align_config *cfg, float *d_out)
{
    if (!m || !mb || !cfg || mb->multi != m) return APD_ERR_INVALID_ARG;
    const uint32_t n = nd(m), n_seq = mb->n_seq;
    if (n_seq == 0) return APD_OK;
    const uint64_t slab = apd_slab_floats(n_seq, n);
    const size_t gather_bytes = std::max<size_t>((size_t)slab * n * sizeof(float), 16);
    if (!d_out) {                                                          // the handle's own result matrix on devices[0]
        const size_t need = (size_t)n_seq * n_seq * sizeof(float);
        if (m->result_bytes < need) {
            apd::bind_device(m->ctx[0]);
            if (m->d_result) { hipStreamSynchronize(m->ctx[0]->stream); hipFree(m->d_result); m->d_result = nullptr; m->result_bytes = 0; }
            if (hipMalloc((void **)&m->d_result, need) != hipSuccess) return multi_fail(m, APD_ERR_OOM, "device " + std::to_string(m->devices[0]) + ": result matrix");
            m->result_bytes = need;
        }
        d_out = m->d_result;
    }
    const bool peer = m->comms.empty();
    if (peer && m->gathered_valid) {
        // A larger batch than the last one makes ensure_gather free and re-allocate a device's gather buffer -- while devices[0]'s
        // stream may still be copying the previous slab out of it (hipFree drains only the owning device's streams).  Wait for that
        // gather on the host first; the common case (same size again) never gets here.
        bool grows = false;
        for (uint32_t i = 0; i < n; ++i) grows |= m->ctx[i]->ws_gather_bytes < gather_bytes;
        if (grows) {
            apd::bind_device(m->ctx[0]);
            if (hipEventSynchronize(m->gathered) != hipSuccess) return multi_fail(m, APD_ERR_HIP, "hipEventSynchronize(gathered)");
        }
    }
    // 1. every device: its pair tiles into its slab, in place inside its gather buffer
    std::vector<int> rc;
    m->pool->run(n, [&](uint32_t i) -> int {
        apd_context *c = m->ctx[i];
        if (apd::bind_device(c) != hipSuccess) return (int)APD_ERR_HIP;      // worker threads have their own current device
        int r = ensure_gather(c, gather_bytes);
        if (r != APD_OK) return r;
        // peer fallback: devices[0] must have copied this device's previous slab before it is poisoned again
        if (peer && i != 0 && m->gathered_valid && hipStreamWaitEvent(c->stream, m->gathered, 0) != hipSuccess) return (int)APD_ERR_HIP;
        r = apd_align_tiles_async(c, mb->per_device[i], cfg, i, n, (float *)c->ws_gather + (size_t)slab * i);
        if (r == APD_OK && peer) {
            APD_AFFINITY(c, "slab_ready event record");
            if (hipEventRecord(m->slab_ready[i], c->stream) != hipSuccess) r = APD_ERR_HIP;
        }
        return r;
    }, rc);
    int st = merge_status(m, rc);
    if (st != APD_OK) return st;
    // 2. ONE all-gather of the equal-sized slabs (in place: sendbuff = recvbuff + i * slab), grouped over the devices
    if (!peer) {
        ncclResult_t r = ncclGroupStart();
        for (uint32_t i = 0; i < n && r == ncclSuccess; ++i) {
            float *g = (float *)m->ctx[i]->ws_gather;
            apd::bind_device(m->ctx[i]);
            r = ncclAllGather(g + (size_t)slab * i, g, (size_t)slab, ncclFloat, m->comms[i], m->ctx[i]->stream);
        }
        const ncclResult_t r2 = ncclGroupEnd();
        if (r != ncclSuccess || r2 != ncclSuccess)
            return multi_fail(m, APD_ERR_COMM, std::string("ncclAllGather: ") + ncclGetErrorString(r != ncclSuccess ? r : r2));
    } else {
        apd_context *c0 = m->ctx[0];
        apd::bind_device(c0);
        float *g0 = (float *)c0->ws_gather;
        for (uint32_t i = 1; i < n; ++i) {
            const float *gi = (const float *)m->ctx[i]->ws_gather + (size_t)slab * i;
            if (hipStreamWaitEvent(c0->stream, m->slab_ready[i], 0) != hipSuccess ||
                hipMemcpyPeerAsync(g0 + (size_t)slab * i, c0->device, gi, m->ctx[i]->device, (size_t)slab * sizeof(float), c0->stream) != hipSuccess)
                return multi_fail(m, APD_ERR_HIP, "device " + std::to_string(m->devices[i]) + ": peer copy of the slab failed");
        }
        if (hipEventRecord(m->gathered, c0->stream) != hipSuccess) return multi_fail(m, APD_ERR_HIP, "hipEventRecord(gathered)");
        m->gathered_valid = true;
    }
    // 3. unpack on devices[0] (rank 0 is the consumer: UPGMA runs there)
    st = apd_unpack_tiles_async(m->ctx[0], mb->per_device[0], n, (const float *)m->ctx[0]->ws_gather, d_out);
    if (st != APD_OK) { rc.assign(n, APD_OK); rc[0] = st; return merge_status(m, rc); }
    m->last_result = d_out;
    return APD_OK;
}

static int apd_multi_synchronize_impl(apd_multi *m)
{
    if (!m) return APD_ERR_INVALID_ARG;
    std::vector<int> rc(nd(m), APD_OK);
    for (uint32_t i = 0; i < nd(m); ++i) rc[i] = apd_synchronize(m->ctx[i]);   // also reports APD_ERR_INCOMPLETE (devices[0] unpacks)
    return merge_status(m, rc);
}

static int apd_multi_align_all_impl(apd_multi *m, const apd_multi_batch *mb, const apd_align_config *cfg, float *out)
{
    if (!m || !mb || !cfg || mb->multi != m || (!out && mb->n_seq)) return APD_ERR_INVALID_ARG;
    if (mb->n_seq == 0) return APD_OK;
    int rc = apd_multi_align_all_async(m, mb, cfg, nullptr);
    if (rc != APD_OK) { apd_multi_synchronize(m); return rc; }
    apd_context *c0 = m->ctx[0];
    apd::bind_device(c0);
    if (hipMemcpyAsync(out, m->last_result, (size_t)mb->n_seq * mb->n_seq * sizeof(float), hipMemcpyDeviceToHost, c0->stream) != hipSuccess)
        rc = multi_fail(m, APD_ERR_HIP, "device " + std::to_string(c0->device) + ": copy of the matrix to the host failed");
    const int rs = apd_multi_synchronize(m);
    return rc != APD_OK ? rc : rs;
}

// One-shot form: everything above made and torn down inside one call (kept for callers that align once).
static int apd_align_all_multi_impl(const int *devices, uint32_t n_devices, const float *frames, const uint64_t *offsets, uint32_t n_seq,
                                   uint32_t dim, const apd_align_config *cfg, float *out, uint32_t *ranks_seen)
{
    if (!devices || n_devices == 0 || !offsets || !cfg || (n_seq && !out)) return APD_ERR_INVALID_ARG;
    apd_multi *m = nullptr;
    int rc = apd_multi_create(devices, n_devices, &m);
    if (rc != APD_OK) return rc;
    if (ranks_seen) rc = apd_multi_ranks_seen(m, ranks_seen);
    apd_multi_batch *mb = nullptr;
    if (rc == APD_OK) rc = apd_multi_batch_create(m, frames, nullptr, offsets, n_seq, dim, &mb);
    if (rc == APD_OK) rc = apd_multi_align_all(m, mb, cfg, out);
    if (rc != APD_OK) std::fprintf(stderr, "[apd] apd_align_all_multi: %s\n", apd_multi_last_error(m));   // the handle (and its text) dies here
    apd_multi_destroy(m);
    return rc;
}


// ---- the C boundary of the entry points above: nothing unwinds across it (std::bad_alloc, std::system_error of a thread
// that cannot be started, ...) ------------------------------------------------------------------------------------------
#define APD_GUARDED(call)                                              \
    try { return call; }                                              \
    catch (const std::bad_alloc &) { return APD_ERR_OOM; }            \
    catch (...) { return APD_ERR_HIP; }
extern "C" int apd_multi_create(const int *devices, uint32_t n_devices, apd_multi **out) { APD_GUARDED(apd_multi_create_impl(devices, n_devices, out)) }
extern "C" int apd_multi_batch_create(apd_multi *m, const float *frames, const float *const *d_frames, const uint64_t *offsets, uint32_t n_seq, uint32_t dim, apd_multi_batch **out) { APD_GUARDED(apd_multi_batch_create_impl(m, frames, d_frames, offsets, n_seq, dim, out)) }
extern "C" int apd_multi_batch_refill(apd_multi *m, apd_multi_batch *mb, const float *frames, const float *const *d_frames) { APD_GUARDED(apd_multi_batch_refill_impl(m, mb, frames, d_frames)) }
extern "C" int apd_multi_align_all_async(apd_multi *m, const apd_multi_batch *mb, const apd_align_config *cfg, float *d_out) { APD_GUARDED(apd_multi_align_all_async_impl(m, mb, cfg, d_out)) }
extern "C" int apd_multi_synchronize(apd_multi *m) { APD_GUARDED(apd_multi_synchronize_impl(m)) }
extern "C" int apd_multi_align_all(apd_multi *m, const apd_multi_batch *mb, const apd_align_config *cfg, float *out) { APD_GUARDED(apd_multi_align_all_impl(m, mb, cfg, out)) }
extern "C" int apd_align_all_multi(const int *devices, uint32_t n_devices, const float *frames, const uint64_t *offsets, uint32_t n_seq, uint32_t dim, const apd_align_config *cfg, float *out, uint32_t *ranks_seen) { APD_GUARDED(apd_align_all_multi_impl(devices, n_devices, frames, offsets, n_seq, dim, cfg, out, ranks_seen)) }

// ---- SURVEY.md section 8(b)'s one-call forms (alignments.rs:17-67 + main.rs:187-195; clustering.rs:81-110)
static int apd_dtw_all_pairs_impl(const float *frames, const uint64_t *offsets, uint32_t n_seq, uint32_t dim, float band_pct, float ins_pen,
                                  float del_pen, float match_pen, int n_devices, float *out)
{
    std::vector<int> devices((size_t)std::max(n_devices, 1));
    for (size_t i = 0; i < devices.size(); ++i) devices[i] = (int)i;
    const apd_align_config cfg{band_pct, ins_pen, del_pen, match_pen};
    return apd_align_all_multi_impl(devices.data(), (uint32_t)devices.size(), frames, offsets, n_seq, dim, &cfg, out, nullptr);
}
extern "C" int apd_dtw_all_pairs(const float *frames, const uint64_t *offsets, uint32_t n_seq, uint32_t dim, float band_pct, float ins_pen,
                                 float del_pen, float match_pen, int n_devices, float *out)
{ APD_GUARDED(apd_dtw_all_pairs_impl(frames, offsets, n_seq, dim, band_pct, ins_pen, del_pen, match_pen, n_devices, out)) }

static int apd_upgma_impl(const float *dist, uint32_t n, float perc, apd_cluster_op *ops, uint32_t *n_ops, uint32_t *roots, uint32_t *n_roots)
{
    apd_context *ctx = nullptr;
    int rc = apd_create(0, &ctx);
    if (rc != APD_OK) return rc;
    rc = apd_clustering(ctx, dist, 0, n, perc, ops, n_ops, roots, n_roots, nullptr);
    apd_destroy(ctx);
    return rc;
}
extern "C" int apd_upgma(const float *dist, uint32_t n, float perc, apd_cluster_op *ops, uint32_t *n_ops, uint32_t *roots, uint32_t *n_roots)
{ APD_GUARDED(apd_upgma_impl(dist, n, perc, ops, n_ops, roots, n_roots)) }

// ------------------------------------------------------------------------------------------------------ runtime identity
extern "C" uint64_t apd_runtime_info(char *out, uint64_t capacity)
{
    auto path_of = [](const void *sym) -> std::string {
        Dl_info info;
        return (dladdr(sym, &info) && info.dli_fname) ? info.dli_fname : "?";
    };
    int hip_version = 0, nccl_version = 0;
    hipRuntimeGetVersion(&hip_version);
    ncclGetVersion(&nccl_version);
    const std::string text = "hip " + std::to_string(hip_version) + " @ " + path_of((const void *)&hipStreamSynchronize) + "; rccl " +
                             std::to_string(nccl_version) + " @ " + path_of((const void *)&ncclAllGather);
    if (out && capacity) {
        const size_t k = std::min<size_t>(text.size(), (size_t)capacity - 1);
        std::memcpy(out, text.data(), k);
        out[k] = '\0';
    }
    return text.size() + 1;
}

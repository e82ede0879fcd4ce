// Internal declarations shared by the HIP translation units of libapd_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "../../include/apd.h"

struct apd_context;
namespace apd {

constexpr int kTile = 16;              // sequences per tile side -> 256 pair slots per tile
constexpr int kSlotsPerTile = kTile * kTile;
constexpr int kWave = 64;

// How the per-pair Sakoe-Chiba band is obtained.
struct BandSpec {
    float pct;                 // Discovery.warping_band_percentage (discovery.rs:40)
    uint32_t explicit_band;    // AlignmentParams.warping_band when use_explicit
    int use_explicit;
    float ins, del, mat;
};

// One launch of the fused-pair alignment over a list of tiles.
struct AlignLaunch {
    const float *d_frames;       // resident layout, see dtw_generic.hip: [frames | sentinel H | sentinel E] per sequence, dpad floats per frame
    uint32_t frames_bytes;       // size of d_frames in bytes (0 if >= 4 GiB: buffer addressing unavailable)
    const uint32_t *d_seq_off;   // [n_seq+1] padded frame offsets (sequence s owns seq_off[s+1]-seq_off[s]-2 real frames)
    const float *d_seq_nmax;     // [n_seq] largest squared frame norm of every resident sequence (bound of the hybrid form's threshold test)
    const uint4 *d_tiles;        // [n_tiles] (tile_a, tile_b, index of the tile in the slab, unused), tile_a <= tile_b
    uint32_t n_tiles;
    uint32_t n_seq;
    uint32_t dim, dpad;
    BandSpec band;
    float *d_slab;               // [tiles of the rank][2][kTile][kTile]
    // Device-side choice between the fast kernels and the literal, NaN-faithful one (no host round trip): the word the repack
    // kernel raises when a frame holds a NaN / infinity.  Fast kernels return at once when it is set; the fallback launch of
    // the generic kernel (launch_generic_fallback) returns at once when it is clear.  nullptr: no check.
    const uint32_t *d_nonfinite;
    uint32_t w_max;              // upper bound of w over the pairs of this launch
    uint32_t n_max;              // upper bound of the longer length over the pairs of this launch
    int variant;                 // 0 auto
    int hybrid;                  // 1: norm-expansion distances with exact recomputation below tau (see dtw_systolic.h)
    int strict;                  // 1: the reference's arithmetic operation for operation, whatever the penalties (apd_set_distance_mode 2)
    float tau;
};

// Frame dimensions with instantiated kernels.  A batch of another dimension <= 26 is zero-padded up to the next one when it
// is made resident (zero components change neither (x - y)^2 sums nor norms and dot products, bit for bit).
constexpr int kKernelDims[] = {8, 10, 13, 16, 20, 26};
constexpr bool is_kernel_dim(uint32_t d) { for (int k : kKernelDims) if ((uint32_t)k == d) return true; return false; }
constexpr uint32_t kernel_dim(uint32_t d) { for (int k : kKernelDims) if ((uint32_t)k >= d) return (uint32_t)k; return d; }
// Cells per lane the register file holds: (C + 2) frames of ceil4(D + 1) floats plus ~50 registers of state.
constexpr int max_cells_per_lane(uint32_t d) { return d <= 13 ? 9 : (d <= 16 ? 7 : 5); }
// Column strips of the full-matrix kernel hold CW frames per lane and one DP row: wider strips fit.
constexpr int max_strip_columns(uint32_t d) { return d <= 10 ? 13 : (d <= 13 ? 11 : max_cells_per_lane(d)); }   // 13 x 14 floats do not fit at D = 13

// geom_key = G * 100 + C of the systolic kernel, or 0 for the generic kernel (see pick_geometry_key)
hipError_t launch_align(const AlignLaunch &L, int geom_key, hipStream_t stream, std::string &err, int *status);
// The generic kernel over ALL tiles of L as a small persistent grid that does nothing unless *L.d_nonfinite is set.
// *fits = false (and nothing launched) if the band of L.w_max needs more LDS than a workgroup can have.
hipError_t launch_generic_fallback(const AlignLaunch &L, hipStream_t stream, bool *fits);
// true if the literal kernel can hold a band of w_max in LDS AND the runtime grants it that much (the hipFuncSetAttribute is made
// here, before anything is enqueued: a refusal sends the caller down the host-side choice instead of failing mid-call)
bool generic_fallback_fits(uint32_t w_max);
int pick_geometry_key(uint32_t need, uint32_t dim, int variant, bool uniform_pen, bool fast_shift);
// >= 20000: full-matrix kernel, 20000 + (pairs per wavefront) * 100 + CW, for pairs of at most `rows` x `cols` frames (0 if it does not apply)
int pick_full_key(uint32_t cols, uint32_t rows, uint32_t dim, int variant);   // (>= 10000: wide kernel, 10000 + NW * 100 + C)
double full_key_cost(uint32_t cols, uint32_t rows, uint32_t dim, int key);    // modelled cost of one pair on that geometry (+inf: does not apply)
// >= 30000: the same column strips with a binding band (two DPs): 30000 + (pairs per wavefront) * 100 + CW
int pick_banded_strip_key(uint32_t cols, uint32_t rows, uint32_t dim, int variant);
// d_flags[0] is raised when a frame holds a NaN or an infinity (the fast kernels' selects and sentinels assume finite features)
// d_seq_nmax[p] (zeroed by the caller) receives the largest squared frame norm of resident sequence p
hipError_t launch_pad(const float *d_src, float *d_dst, const uint32_t *d_seq_off, const uint32_t *d_src_off, uint32_t n_seq,
                      uint64_t n_frames_padded, uint32_t src_dim, uint32_t dim, uint32_t dpad, uint32_t *d_flags, float *d_seq_nmax,
                      hipStream_t stream);
// writes EVERY entry of d_out (diagonal included); *d_status |= 1 when a pair score of a batch with finite frames is NaN,
// i.e. still carries the poison its slab was filled with before the alignment launches
hipError_t launch_unpack(const float *d_gathered, float *d_out, const uint32_t *d_order, uint32_t n_seq, uint32_t world,
                         uint64_t slab_floats, const uint32_t *d_flags, uint32_t *d_status, hipStream_t stream);
hipError_t launch_selftest(int *d_result, hipStream_t stream);
hipError_t launch_sqrt_sweep(uint32_t first, uint64_t count, unsigned long long *d_out, hipStream_t stream);
// Features of magnitude >= 2^60 (or NaN / infinite) raise the batch's flag: below it every squared distance and every frame
// norm is finite (26 * (2 * 2^60)^2 < 2^127), which the fast kernels' square roots and norm expansion rely on.
constexpr float kFeatureBound = 0x1p60f;

// comm.hip: called by apd_destroy for every communicator still alive on the context
void orphan_comm(apd_comm *comm);
// companions.hip: likewise for the resident feature objects
void orphan_encoder(apd_encoder *enc);
void orphan_cepstrum_plan(apd_cepstrum_plan *plan);

// numerics.rs:125-133 on a device array (clustering.hip): radix select of the k-th smallest non-NaN value.
int device_select(apd_context *ctx, const float *d_x, uint64_t len, uint64_t k, float *value);
uint64_t percentile_index(uint64_t len, float perc);

// Host-side mirrors of the device band arithmetic (bit-identical f32 product).
inline uint32_t host_band_from_pct(float pct, uint32_t len)
{
    float p = pct * (float)len;                 // discovery.rs:40
    if (!(p > 0.0f)) return 0;                  // NaN / negative saturate to 0 (Rust `as usize`)
    if (p >= 4294967040.0f) return 0xFFFFFFFFu;
    return (uint32_t)p;
}
inline uint32_t host_w(const BandSpec &b, uint32_t n, uint32_t m)
{
    uint32_t mx = n > m ? n : m, gap = n > m ? n - m : m - n;
    uint32_t band = b.use_explicit ? b.explicit_band : host_band_from_pct(b.pct, mx);
    if (band > mx) band = mx;                   // any band >= max(n,m) already covers the full matrix
    return (band > gap ? band : gap) + 2;       // alignments.rs:173
}

// ---- device affinity.  Every entry point binds its context's device to the calling thread through bind_device() before it
// allocates, records an event or launches.  With APD_DEBUG_AFFINITY=1 the library also remembers, per thread, WHICH CONTEXT was
// bound last, and APD_AFFINITY(ctx) -- placed at the allocations, event records and launches inside the library -- fails the call
// unless that context is `ctx` and hipGetDevice() agrees.  Comparing contexts, not device numbers, is what lets the one-GPU
// rehearsals (several "ranks" that all name device 0, tests/test_gpu_multi.py) catch a worker thread that forgot to bind.
extern thread_local const apd_context *tl_bound_context;
bool affinity_debug();
hipError_t bind_device(const apd_context *ctx);
bool affinity_ok(const apd_context *ctx, const char *where);

}  // namespace apd

#define APD_AFFINITY(ctx, where)                                                                   \
    do { if (apd::affinity_debug() && !apd::affinity_ok((ctx), (where))) return APD_ERR_HIP; } while (0)

struct apd_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool timing = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // side streams for the launches of different tile classes (independent work, joined back into `stream` with events)
    static constexpr int kSideStreams = 4;
    hipStream_t side[kSideStreams] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t side_done[kSideStreams] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr;
    // batches created on this context and not yet destroyed: apd_destroy releases their device memory and orphans them, so
    // that a later apd_batch_destroy (destruction order is the caller's, e.g. a garbage collector's) only frees the host part
    std::set<apd_batch *> batches;
    // communicators made on this context and not yet destroyed: apd_destroy tears their RCCL side down and orphans them, so
    // that a later apd_comm_destroy only frees the host part (same contract as for batches)
    std::set<apd_comm *> comms;
    std::set<apd_encoder *> encoders;                 // resident feature objects (companions.hip), same contract
    std::set<apd_cepstrum_plan *> cepstrum_plans;
    // buffers handed out by apd_device_alloc and not yet freed: apd_destroy releases them (a host whose destructors run in any
    // order -- a garbage collector's -- may free a buffer after its context: apd_device_free on a destroyed context is never
    // called by the mirrors, and nothing leaks)
    std::set<void *> buffers;
    bool timed = false;
    int variant = 0;
    int distance_mode = 1;            // 0 exact differences, 1 hybrid, 2 strict (bit-identical to the CPU arithmetic)
    float tau = 1.0f / 64.0f;
    std::string last_error;
    // reusable device workspaces
    void *ws_tiles = nullptr; size_t ws_tiles_bytes = 0;
    void *ws_slab = nullptr; size_t ws_slab_bytes = 0;
    void *ws_misc = nullptr; size_t ws_misc_bytes = 0;
    void *ws_gather = nullptr; size_t ws_gather_bytes = 0;   // gathered slabs of apd_align_all_sharded_async
    uint32_t *d_status = nullptr;     // sticky device word: bit 0 = an unpack met an unwritten (poisoned) pair score
    uint32_t drop_tiles = 0;          // fault injection (apd_set_fault_injection)
    apd_batch *pair_batch = nullptr;  // apd_align_pair: the last pair's two-sequence batch, refilled while (n, m, dim) repeat
    uint64_t pair_n = 0, pair_m = 0;
    uint32_t pair_dim = 0;
};

struct apd_batch {
    apd_context *ctx = nullptr;
    uint32_t n_seq = 0, dim = 0, dpad = 0;   // dim: resident (kernel) frame dimension, >= src_dim
    uint32_t src_dim = 0;                    // the caller's frame dimension
    uint64_t total_frames = 0;
    float *d_frames = nullptr;        // padded layout with sentinels (see dtw_generic.hip)
    uint32_t frames_bytes = 0;
    uint32_t *d_meta = nullptr;       // ONE allocation: [seq_off n+1 | src_off n+1 | order n+1 | flags 4 | nmax n+1], filled by one copy of h_meta
    std::vector<uint32_t> h_meta;     // host image of d_meta, alive as long as the batch (the H2D copy is asynchronous)
    uint32_t *d_seq_off = nullptr;
    uint32_t *d_src_off = nullptr;    // first frame of resident sequence p in the caller's frame array
    uint32_t *d_order = nullptr;      // resident position p -> caller's sequence index
    uint32_t *d_flags = nullptr;      // [0] 1: some frame holds a NaN / infinity (raised by the repack kernel)
    float *d_seq_nmax = nullptr;      // [n_seq] largest squared frame norm per resident sequence (written by the repack kernel)
    mutable int nonfinite = -1;       // host copy of d_flags[0]; -1: not read back yet
    std::vector<uint32_t> order;      // host copy of d_order
    std::vector<uint64_t> offsets;    // frame offsets of the RESIDENT order (host)
    uint32_t min_len = 0, max_len = 0;
    // device-resident tile lists, grouped by the kernel geometry each tile needs
    struct TileClass { int geom_key; uint32_t first, count, w_max, n_max; };
    struct TilePlan { uint4 *d_tiles = nullptr; std::vector<TileClass> classes; };
    mutable std::map<std::string, TilePlan> tile_cache;   // keyed by rank/world/band/variant
};

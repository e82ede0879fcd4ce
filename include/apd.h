/*
 * apd.h -- C ABI of the MI355X-native alignment + clustering path.
 *
 * Drop-in boundary for the one hot path of dkohlsdorf/audio_pattern_discovery:
 * AlignmentWorkers::align_all (src/alignments.rs:31-67) feeding
 * AgglomerativeClustering::clustering (src/clustering.rs:81-110), plus the two feature
 * companions NDSequence::new (src/spectrogram.rs:31-94) and AutoEncoder::predict
 * (src/neural.rs:55-71).  The reference has no FFI of its own (one Rust crate, plain `pub`
 * items called from src/main.rs:187-203); each entry point below names the Rust item a
 * binding crate would route to it -- INTEGRATION.md shows that binding.
 *
 * Conventions: plain pointers and sizes only; every function returns an apd_status
 * (0 = OK, negative = error), never throws or aborts across the boundary; the caller owns
 * host buffers, the library owns device buffers behind the opaque handles; functions taking
 * a context are synchronous unless their name ends in _async.  Pointers named d_* are
 * DEVICE pointers (HBM of the context's GPU), everything else is host memory.
 *
 * There is no CPU fallback: without a gfx950 device apd_create fails with
 * APD_ERR_NO_DEVICE and nothing else can be called.
 */
#ifndef APD_H
#define APD_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum apd_status {
    APD_OK = 0,
    APD_ERR_INVALID_ARG = -1,
    APD_ERR_NO_DEVICE = -2,      /* no HIP device / not gfx950 */
    APD_ERR_HIP = -3,            /* a HIP runtime call failed: apd_last_error() has the text */
    APD_ERR_OOM = -4,
    APD_ERR_EMPTY_SEQUENCE = -5, /* a zero-length sequence: the reference underflows usize at alignments.rs:120 */
    APD_ERR_BAND_TOO_WIDE = -6,  /* 2*w+1 exceeds what one wavefront's LDS state can hold */
    APD_ERR_INDEX = -7,          /* percentile index past the data: the reference panics at numerics.rs:132 */
    APD_ERR_UNSUPPORTED = -8,
    APD_ERR_INCOMPLETE = -9,     /* a pair score was never written (a launch was cut short or skipped): the output holds NaN
                                    there.  The reference swallows a failed worker and leaves silent zeros
                                    (alignments.rs:64-66); this library poisons and reports instead */
    APD_ERR_COMM = -10           /* an RCCL call failed: apd_last_error() has the text */
} apd_status;

typedef struct apd_context apd_context;   /* one GPU + one HIP stream + workspaces */
typedef struct apd_batch apd_batch;       /* Arc<Vec<NDSequence>> resident in HBM (alignments.rs:12) */
typedef struct apd_comm apd_comm;         /* this rank's end of an RCCL communicator over the GPUs that share the pair tiles */
typedef struct apd_encoder apd_encoder;   /* AutoEncoder's w_encode / b_encode resident on a context's GPU (neural.rs:13-17) */
typedef struct apd_cepstrum_plan apd_cepstrum_plan;   /* window, filterbank, DCT and twiddle tables + the offsets of a corpus, resident */

/* The four Discovery fields the path reads (src/discovery.rs:17-20, project/config/Discovery.toml:17-20). */
typedef struct apd_align_config {
    float warping_band_percentage;
    float insertion_penalty;
    float deletion_penalty;
    float match_penalty;
} apd_align_config;

/* AlignmentParams (src/alignments.rs:77-83). */
typedef struct apd_alignment_params {
    uint64_t warping_band;
    float insertion_penalty;
    float deletion_penalty;
    float match_penalty;
} apd_alignment_params;

/* Merge (src/clustering.rs:8-13) and ClusteringOperation (src/clustering.rs:19-25). */
enum { APD_SEQUENCE2SEQUENCE = 0, APD_SEQUENCE2CLUSTER = 1, APD_CLUSTER2SEQUENCE = 2, APD_CLUSTER2CLUSTER = 3 };
typedef struct apd_cluster_op {
    uint32_t merge_i;
    uint32_t merge_j;
    uint32_t into;
    float distance;
    uint32_t operation;
} apd_cluster_op;

/* ---- context ------------------------------------------------------------------------- */
/* Environment: APD_DEBUG_AFFINITY=1 makes every allocation, event record and launch inside the library check that the calling
 * thread is bound to THIS context (not merely to the same device number) and fail with APD_ERR_HIP otherwise -- how the
 * several-ranks-on-one-GPU rehearsals catch a worker thread that forgot hipSetDevice (tests/test_gpu_multi.py). */
int apd_create(int device, apd_context **ctx);
/* Also releases the device memory of every batch, and the RCCL side of every communicator, still alive on the context; such
 * a batch / communicator may (and must, for its host part) still be passed to apd_batch_destroy / apd_comm_destroy
 * afterwards, in any order, and is refused by every other call. */
int apd_destroy(apd_context *ctx);
/* Run on the caller's hipStream_t (e.g. torch's current stream) instead of the context's own. */
int apd_set_stream(apd_context *ctx, void *hip_stream);
/* Waits for the context's stream.  Also reports (once) what asynchronous calls found since the last report:
 * APD_ERR_INCOMPLETE if an unpack met a pair score that no kernel wrote. */
int apd_synchronize(apd_context *ctx);
const char *apd_status_string(int status);
const char *apd_last_error(apd_context *ctx);
/* Record HIP events around every alignment kernel launch; apd_last_kernel_ms returns the
 * duration of the most recent one (ms, on the launch stream), < 0 if none was timed. */
int apd_set_timing(apd_context *ctx, int enabled);
float apd_last_kernel_ms(apd_context *ctx);
/* Tuning knob for experiments: 0 = pick automatically.  See DESIGN.md "Kernel variants". */
int apd_set_variant(apd_context *ctx, int variant);
/* Local-distance form of the fast kernels with UNIT penalties (1.0, 1.0, 1.0 -- the shipped Discovery.toml).
 * mode 1 (default; D >= 8 in the band kernels, D >= 10 in the strip kernels): |x|^2 + |y|^2 - 2 x.y from precomputed frame
 * norms, recomputed in the difference form wherever the result is below tau * (|x|^2 + |y|^2) (cancellation region; tau <= 0
 * keeps the current value, default 1/64): ~3e-7 relative measured (tolerance asked: 1e-4); exact copies score exactly 0.
 * mode 0: the difference form sqrt(sum (x_k-y_k)^2).  In the band-form kernels it is computed operation for operation as
 * numerics.rs:114-120 does (every difference, square and partial sum rounded on its own, correctly rounded sqrt), and with
 * unit penalties the fast select picks the reference's predecessor for every non-NaN input: scores are bit-identical to the CPU
 * code, about 1.9x the time of mode 1.  (The strip kernels of full-band batches keep an fma chain in mode 0: ~2e-7.)
 * mode 2 (strict): the reference's arithmetic in EVERY kernel family -- band kernels as mode 0, strip kernels through their
 * literal-select path: every score bit-identical to the CPU code -- the mode that meets 1e-4 on EVERY entry.  cfg 3: 1.44 s instead
 * of 0.75 s (the square root is v_sqrt_f32 + an exact two-sided fix-up, checked for every f32 input by apd_selftest_sqrt).
 * With any other penalties the recurrence is discontinuous in its inputs (the penalty added depends on which predecessor wins
 * a strict comparison), so the library ignores the mode and computes operation for operation as numerics.rs:114-120 /
 * alignments.rs:129-160 do: bit-identical to the CPU arithmetic.
 * What modes 0 / 2 buy over mode 1: the reference resolves an EXACT tie between the DELETE and INSERT predecessors by taking
 * MATCH even when MATCH is larger; when such a tie arises by coincidence of two rounded f32 sums (real-valued features),
 * mode 1 -- whose distances differ in the last bit -- does not see a tie and keeps the smaller predecessor, and that entry can
 * differ by a few 1e-4 relative.  Counted over every entry of the BASELINE shapes (tests/test_gpu_census.py): none on cfg 2,
 * cfg 3 and cfg 5's shape, one pair of 8.4 million on cfg 4; over 60 corpora per shape 1-6 entries in 1e8, worst seen 2.2e-3
 * (tools/census_sweep.py).  Ties that are structural (identical frames, integer features,
 * +INF) are reproduced in every mode. */
int apd_set_distance_mode(apd_context *ctx, int mode, float tau);
/* Device self-test of the cross-lane primitives the kernels rely on (DPP wave shifts). */
int apd_selftest(apd_context *ctx);
/* TEST HOOK: strict mode's square root (v_sqrt_f32 + an exact two-sided fix-up, csrc/dtw_common.h sqrt_rn_finite) against the
 * compiler's correctly rounded sqrtf (= Rust's f32::sqrt, numerics.rs:119) for the `count` (<= 2^32) consecutive f32 bit patterns
 * from first_bits, restricted to its domain 2^-96 <= x < +INF: *mismatches = how many differ (must be 0), *first_mismatch the
 * first such pattern, raw_ulp_hist[5] (may be NULL) = how often the bare v_sqrt_f32 is off by <= -2, -1, 0, +1, >= +2 ulps.
 * The whole f32 range takes about a second (tests/test_gpu_sqrt.py). */
int apd_selftest_sqrt(apd_context *ctx, uint32_t first_bits, uint64_t count, uint64_t *mismatches, uint32_t *first_mismatch,
                      uint64_t *raw_ulp_hist);
/* TEST HOOK: binds the calling thread to `bound`, then runs the library's affinity check for `checked` (see APD_DEBUG_AFFINITY
 * above): APD_OK if the check is off (*enabled = 0) or the two are the same context, APD_ERR_HIP with the text in
 * apd_last_error(checked) otherwise -- also when both contexts sit on the same device, which is the point. */
int apd_debug_affinity_probe(apd_context *bound, apd_context *checked, int *enabled);
/* TEST HOOK: *busy = 1 while work enqueued on the context's stream has not finished (hipStreamQuery), 0 once it is idle.  What the
 * "_async only enqueues" tests assert on, instead of wall-clock ratios. */
int apd_stream_busy(apd_context *ctx, int *busy);
/* TEST HOOK (fault injection): the next alignment launches leave the last `drop_tiles` tiles of every kernel class
 * unprocessed, as a launch that is cut short would.  0 = off.  Exists so that the poison / APD_ERR_INCOMPLETE path can
 * be tested (tests/test_gpu_dtw.py); never set it in production. */
int apd_set_fault_injection(apd_context *ctx, uint32_t drop_tiles);

/* ---- Discovery::alignment_params (src/discovery.rs:38-45) ----------------------------- */
int apd_discovery_alignment_params(const apd_align_config *cfg, uint64_t n_size, apd_alignment_params *out);

/* ---- AlignmentWorkers::new (src/alignments.rs:17-26) ---------------------------------- */
/* frames: packed [offsets[n_seq]][dim] f32 row-major (NDSequence.frames of every sequence,
 * spectrogram.rs:16-17, back to back); offsets: n_seq+1 frame offsets.  If frames_on_device
 * != 0, `frames` is a device pointer and stays owned by the caller (it is only read during
 * this call).  The batch keeps its own HBM copy in the kernels' padded layout. */
int apd_batch_create(apd_context *ctx, const float *frames, const uint64_t *offsets, uint32_t n_seq,
                     uint32_t dim, int frames_on_device, apd_batch **batch);
int apd_batch_destroy(apd_batch *batch);
uint32_t apd_batch_len(const apd_batch *batch);
/* New frame VALUES for the same sequence lengths (the same `offsets` and `dim` the batch was created with): re-runs only
 * the repack kernel, asynchronously on the context's stream; the batch keeps its device buffers and its cached tile
 * plans.  For pipelines that call align_all repeatedly on features recomputed in HBM (bench.py's step). */
int apd_batch_refill(apd_context *ctx, apd_batch *batch, const float *frames, int frames_on_device);
/* 1 if the resident frames hold a NaN or an infinity (then every pair goes through the literal, NaN-faithful kernel:
 * NaN compares false and takes the MATCH branch, alignments.rs:153-159), 0 if all are finite.  Synchronises. */
int apd_batch_nonfinite(apd_context *ctx, const apd_batch *batch, int *nonfinite);

/* ---- AlignmentWorkers::align_all (src/alignments.rs:31-67) ---------------------------- */
/* out: n_seq*n_seq f32 row-major, out[i*n+j] = Alignment::score of (x = seq i, y = seq j),
 * diagonal 0.0 (alignments.rs:21-23,51,57).  Blocking. */
int apd_align_all(apd_context *ctx, const apd_batch *batch, const apd_align_config *cfg, float *out);
/* Same, result left in HBM (d_out: n_seq*n_seq floats); asynchronous on the context's stream. */
int apd_align_all_device_async(apd_context *ctx, const apd_batch *batch, const apd_align_config *cfg,
                               float *d_out);

/* Sharded form (the reference's static row blocks, alignments.rs:33-37, become pair tiles):
 * the sequences are taken in LENGTH ORDER (apd_length_order: longest first, equal lengths by
 * ascending index -- a batch keeps them resident in that order), and the upper triangle of the
 * n x n pair matrix over those positions is cut into apd_tile_size() x apd_tile_size() tiles, so
 * that the sequences of a tile have like lengths; rank r of `world` owns tiles r, r+world, ...
 * Each rank fills one packed slab of apd_slab_floats() floats; after an all-gather of the `world`
 * slabs (rank order) into d_gathered, apd_unpack_tiles_async scatters them into the n x n matrix
 * indexed by the caller's sequence numbers. */
uint32_t apd_tile_size(void);
uint64_t apd_num_tiles(uint32_t n_seq);
uint64_t apd_rank_tiles(uint32_t n_seq, uint32_t rank, uint32_t world);
uint64_t apd_slab_floats(uint32_t n_seq, uint32_t world);
/* "_async" for the alignment entry points (this one, apd_align_all_device_async, apd_align_all_sharded_async,
 * apd_multi_align_all_async) means what it says: the kernels, the collective and the unpack are only ENQUEUED when the call
 * returns, also right after apd_batch_create / apd_batch_refill.  The choice between the fast kernels and the literal,
 * NaN-faithful one (a batch with a NaN / infinite feature must take the latter) is made ON THE DEVICE: the repack kernel
 * leaves its verdict in a flag word, the fast kernels return at once when it is raised, and a small persistent launch of the
 * literal kernel behind them returns at once when it is not.  One exception: a band so wide that the literal kernel cannot
 * hold it in LDS (2w+1 > 20 480 offsets) -- then the first alignment after a fill reads the flag back on the host (4 bytes,
 * one synchronisation of the context's stream).  The first call on a new batch / band also builds the tile plan on the host
 * (a small upload and a stream synchronisation); later calls reuse it. */
int apd_align_tiles_async(apd_context *ctx, const apd_batch *batch, const apd_align_config *cfg,
                          uint32_t rank, uint32_t world, float *d_slab);
int apd_unpack_tiles_async(apd_context *ctx, const apd_batch *batch, uint32_t world, const float *d_gathered,
                           float *d_out);
/* ---- the same over several GPUs, collective included (RCCL over xGMI; no torch, no MPI) --------------------------
 * The reference runs `alignment_workers` threads over row blocks (alignments.rs:33-41); here a worker is a GPU.  Two ways
 * to bring the GPUs together:
 *
 * (1) one PROCESS PER GPU (how bench.py is launched): rank 0 calls apd_comm_unique_id and hands the 128 bytes to every
 *     rank over whatever channel the host has (a file, a socket, MPI, torch.distributed's store); every rank then calls
 *     apd_comm_create on its own context.  apd_align_all_sharded_async = this rank's tiles (apd_align_tiles_async) +
 *     ONE ncclAllGather of the equal-sized slabs on the context's stream + apd_unpack_tiles_async: afterwards EVERY
 *     rank holds the full n x n matrix in d_out (rank 0 is the consumer: UPGMA runs there).
 * (2) ONE process driving n_devices GPUs: apd_align_all_multi (ncclCommInitAll, one context + one resident copy of the
 *     batch per device, per-device streams, the same all-gather inside one ncclGroup, unpack on devices[0], copy to the
 *     host).  n_devices == 1 degenerates to apd_align_all and returns the identical bits. */
#define APD_COMM_ID_BYTES 128
int apd_comm_unique_id(void *id_bytes /* APD_COMM_ID_BYTES */);
int apd_comm_create(apd_context *ctx, const void *id_bytes, uint32_t rank, uint32_t world, apd_comm **comm);
int apd_comm_destroy(apd_comm *comm);
int apd_comm_count(const apd_comm *comm, uint32_t *world);   /* ncclCommCount: the ranks RCCL actually sees */
int apd_comm_rank(const apd_comm *comm, uint32_t *rank);     /* ncclCommUserRank */
/* d_out: n_seq*n_seq floats in this rank's HBM; the batch must hold the same sequences on every rank.  Asynchronous on
 * the context's stream.  With comm == NULL the call is apd_align_all_device_async (world 1). */
int apd_align_all_sharded_async(apd_context *ctx, apd_comm *comm, const apd_batch *batch, const apd_align_config *cfg,
                                float *d_out);
/* The same with the slab gathered by the caller's own transport between the two halves is apd_align_tiles_async +
 * apd_unpack_tiles_async below.  apd_all_gather_async is the library's collective on its own: d_send (count floats) of
 * every rank, concatenated in rank order into d_recv (count * world floats), on the context's stream. */
int apd_all_gather_async(apd_context *ctx, apd_comm *comm, const float *d_send, float *d_recv, uint64_t count);
/* frames / offsets / out: host memory, as apd_batch_create + apd_align_all take them.  devices: HIP device ordinals.
 * Blocking.  ONE-SHOT convenience over the persistent handle below (apd_multi_create + apd_multi_batch_create +
 * apd_multi_align_all + destroy): it pays communicator setup on every call -- a host that aligns more than once
 * (main.rs:187-195 in a loop) keeps an apd_multi instead. */
int apd_align_all_multi(const int *devices, uint32_t n_devices, const float *frames, const uint64_t *offsets, uint32_t n_seq,
                        uint32_t dim, const apd_align_config *cfg, float *out, uint32_t *ranks_seen);

/* The two one-call entry points SURVEY.md section 8(b) spells out for the Rust shim and the C++ harness -- the whole alignment leg
 * (AlignmentWorkers::new + align_all, src/alignments.rs:17-67, main.rs:187-195) and the whole clustering leg
 * (AgglomerativeClustering::clustering, src/clustering.rs:81-110) with host buffers in and out, the library owning contexts,
 * device buffers and communicators for the duration of the call.  Synchronous, thread-compatible, 0 = OK, nothing unwinds.
 * apd_dtw_all_pairs: frames [sum len][dim], offsets [n_seq + 1] in frames, out [n_seq][n_seq] caller-owned; devices 0 ..
 *   n_devices - 1 (n_devices <= 1: device 0 alone); default distance mode.  = apd_align_all_multi over those devices.
 * apd_upgma: dist [n][n] host, ops capacity >= n, roots capacity >= n; on device 0.  = apd_create + apd_clustering + apd_destroy.
 * A host that calls them more than once keeps an apd_multi / apd_context instead and saves the set-up. */
int apd_dtw_all_pairs(const float *frames, const uint64_t *offsets, uint32_t n_seq, uint32_t dim, float band_pct,
                      float ins_pen, float del_pen, float match_pen, int n_devices, float *out);
int apd_upgma(const float *dist, uint32_t n, float perc, apd_cluster_op *ops, uint32_t *n_ops, uint32_t *roots, uint32_t *n_roots);

/* ---- (2) as a persistent handle: AlignmentWorkers { data, result } (alignments.rs:11-14) over n_devices GPUs ---------
 * apd_multi owns, for its whole life: one apd_context per device (own stream), the RCCL communicators of
 * ncclCommInitAll, one host worker thread per device (the reference's `alignment_workers` threads, alignments.rs:35-41:
 * here a worker feeds a GPU), the gather workspaces and -- per apd_multi_batch -- one resident copy of the corpus and the
 * cached tile plans on every device.  Create once, align many: a second apd_multi_align_all pays kernels + one
 * all-gather + unpack only.
 * If RCCL cannot make the communicators (or APD_MULTI_COLLECTIVE=peer is set), the handle falls back to gathering the slabs
 * onto devices[0] with hipMemcpyPeerAsync; apd_multi_collective() says which and carries RCCL's error text.  Results are
 * the same bits either way (the collective only moves the slabs).  With the peer-copy collective forced a device may be named
 * several times in `devices` (several ranks on one GPU): a rehearsal aid for single-GPU boxes, nothing else. */
/* A handle is driven by ONE host thread at a time (its own worker threads are internal); different handles and contexts
 * may be used from different threads concurrently. */
typedef struct apd_multi apd_multi;
typedef struct apd_multi_batch apd_multi_batch;
int apd_multi_create(const int *devices, uint32_t n_devices, apd_multi **multi);
int apd_multi_destroy(apd_multi *multi);                     /* also destroys the batches still alive on it */
uint32_t apd_multi_size(const apd_multi *multi);             /* n_devices */
int apd_multi_ranks_seen(const apd_multi *multi, uint32_t *ranks);   /* ncclCommCount (n_devices in the peer-copy fallback) */
const char *apd_multi_collective(const apd_multi *multi);    /* "rccl: ..." or "peer-copy fallback: <why>" */
const char *apd_multi_last_error(const apd_multi *multi);    /* "device k: <text>" of the last failing call */
/* Device i's context, owned by the handle (never apd_destroy it): for the per-device settings (apd_set_distance_mode,
 * apd_set_timing, apd_set_variant), the feature kernels (apd_encode, apd_cepstrum_batch) and buffers on that device. */
apd_context *apd_multi_context(apd_multi *multi, uint32_t i);
/* AlignmentWorkers::new (alignments.rs:17-26) on every device.  Either `frames` (host, packed as for apd_batch_create,
 * uploaded to every device by its worker thread) or `d_frames` (n_devices device pointers, d_frames[i] in device i's
 * HBM, only read during the call) must be given. */
int apd_multi_batch_create(apd_multi *multi, const float *frames, const float *const *d_frames, const uint64_t *offsets,
                           uint32_t n_seq, uint32_t dim, apd_multi_batch **batch);
int apd_multi_batch_refill(apd_multi *multi, apd_multi_batch *batch, const float *frames, const float *const *d_frames);
int apd_multi_batch_destroy(apd_multi_batch *batch);
/* AlignmentWorkers::align_all (alignments.rs:31-67): every device aligns its pair tiles (g = i mod n_devices), ONE grouped
 * ncclAllGather, unpack on devices[0].  d_out: n_seq*n_seq floats in devices[0]'s HBM (NULL: a buffer owned by the handle,
 * see apd_multi_result).  Returns when every device's work is ENQUEUED; apd_multi_synchronize waits for all devices and
 * reports APD_ERR_INCOMPLETE like apd_synchronize. */
int apd_multi_align_all_async(apd_multi *multi, const apd_multi_batch *batch, const apd_align_config *cfg, float *d_out);
int apd_multi_synchronize(apd_multi *multi);
/* Device address (devices[0]) of the matrix the last apd_multi_align_all_async(.., d_out = NULL) wrote; NULL if none. */
const float *apd_multi_result(const apd_multi *multi);
/* Blocking form: out = n_seq*n_seq floats, host. */
int apd_multi_align_all(apd_multi *multi, const apd_multi_batch *batch, const apd_align_config *cfg, float *out);

/* ---- device buffers -------------------------------------------------------------------------------------------------
 * For hosts that keep features or the matrix resident between calls (what a Rust binding wraps in a Drop-ing DeviceVec):
 * plain hipMalloc / hipFree / hipMemcpy on the context's device, the copies ordered on the context's stream and
 * complete when the call returns.  apd_destroy releases the buffers of a context that were never freed (after which they
 * must not be passed to apd_device_free); freeing a pointer the context did not hand out is APD_ERR_INVALID_ARG. */
int apd_device_alloc(apd_context *ctx, uint64_t bytes, void **d_ptr);
int apd_device_free(apd_context *ctx, void *d_ptr);
int apd_copy_to_device(apd_context *ctx, void *d_dst, const void *src, uint64_t bytes);
int apd_copy_to_host(apd_context *ctx, void *dst, const void *d_src, uint64_t bytes);
int apd_device_fill(apd_context *ctx, void *d_dst, int byte_value, uint64_t bytes);   /* hipMemsetAsync on the stream */
/* Which HIP and RCCL libraries this process actually mapped for the library's calls (dladdr of hipMalloc /
 * ncclAllGather) and their versions, as one line of text; returns the length needed (incl. NUL). */
uint64_t apd_runtime_info(char *out, uint64_t capacity);

/* Host-side views of the same sharding (no GPU needed): the length order (order[p] = sequence at position p), the
 * (tile_a, tile_b) list of a rank over those positions, 2 uint32 per tile, and the scatter of gathered slabs held in
 * HOST memory (for a host that gathers over its own transport). */
int apd_length_order(const uint64_t *offsets, uint32_t n_seq, uint32_t *order);
int apd_rank_tile_list(uint32_t n_seq, uint32_t rank, uint32_t world, uint32_t *tile_ab, uint64_t capacity,
                       uint64_t *n_tiles);
int apd_unpack_tiles_host(const uint64_t *offsets, uint32_t n_seq, uint32_t world, const float *gathered, float *out);

/* Work accounting for the metric: cells = sum over ordered pairs of the
 * cells alignments.rs:174-175 visits; alg_bytes = sum of 4*dim*(n+m)+4. */
int apd_align_work(const uint64_t *offsets, uint32_t n_seq, uint32_t dim, const apd_align_config *cfg,
                   uint32_t rank, uint32_t world, uint64_t *pairs, uint64_t *cells, uint64_t *alg_bytes);

/* ---- Alignment::new + construct_alignment + score (src/alignments.rs:107-180) --------- */
/* x: [n][dim], y: [m][dim] host.  *score = Alignment::score() after construct_alignment(x, y, params). */
int apd_align_pair(apd_context *ctx, const float *x, uint64_t n, const float *y, uint64_t m, uint32_t dim,
                   const apd_alignment_params *params, float *score);

/* ---- numerics::percentile (src/numerics.rs:125-133) ----------------------------------- */
/* x: len floats, host or (x_on_device != 0) device. */
int apd_percentile(apd_context *ctx, const float *x, uint64_t len, float perc, int x_on_device, float *value);

/* ---- AgglomerativeClustering::clustering (src/clustering.rs:81-110) ------------------- */
/* distances: n*n host (or device if distances_on_device).  ops capacity >= n, roots capacity >= n.
 * roots = dendrogram.clusters() in ascending id order (the reference returns a HashSet).
 * Workspace: about 6 n^2 floats of device memory for the duration of the call (cluster sums, row-sum predictions, the two working
 * copies of the matrix and the buffers they are re-laid out into, member lists); APD_ERR_OOM if it does not fit.
 * Environment (tests / measurement only; results are bit-identical whatever they say): APD_UPGMA_TWO_LAUNCH=0|2 (always / never
 * replay the segment launches), APD_UPGMA_DEFRAG=<merges> (least number of merges between two re-layouts, 0 = never),
 * APD_UPGMA_NO_GRAPH=1 (plain launches instead of graph replay, for profilers), APD_DEBUG_UPGMA_TIMING=1 (phase stamps on stderr). */
int apd_clustering(apd_context *ctx, const float *distances, int distances_on_device, uint32_t n,
                   float perc, apd_cluster_op *ops, uint32_t *n_ops, uint32_t *roots, uint32_t *n_roots,
                   float *threshold);
/* AgglomerativeClustering::cluster_sets (src/clustering.rs:40-76); host-only bookkeeping.
 * members capacity >= n + n_ops + 2, set_off capacity >= n_roots+1. */
int apd_cluster_sets(const apd_cluster_op *ops, uint32_t n_ops, const uint32_t *roots, uint32_t n_roots,
                     uint32_t n, uint32_t *members, uint32_t *set_off, uint32_t *n_sets);

/* ---- companions ---------------------------------------------------------------------- */
/* AutoEncoder::predict over every frame = NDSequence::encoded (src/neural.rs:55-71,
 * src/spectrogram.rs:103-121).  x: [t][d_in]; w_encode: [d_in][latent] (Mat{flat, cols=latent},
 * numerics.rs:171-174); b_encode: [latent]; out: [t][latent].  *_on_device selects HBM pointers
 * for x and out. */
int apd_encode(apd_context *ctx, const float *x, uint64_t t, uint32_t d_in, const float *w_encode,
               const float *b_encode, uint32_t latent, int on_device, float *out);
/* Cepstrum frames of NDSequence::new (src/spectrogram.rs:31-80).  Returns the frame count in
 * *n_frames and bins per frame in *n_bins; out may be NULL to query sizes. */
int apd_cepstrum(apd_context *ctx, const int16_t *samples, uint64_t n_samples, uint32_t fft_size,
                 uint32_t fft_step, uint32_t filter_size, int on_device, float *out, uint64_t *n_frames,
                 uint32_t *n_bins);

/* apd_cepstrum for n_seq recordings stored back to back (sample_offsets: n_seq+1): one launch for the whole corpus.
 * frame_offsets (n_seq+1, host, always written) and out ([frame_offsets[n_seq]][*n_bins], packed) are exactly the
 * `offsets` / `frames` arguments of apd_batch_create, so with on_device != 0 features go from audio to the
 * alignment without leaving HBM.  out may be NULL to query sizes. */
int apd_cepstrum_batch(apd_context *ctx, const int16_t *samples, const uint64_t *sample_offsets, uint32_t n_seq,
                       uint32_t fft_size, uint32_t fft_step, uint32_t filter_size, int on_device, float *out,
                       uint64_t *frame_offsets, uint32_t *n_bins);

/* The same two as RESIDENT objects + enqueue-only calls, for hosts that build features repeatedly or on several GPUs (the
 * reference does it per recording with par_iter, src/main.rs:150-161).  apd_encoder_create / apd_cepstrum_plan_create upload the
 * weights / build and upload the tables and the corpus' offsets ONCE (blocking; the host arrays are free on return; frame_offsets
 * and *n_bins are written as apd_cepstrum_batch writes them); apd_encode_async / apd_cepstrum_batch_async then only ENQUEUE the
 * kernel on the context's stream: device pointers only, no allocation, no table building, no synchronisation
 * (tests/test_gpu_companions.py::test_async_feature_stage_only_enqueues).  A plan serves every corpus with the same sample offsets.
 * Objects die with apd_*_destroy, or device-side with their context (apd_destroy), after which only the destroy call is legal.
 * With several GPUs each context runs the whole corpus' kernel concurrently -- replicated on purpose: the kernels (0.15 ms for cfg 4,
 * 19 ms for cfg 5, whole corpus) cost less than an all-gather of their output would. */
int apd_encoder_create(apd_context *ctx, const float *w_encode, const float *b_encode, uint32_t d_in, uint32_t latent,
                       apd_encoder **encoder);
int apd_encoder_destroy(apd_encoder *encoder);
int apd_encode_async(apd_context *ctx, const apd_encoder *encoder, const float *d_x, uint64_t t, float *d_out);
int apd_cepstrum_plan_create(apd_context *ctx, const uint64_t *sample_offsets, uint32_t n_seq, uint32_t fft_size, uint32_t fft_step,
                             uint32_t filter_size, uint64_t *frame_offsets, uint32_t *n_bins, apd_cepstrum_plan **plan);
int apd_cepstrum_plan_destroy(apd_cepstrum_plan *plan);
int apd_cepstrum_batch_async(apd_context *ctx, const apd_cepstrum_plan *plan, const int16_t *d_samples, float *d_out);

/* NDSequence::interesting_ranges (src/spectrogram.rs:174-216), the "VAT" pre-segmentation that precedes the path:
 * per-frame std, mean of the `moving_average` previous values, percentile threshold, runs longer than min_len.
 * frames: [t][n_bins] (device if on_device).  ranges: (start, stop) frame pairs, host; *n_ranges may exceed capacity. */
int apd_interesting_ranges(apd_context *ctx, const float *frames, uint64_t t, uint32_t n_bins, uint32_t moving_average,
                           float perc, uint64_t min_len, int on_device, uint64_t *ranges, uint64_t capacity,
                           uint64_t *n_ranges);

/* ---- formats either side of the path (host only, no context needed) ------------------------------------------------ */
/* AutoEncoder::from_file / save_file (src/neural.rs:30-44): the bincode 1.x image of
 *   struct AutoEncoder { w_encode, w_decode, b_encode, b_decode: Mat }  (neural.rs:13-19)
 *   struct Mat { flat: Vec<f32>, cols: usize }                             (numerics.rs:171-174)
 * i.e. four times { u64 length, length x f32, u64 cols }, little-endian.  apd_autoencoder_parse checks the image (sizes,
 * shapes D x L / L x D / 1 x L / 1 x D, no trailing bytes) and says where each matrix lies; apd_autoencoder_copy reads one
 * out (byte order handled); w_encode and b_encode are what apd_encode takes. */
typedef struct apd_mat_view { uint64_t offset /* byte offset of flat[0] */, len /* values */, cols; } apd_mat_view;
typedef struct apd_autoencoder_view { apd_mat_view w_encode, w_decode, b_encode, b_decode; } apd_autoencoder_view;
int apd_autoencoder_parse(const void *bytes, uint64_t n_bytes, apd_autoencoder_view *view);
int apd_autoencoder_copy(const void *bytes, const apd_mat_view *mat, float *out /* mat->len floats */);
/* out == NULL: *n_bytes = size of the image.  w_encode: [d_in][latent], w_decode: [latent][d_in], b_*: [latent], [d_in]. */
int apd_autoencoder_serialize(const float *w_encode, const float *w_decode, const float *b_encode, const float *b_decode,
                              uint32_t d_in, uint32_t latent, void *out, uint64_t capacity, uint64_t *n_bytes);

/* Discovery (src/discovery.rs:7-26) and Discovery::from_toml (:28-36) on the text of project/config/Discovery.toml: flat
 * `key = value  # comment` lines; every field exactly once, integers for the usize fields; keys the struct does not have
 * (and anything under a [table] header) are ignored, as serde does without deny_unknown_fields. */
typedef struct apd_discovery {
    uint64_t dft_win, dft_step, ceps_filter, vat_moving;
    float vat_percentile;
    uint64_t vat_min_len, alignment_workers;
    float clustering_percentile, warping_band_percentage, insertion_penalty, deletion_penalty, match_penalty;
    uint64_t auto_encoder;
    float learning_rate;
    uint64_t epochs;
    float epoch_drop, drop;
} apd_discovery;
int apd_discovery_parse_toml(const char *text, apd_discovery *out);

/* Templates::dendrograms (src/reporting.rs:135-169), the strings only: for every root that some op made, the TikZ-qtree
 * bracket string "[.k [<left> <right> ] ]" with leaves rendered as labels[leaf] (the reference puts its image_ref there,
 * reporting.rs:211-221) -- NUL-terminated, concatenated in the order of `roots`; which_root[i] = index into roots of
 * string i (roots never merged have none, reporting.rs:200).  out == NULL: sizes only.  The LaTeX / file output of :171-203
 * is presentation and stays with the caller.  Replay semantics are the reference's HashMap's for ANY ids (a repeated `into`
 * overwrites, an op may name its own `into` as an operand: the string built earlier is embedded); a string beyond 1 GiB or an
 * allocation failure is APD_ERR_OOM, never an abort. */
int apd_dendrograms(const apd_cluster_op *ops, uint32_t n_ops, const uint32_t *roots, uint32_t n_roots,
                    const char *const *labels, uint32_t n_labels, char *out, uint64_t capacity, uint64_t *n_bytes,
                    uint32_t *which_root, uint32_t *n_strings);

#ifdef __cplusplus
}
#endif
#endif

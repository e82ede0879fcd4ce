/*
 * apd_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference's alignment + clustering path
 * (dkohlsdorf/audio_pattern_discovery, src/alignments.rs, src/clustering.rs,
 * src/numerics.rs, src/discovery.rs, src/spectrogram.rs, src/neural.rs).
 *
 * PARITY UNPINNED: the reference is a Rust crate with zero tests, zero golden
 * vectors and no lock file, and no Rust toolchain exists in the build image, so
 * this oracle could be checked neither against reference fixtures nor against
 * reference outputs.  It is pinned only by (a) hand-derived known answers read
 * off the cited source lines (tests/test_oracle.py) and (b) an independent
 * numpy re-derivation of the same recurrence (oracle/np_reference.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The shipped path (audio_pattern_discovery_amd/) never does.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off, no fast-math: every f32
 * add/mul/div/sqrt is one IEEE operation in the order the Rust source performs it).
 */
#ifndef APD_ORACLE_H
#define APD_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* numerics.rs:114-120 -- sqrt(sum_k (x_k-y_k)^2), k ascending, f32 accumulate. */
float orc_euclidean(const float *x, const float *y, uint32_t dim);

/* discovery.rs:38-45 -- warping_band = (pct * n_size as f32) as usize (saturating cast). */
uint64_t orc_warping_band(float pct, uint64_t n_size);

/* alignments.rs:165-180 loop bounds -- number of (i,j) cells construct_alignment visits. */
uint64_t orc_dtw_cells(uint64_t n, uint64_t m, uint64_t band);

/* alignments.rs:107-180 -- Alignment::new + construct_alignment + score for one ordered pair.
 * Dense rolling-row tables; arithmetic and select order exactly as the source.
 * x: [n][dim], y: [m][dim] row-major.  n == 0 xor m == 0 is undefined in the
 * reference (usize underflow at alignments.rs:120); returns NaN for it here. */
float orc_dtw_pair(const float *x, uint64_t n, const float *y, uint64_t m, uint32_t dim,
                   uint64_t band, float ins_pen, float del_pen, float match_pen);

/* Same result as orc_dtw_pair but keeping the reference's COST structure:
 * one hash map keyed (i,j) per pair (SipHash-1-3 like Rust's default hasher),
 * 3 lookups + 1 insert per cell, map dropped per pair.  Used only as the timed
 * "reference-like" CPU baseline (kind = "port"). */
float orc_dtw_pair_hashmap(const float *x, uint64_t n, const float *y, uint64_t m, uint32_t dim,
                           uint64_t band, float ins_pen, float del_pen, float match_pen);

/* alignments.rs:17-67 -- AlignmentWorkers::new + align_all.
 * frames: packed [sum len][dim]; offsets: n_seq+1 frame offsets.  out: n_seq*n_seq
 * row-major, zero-initialised here, diagonal left 0.0, out[i*n+j] = score(x=i,y=j).
 * workers = alignment_workers (contiguous row blocks of n/workers+1, alignments.rs:33-37).
 * use_hashmap != 0 selects orc_dtw_pair_hashmap.  Returns 0, or -1 on bad args. */
int orc_align_all(const float *frames, const uint64_t *offsets, uint32_t n_seq, uint32_t dim,
                  float band_pct, float ins_pen, float del_pen, float match_pen,
                  uint32_t workers, int use_hashmap, float *out);

/* Time-bounded sample of ordered pairs for the CPU baseline: computes pairs
 * (pi[k], pj[k]) for k < n_pairs with `workers` threads; returns cells visited via *cells. */
int orc_align_sample(const float *frames, const uint64_t *offsets, uint32_t n_seq, uint32_t dim,
                     float band_pct, float ins_pen, float del_pen, float match_pen,
                     const uint32_t *pi, const uint32_t *pj, uint64_t n_pairs,
                     uint32_t workers, int use_hashmap, float *out, uint64_t *cells);

/* numerics.rs:125-133 -- drop NaN, ascending sort, element at (len as f32 * perc) as usize.
 * Returns 0 and *value, or -1 where the reference would panic (index out of range). */
int orc_percentile(const float *x, uint64_t len, float perc, float *value);

/* clustering.rs:8-25 */
enum { ORC_S2S = 0, ORC_S2C = 1, ORC_C2S = 2, ORC_C2C = 3 };
typedef struct {
    uint32_t merge_i, merge_j, into;
    float distance;
    uint32_t operation;
} orc_cluster_op;

/* clustering.rs:81-210 -- literal AgglomerativeClustering::clustering.  HashSet
 * iteration (clustering.rs:180-181) is replaced by ascending cluster id, strict '<'.
 * ops capacity >= n; roots capacity >= n.  Returns 0 / -1 (reference would panic). */
int orc_clustering(const float *dist, uint32_t n, float perc, orc_cluster_op *ops,
                   uint32_t *n_ops, uint32_t *roots, uint32_t *n_roots, float *threshold);

/* clustering.rs:81-210 -- the same function as orc_clustering (same ops, same linkage bits, same roots), O(n^3)
 * worst case: linkages are cached per cluster pair and only the merged cluster's row and column are re-summed, in the
 * reference's own order.  Proven equal to the literal loop in tests/test_oracle.py; used to check the device UPGMA at
 * N in the thousands, where the literal O(n^4) loop cannot go.  OpenMP over independent linkages. */
/* OpenMP threads of orc_clustering_fast (0 = libgomp default, which oversubscribes a CPU-share-limited host). */
void orc_set_threads(int threads);
int orc_clustering_fast(const float *dist, uint32_t n, float perc, orc_cluster_op *ops,
                        uint32_t *n_ops, uint32_t *roots, uint32_t *n_roots, float *threshold);

/* clustering.rs:40-76 -- cluster_sets: members[] receives the concatenated leaf lists
 * (in the reference's replay order), set_off[n_sets+1] the boundaries; roots never
 * merged are omitted.  Roots are taken in the order given. */
int orc_cluster_sets(const orc_cluster_op *ops, uint32_t n_ops, const uint32_t *roots,
                     uint32_t n_roots, uint32_t n, uint32_t *members, uint32_t *set_off,
                     uint32_t *n_sets);

/* neural.rs:55-71 + numerics.rs:12-29,71-73,227-258,297-319 -- AutoEncoder::predict per frame.
 * x: [t][d_in]; w: [d_in][latent] row-major (Mat{flat, cols=latent}); b: [latent]; out: [t][latent]. */
void orc_encode(const float *x, uint64_t t, uint32_t d_in, const float *w, const float *b,
                uint32_t latent, float *out);

/* spectrogram.rs:31-80 + numerics.rs:60-66,78-109 -- cepstrum frames of NDSequence::new.
 * DFT and DCT-I follow the mathematical definitions (rustfft 3.0.0 forward unnormalised;
 * rustdct DCT-I = x0/2 + (-1)^k x_{N-1}/2 + sum x_n cos(pi n k/(N-1))) evaluated in f64
 * and rounded to f32 at the points where the reference holds f32 -- third-party
 * arithmetic, PARITY UNPINNED.  Returns frame count; *n_bins receives bins per frame.
 * out may be NULL to query sizes. */
uint64_t orc_cepstrum(const int16_t *samples, uint64_t n_samples, uint32_t fft_size,
                      uint32_t fft_step, uint32_t filter_size, float *out, uint32_t *n_bins);

/* spectrogram.rs:174-187 -- NDSequence::variance(k): per-frame population std, then the mean of the k PREVIOUS
 * values (0.0 for i < k).  frames: [t][n_bins]; out: [t]. */
void orc_variance(const float *frames, uint64_t t, uint32_t n_bins, uint32_t k, float *out);

/* spectrogram.rs:192-216 -- NDSequence::interesting_ranges: threshold = percentile(variance(k), perc); a range opens
 * at the first value >= threshold (the scan STARTS in the recording state at 0), closes at the first value below it,
 * and is kept if stop - start > min_len; a range still open at the end is dropped.  ranges: (start, stop) pairs.
 * Returns 0, or -1 where the reference panics (percentile index out of range). */
int orc_interesting_ranges(const float *frames, uint64_t t, uint32_t n_bins, uint32_t moving_average, float perc,
                           uint64_t min_len, uint64_t *ranges, uint64_t capacity, uint64_t *n_ranges);

#ifdef __cplusplus
}
#endif
#endif

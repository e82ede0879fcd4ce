//! Raw declarations of include/apd.h (libapd_hip.so).  UNCOMPILED: no Rust toolchain in the build image; kept in step with
//! the header by hand -- the ctypes table in audio_pattern_discovery_amd/_lib.py is the machine-checked twin
//! (tests/test_host_abi.py compares it with the header symbol by symbol).
#![allow(non_camel_case_types, dead_code)]
use std::os::raw::{c_char, c_int, c_void};

#[repr(C)] pub struct apd_context { _p: [u8; 0] }
#[repr(C)] pub struct apd_batch { _p: [u8; 0] }
#[repr(C)] pub struct apd_comm { _p: [u8; 0] }
#[repr(C)] pub struct apd_multi { _p: [u8; 0] }
#[repr(C)] pub struct apd_multi_batch { _p: [u8; 0] }
#[repr(C)] pub struct apd_encoder { _p: [u8; 0] }
#[repr(C)] pub struct apd_cepstrum_plan { _p: [u8; 0] }

pub const APD_OK: c_int = 0;
pub const APD_ERR_INVALID_ARG: c_int = -1;
pub const APD_ERR_NO_DEVICE: c_int = -2;
pub const APD_ERR_HIP: c_int = -3;
pub const APD_ERR_OOM: c_int = -4;
pub const APD_ERR_EMPTY_SEQUENCE: c_int = -5;
pub const APD_ERR_BAND_TOO_WIDE: c_int = -6;
pub const APD_ERR_INDEX: c_int = -7;
pub const APD_ERR_UNSUPPORTED: c_int = -8;
pub const APD_ERR_INCOMPLETE: c_int = -9;
pub const APD_ERR_COMM: c_int = -10;
pub const APD_COMM_ID_BYTES: usize = 128;

/// The four Discovery fields the path reads (discovery.rs:17-20).
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct apd_align_config {
    pub warping_band_percentage: f32,
    pub insertion_penalty: f32,
    pub deletion_penalty: f32,
    pub match_penalty: f32,
}

/// AlignmentParams (alignments.rs:77-83).
#[repr(C)] #[derive(Clone, Copy, Debug)]
pub struct apd_alignment_params {
    pub warping_band: u64,
    pub insertion_penalty: f32,
    pub deletion_penalty: f32,
    pub match_penalty: f32,
}

/// ClusteringOperation (clustering.rs:19-25); operation: 0 S2S, 1 S2C, 2 C2S, 3 C2C (clustering.rs:8-13).
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct apd_cluster_op {
    pub merge_i: u32,
    pub merge_j: u32,
    pub into: u32,
    pub distance: f32,
    pub operation: u32,
}

#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct apd_mat_view { pub offset: u64, pub len: u64, pub cols: u64 }
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct apd_autoencoder_view { pub w_encode: apd_mat_view, pub w_decode: apd_mat_view, pub b_encode: apd_mat_view, pub b_decode: apd_mat_view }

/// Discovery (discovery.rs:7-26) as apd_discovery_parse_toml fills it.
#[repr(C)] #[derive(Clone, Copy, Debug, Default)]
pub struct apd_discovery {
    pub dft_win: u64, pub dft_step: u64, pub ceps_filter: u64, pub vat_moving: u64,
    pub vat_percentile: f32,
    pub vat_min_len: u64, pub alignment_workers: u64,
    pub clustering_percentile: f32, pub warping_band_percentage: f32, pub insertion_penalty: f32, pub deletion_penalty: f32, pub match_penalty: f32,
    pub auto_encoder: u64,
    pub learning_rate: f32,
    pub epochs: u64,
    pub epoch_drop: f32, pub drop: f32,
}

extern "C" {
    // context
    pub fn apd_create(device: c_int, ctx: *mut *mut apd_context) -> c_int;
    pub fn apd_destroy(ctx: *mut apd_context) -> c_int;
    pub fn apd_set_stream(ctx: *mut apd_context, hip_stream: *mut c_void) -> c_int;
    pub fn apd_synchronize(ctx: *mut apd_context) -> c_int;
    pub fn apd_status_string(status: c_int) -> *const c_char;
    pub fn apd_last_error(ctx: *mut apd_context) -> *const c_char;
    pub fn apd_set_timing(ctx: *mut apd_context, enabled: c_int) -> c_int;
    pub fn apd_last_kernel_ms(ctx: *mut apd_context) -> f32;
    pub fn apd_set_variant(ctx: *mut apd_context, variant: c_int) -> c_int;
    pub fn apd_set_distance_mode(ctx: *mut apd_context, mode: c_int, tau: f32) -> c_int;
    pub fn apd_selftest(ctx: *mut apd_context) -> c_int;
    pub fn apd_selftest_sqrt(ctx: *mut apd_context, first_bits: u32, count: u64, mismatches: *mut u64, first_mismatch: *mut u32,
                             raw_ulp_hist: *mut u64) -> c_int;
    pub fn apd_stream_busy(ctx: *mut apd_context, busy: *mut c_int) -> c_int;
    pub fn apd_debug_affinity_probe(bound: *mut apd_context, checked: *mut apd_context, enabled: *mut c_int) -> c_int;
    pub fn apd_set_fault_injection(ctx: *mut apd_context, drop_tiles: u32) -> c_int;
    // Discovery::alignment_params (discovery.rs:38-45)
    pub fn apd_discovery_alignment_params(cfg: *const apd_align_config, n_size: u64, out: *mut apd_alignment_params) -> c_int;
    // AlignmentWorkers::new (alignments.rs:17-26)
    pub fn apd_batch_create(ctx: *mut apd_context, frames: *const f32, offsets: *const u64, n_seq: u32, dim: u32,
                            frames_on_device: c_int, batch: *mut *mut apd_batch) -> c_int;
    pub fn apd_batch_destroy(batch: *mut apd_batch) -> c_int;
    pub fn apd_batch_len(batch: *const apd_batch) -> u32;
    pub fn apd_batch_refill(ctx: *mut apd_context, batch: *mut apd_batch, frames: *const f32, frames_on_device: c_int) -> c_int;
    pub fn apd_batch_nonfinite(ctx: *mut apd_context, batch: *const apd_batch, nonfinite: *mut c_int) -> c_int;
    // AlignmentWorkers::align_all (alignments.rs:31-67)
    pub fn apd_align_all(ctx: *mut apd_context, batch: *const apd_batch, cfg: *const apd_align_config, out: *mut f32) -> c_int;
    pub fn apd_align_all_device_async(ctx: *mut apd_context, batch: *const apd_batch, cfg: *const apd_align_config, d_out: *mut f32) -> c_int;
    pub fn apd_tile_size() -> u32;
    pub fn apd_num_tiles(n_seq: u32) -> u64;
    pub fn apd_rank_tiles(n_seq: u32, rank: u32, world: u32) -> u64;
    pub fn apd_slab_floats(n_seq: u32, world: u32) -> u64;
    pub fn apd_align_tiles_async(ctx: *mut apd_context, batch: *const apd_batch, cfg: *const apd_align_config, rank: u32, world: u32,
                                 d_slab: *mut f32) -> c_int;
    pub fn apd_unpack_tiles_async(ctx: *mut apd_context, batch: *const apd_batch, world: u32, d_gathered: *const f32, d_out: *mut f32) -> c_int;
    // several GPUs, process per GPU
    pub fn apd_comm_unique_id(id_bytes: *mut c_void) -> c_int;
    pub fn apd_comm_create(ctx: *mut apd_context, id_bytes: *const c_void, rank: u32, world: u32, comm: *mut *mut apd_comm) -> c_int;
    pub fn apd_comm_destroy(comm: *mut apd_comm) -> c_int;
    pub fn apd_comm_count(comm: *const apd_comm, world: *mut u32) -> c_int;
    pub fn apd_comm_rank(comm: *const apd_comm, rank: *mut u32) -> c_int;
    pub fn apd_align_all_sharded_async(ctx: *mut apd_context, comm: *mut apd_comm, batch: *const apd_batch, cfg: *const apd_align_config,
                                       d_out: *mut f32) -> c_int;
    pub fn apd_all_gather_async(ctx: *mut apd_context, comm: *mut apd_comm, d_send: *const f32, d_recv: *mut f32, count: u64) -> c_int;
    // several GPUs, one process: one-shot and persistent handle
    pub fn apd_align_all_multi(devices: *const c_int, n_devices: u32, frames: *const f32, offsets: *const u64, n_seq: u32, dim: u32,
                               cfg: *const apd_align_config, out: *mut f32, ranks_seen: *mut u32) -> c_int;
    pub fn apd_dtw_all_pairs(frames: *const f32, offsets: *const u64, n_seq: u32, dim: u32, band_pct: f32, ins_pen: f32, del_pen: f32,
                             match_pen: f32, n_devices: c_int, out: *mut f32) -> c_int;
    pub fn apd_upgma(dist: *const f32, n: u32, perc: f32, ops: *mut apd_cluster_op, n_ops: *mut u32, roots: *mut u32, n_roots: *mut u32) -> c_int;
    pub fn apd_multi_create(devices: *const c_int, n_devices: u32, multi: *mut *mut apd_multi) -> c_int;
    pub fn apd_multi_destroy(multi: *mut apd_multi) -> c_int;
    pub fn apd_multi_size(multi: *const apd_multi) -> u32;
    pub fn apd_multi_ranks_seen(multi: *const apd_multi, ranks: *mut u32) -> c_int;
    pub fn apd_multi_collective(multi: *const apd_multi) -> *const c_char;
    pub fn apd_multi_last_error(multi: *const apd_multi) -> *const c_char;
    pub fn apd_multi_context(multi: *mut apd_multi, i: u32) -> *mut apd_context;
    pub fn apd_multi_batch_create(multi: *mut apd_multi, frames: *const f32, d_frames: *const *const f32, offsets: *const u64, n_seq: u32,
                                  dim: u32, batch: *mut *mut apd_multi_batch) -> c_int;
    pub fn apd_multi_batch_refill(multi: *mut apd_multi, batch: *mut apd_multi_batch, frames: *const f32, d_frames: *const *const f32) -> c_int;
    pub fn apd_multi_batch_destroy(batch: *mut apd_multi_batch) -> c_int;
    pub fn apd_multi_align_all_async(multi: *mut apd_multi, batch: *const apd_multi_batch, cfg: *const apd_align_config, d_out: *mut f32) -> c_int;
    pub fn apd_multi_synchronize(multi: *mut apd_multi) -> c_int;
    pub fn apd_multi_result(multi: *const apd_multi) -> *const f32;
    pub fn apd_multi_align_all(multi: *mut apd_multi, batch: *const apd_multi_batch, cfg: *const apd_align_config, out: *mut f32) -> c_int;
    // device buffers
    pub fn apd_device_alloc(ctx: *mut apd_context, bytes: u64, d_ptr: *mut *mut c_void) -> c_int;
    pub fn apd_device_free(ctx: *mut apd_context, d_ptr: *mut c_void) -> c_int;
    pub fn apd_copy_to_device(ctx: *mut apd_context, d_dst: *mut c_void, src: *const c_void, bytes: u64) -> c_int;
    pub fn apd_copy_to_host(ctx: *mut apd_context, dst: *mut c_void, d_src: *const c_void, bytes: u64) -> c_int;
    pub fn apd_device_fill(ctx: *mut apd_context, d_dst: *mut c_void, byte_value: c_int, bytes: u64) -> c_int;
    pub fn apd_runtime_info(out: *mut c_char, capacity: u64) -> u64;
    // host-side views of the sharding, work accounting
    pub fn apd_length_order(offsets: *const u64, n_seq: u32, order: *mut u32) -> c_int;
    pub fn apd_rank_tile_list(n_seq: u32, rank: u32, world: u32, tile_ab: *mut u32, capacity: u64, n_tiles: *mut u64) -> c_int;
    pub fn apd_unpack_tiles_host(offsets: *const u64, n_seq: u32, world: u32, gathered: *const f32, out: *mut f32) -> c_int;
    pub fn apd_align_work(offsets: *const u64, n_seq: u32, dim: u32, cfg: *const apd_align_config, rank: u32, world: u32,
                          pairs: *mut u64, cells: *mut u64, alg_bytes: *mut u64) -> c_int;
    // Alignment (alignments.rs:107-180)
    pub fn apd_align_pair(ctx: *mut apd_context, x: *const f32, n: u64, y: *const f32, m: u64, dim: u32,
                          params: *const apd_alignment_params, score: *mut f32) -> c_int;
    // numerics::percentile, AgglomerativeClustering
    pub fn apd_percentile(ctx: *mut apd_context, x: *const f32, len: u64, perc: f32, x_on_device: c_int, value: *mut f32) -> c_int;
    pub fn apd_clustering(ctx: *mut apd_context, distances: *const f32, distances_on_device: c_int, n: u32, perc: f32,
                          ops: *mut apd_cluster_op, n_ops: *mut u32, roots: *mut u32, n_roots: *mut u32, threshold: *mut f32) -> c_int;
    pub fn apd_cluster_sets(ops: *const apd_cluster_op, n_ops: u32, roots: *const u32, n_roots: u32, n: u32,
                            members: *mut u32, set_off: *mut u32, n_sets: *mut u32) -> c_int;
    // companions
    pub fn apd_encode(ctx: *mut apd_context, x: *const f32, t: u64, d_in: u32, w_encode: *const f32, b_encode: *const f32,
                      latent: u32, on_device: c_int, out: *mut f32) -> c_int;
    pub fn apd_cepstrum(ctx: *mut apd_context, samples: *const i16, n_samples: u64, fft_size: u32, fft_step: u32, filter_size: u32,
                        on_device: c_int, out: *mut f32, n_frames: *mut u64, n_bins: *mut u32) -> c_int;
    pub fn apd_cepstrum_batch(ctx: *mut apd_context, samples: *const i16, sample_offsets: *const u64, n_seq: u32, fft_size: u32,
                              fft_step: u32, filter_size: u32, on_device: c_int, out: *mut f32, frame_offsets: *mut u64, n_bins: *mut u32) -> c_int;
    // the same as resident objects + enqueue-only calls
    pub fn apd_encoder_create(ctx: *mut apd_context, w_encode: *const f32, b_encode: *const f32, d_in: u32, latent: u32,
                              encoder: *mut *mut apd_encoder) -> c_int;
    pub fn apd_encoder_destroy(encoder: *mut apd_encoder) -> c_int;
    pub fn apd_encode_async(ctx: *mut apd_context, encoder: *const apd_encoder, d_x: *const f32, t: u64, d_out: *mut f32) -> c_int;
    pub fn apd_cepstrum_plan_create(ctx: *mut apd_context, sample_offsets: *const u64, n_seq: u32, fft_size: u32, fft_step: u32,
                                    filter_size: u32, frame_offsets: *mut u64, n_bins: *mut u32, plan: *mut *mut apd_cepstrum_plan) -> c_int;
    pub fn apd_cepstrum_plan_destroy(plan: *mut apd_cepstrum_plan) -> c_int;
    pub fn apd_cepstrum_batch_async(ctx: *mut apd_context, plan: *const apd_cepstrum_plan, d_samples: *const i16, d_out: *mut f32) -> c_int;
    pub fn apd_interesting_ranges(ctx: *mut apd_context, frames: *const f32, t: u64, n_bins: u32, moving_average: u32, perc: f32,
                                  min_len: u64, on_device: c_int, ranges: *mut u64, capacity: u64, n_ranges: *mut u64) -> c_int;
    // formats either side of the path (host only)
    pub fn apd_autoencoder_parse(bytes: *const c_void, n_bytes: u64, view: *mut apd_autoencoder_view) -> c_int;
    pub fn apd_autoencoder_copy(bytes: *const c_void, mat: *const apd_mat_view, out: *mut f32) -> c_int;
    pub fn apd_autoencoder_serialize(w_encode: *const f32, w_decode: *const f32, b_encode: *const f32, b_decode: *const f32, d_in: u32,
                                     latent: u32, out: *mut c_void, capacity: u64, n_bytes: *mut u64) -> c_int;
    pub fn apd_discovery_parse_toml(text: *const c_char, out: *mut apd_discovery) -> c_int;
    pub fn apd_dendrograms(ops: *const apd_cluster_op, n_ops: u32, roots: *const u32, n_roots: u32, labels: *const *const c_char,
                           n_labels: u32, out: *mut c_char, capacity: u64, n_bytes: *mut u64, which_root: *mut u32, n_strings: *mut u32) -> c_int;
}

/// A negative status becomes a panic carrying the library's text: the reference panics in the same places (poisoned mutex
/// alignments.rs:56, usize underflow :120, percentile index numerics.rs:132).
pub fn check(status: c_int) {
    if status != APD_OK {
        let text = unsafe { std::ffi::CStr::from_ptr(apd_status_string(status)) }.to_string_lossy().into_owned();
        panic!("libapd_hip: {} ({})", text, status);
    }
}

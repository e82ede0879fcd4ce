//! Drop-in for src/alignments.rs of dkohlsdorf/audio_pattern_discovery: the same public items (alignments.rs:11-14, 17, 31,
//! 77-93, 99-125, 165), bodies on the MI355X through libapd_hip.so.  UNCOMPILED (no Rust toolchain in the build image).
use crate::apd_sys::*;
use crate::discovery::Discovery;
use crate::spectrogram::NDSequence;
use std::collections::HashMap;
use std::os::raw::c_int;
use std::sync::{Arc, Mutex};

/// HIP device ordinals from APD_DEVICES ("0,1,2,3"), default device 0.
fn devices() -> Vec<c_int> {
    std::env::var("APD_DEVICES").ok()
        .map(|s| s.split(',').filter_map(|t| t.trim().parse().ok()).collect::<Vec<c_int>>())
        .filter(|v| !v.is_empty())
        .unwrap_or_else(|| vec![0])
}

/// The GPU side of an AlignmentWorkers: made by the first align_all, kept for the next ones (create once, align many).
struct Resident {
    multi: *mut apd_multi,              // contexts, RCCL communicators, worker threads: one per device
    batch: *mut apd_multi_batch,        // the corpus resident on every device
}
unsafe impl Send for Resident {}
impl Drop for Resident {
    fn drop(&mut self) { unsafe { apd_multi_destroy(self.multi); } }   // also destroys the batch
}

/// alignments.rs:11-14 -- field names and types as in the reference, so main.rs:189-195 compiles unchanged.
pub struct AlignmentWorkers {
    pub data: Arc<Vec<NDSequence>>,
    pub result: Arc<Mutex<Vec<f32>>>,
    resident: Option<Resident>,
}

impl AlignmentWorkers {
    /// alignments.rs:17-26: takes the sequences, allocates the n*n zero matrix.
    pub fn new(data: Vec<NDSequence>) -> AlignmentWorkers {
        let n = data.len();
        AlignmentWorkers { data: Arc::from(data), result: Arc::from(Mutex::from(vec![0.0f32; n * n])), resident: None }
    }

    /// alignments.rs:31-67: blocks until result[i*n+j] = score(x = data[i], y = data[j]) for every i != j; diagonal 0.0.
    /// `alignment_workers` is not read: the workers are the GPUs of APD_DEVICES.
    pub fn align_all(&mut self, params: &Discovery) {
        let n = self.data.len();
        if n == 0 { return; }
        let cfg = apd_align_config {
            warping_band_percentage: params.warping_band_percentage, insertion_penalty: params.insertion_penalty,
            deletion_penalty: params.deletion_penalty, match_penalty: params.match_penalty,
        };
        if self.resident.is_none() {
            let dim = self.data[0].n_bins as u32;
            let mut offsets = vec![0u64; n + 1];
            let mut frames: Vec<f32> = Vec::new();
            for (s, seq) in self.data.iter().enumerate() {
                offsets[s + 1] = offsets[s] + seq.len() as u64;
                frames.extend_from_slice(&seq.frames);                    // NDSequence.frames: [T][n_bins] row-major (spectrogram.rs:16-17)
            }
            let devs = devices();
            let mut multi = std::ptr::null_mut();
            let mut batch = std::ptr::null_mut();
            unsafe {
                check(apd_multi_create(devs.as_ptr(), devs.len() as u32, &mut multi));
                let rc = apd_multi_batch_create(multi, frames.as_ptr(), std::ptr::null(), offsets.as_ptr(), n as u32, dim, &mut batch);
                if rc != APD_OK { let why = last_error(multi); apd_multi_destroy(multi); panic!("libapd_hip: {} ({})", why, rc); }
            }
            self.resident = Some(Resident { multi, batch });
        }
        let r = self.resident.as_ref().unwrap();
        let mut result = self.result.lock().unwrap();                     // the reference panics on a poisoned mutex too (:56)
        let rc = unsafe { apd_multi_align_all(r.multi, r.batch, &cfg, result.as_mut_ptr()) };
        if rc != APD_OK { panic!("libapd_hip: {} ({})", last_error(r.multi), rc); }   // APD_ERR_INCOMPLETE: NaN left where no score was written
    }
}

fn last_error(multi: *mut apd_multi) -> String {
    unsafe { std::ffi::CStr::from_ptr(apd_multi_last_error(multi)) }.to_string_lossy().into_owned()
}

/// alignments.rs:77-83
#[derive(Clone, Debug)]
pub struct AlignmentParams {
    pub warping_band: usize,
    pub insertion_penalty: f32,
    pub deletion_penalty: f32,
    pub match_penalty: f32,
}

impl AlignmentParams {
    /// alignments.rs:86-93
    pub fn default(len: usize) -> AlignmentParams {
        AlignmentParams { warping_band: len, insertion_penalty: 1.0, deletion_penalty: 1.0, match_penalty: 1.0 }
    }
}

thread_local! {
    /// One context per thread for the single-pair entry point (contexts are cheap; streams are not shared between threads).
    static PAIR_CTX: *mut apd_context = {
        let mut ctx = std::ptr::null_mut();
        unsafe { check(apd_create(devices()[0], &mut ctx)); }
        ctx
    };
}

/// alignments.rs:99-104.  `sparse` stays in the struct for source compatibility but is never filled: no caller of the
/// reference reads it (only score() is used), and the DP table never leaves the GPU's registers.
#[derive(Debug)]
pub struct Alignment {
    pub n: usize,
    pub m: usize,
    pub sparse: HashMap<(usize, usize), f32>,
    score: f32,
}

impl Alignment {
    /// alignments.rs:107-111
    pub fn new() -> Alignment {
        Alignment { n: 0, m: 0, sparse: HashMap::new(), score: std::f32::INFINITY }
    }

    /// alignments.rs:116-125: INF for two empty sequences, else D[n-1][m-1] / (n + m) (INF if that cell was never visited).
    pub fn score(&self) -> f32 {
        if self.n == 0 && self.m == 0 { std::f32::INFINITY } else { self.score }
    }

    /// alignments.rs:165-180
    pub fn construct_alignment(&mut self, x: &NDSequence, y: &NDSequence, params: &AlignmentParams) {
        self.n = x.len();
        self.m = y.len();
        let p = apd_alignment_params { warping_band: params.warping_band as u64, insertion_penalty: params.insertion_penalty,
                                       deletion_penalty: params.deletion_penalty, match_penalty: params.match_penalty };
        PAIR_CTX.with(|ctx| unsafe {
            check(apd_align_pair(*ctx, x.frames.as_ptr(), self.n as u64, y.frames.as_ptr(), self.m as u64, x.n_bins as u32, &p, &mut self.score));
        });
    }
}

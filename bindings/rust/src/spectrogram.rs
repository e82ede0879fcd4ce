//! Drop-in for the path-facing part of src/spectrogram.rs of dkohlsdorf/audio_pattern_discovery: NDSequence with the same public
//! fields (spectrogram.rs:13-24) and `new`, `vec`, `encoded`, `len`, `at`, `interesting_ranges` (spectrogram.rs:31-121, 152-216),
//! bodies on the MI355X through libapd_hip.so.  UNCOMPILED (no Rust toolchain in the build image).
//!
//! Not carried over: `img_spectrogram` / `len_spectrogram` (spectrogram.rs:123-161) -- PNG rendering for the HTML report, out of the
//! path's scope.  They read the `spectrogram` field, which `new` below therefore leaves empty; a maintainer who keeps the report keeps
//! the reference's magnitude loop (spectrogram.rs:74-79) for that one field.  `Slice` (spectrogram.rs:223-260) is plain bookkeeping
//! and stays as it is in the reference.
use crate::apd_sys::*;
use crate::audio::AudioData;
use crate::neural::AutoEncoder;
use std::os::raw::c_int;

pub struct NDSequence {
    pub n_bins: usize,
    pub frames: Vec<f32>,
    pub dft_win: usize,
    pub spectrogram: Vec<f32>,
    pub audio_id: usize,
}

/// One context on the first device of APD_DEVICES, made on first use and kept for the process (the feature stage is per recording,
/// main.rs:150-161: a context per call would cost more than the kernel).
pub(crate) fn feature_context() -> *mut apd_context {
    use std::sync::Once;
    static mut CTX: *mut apd_context = std::ptr::null_mut();
    static INIT: Once = Once::new();
    unsafe {
        INIT.call_once(|| {
            let dev: c_int = std::env::var("APD_DEVICES").ok().and_then(|s| s.split(',').next().and_then(|t| t.trim().parse().ok())).unwrap_or(0);
            let mut ctx = std::ptr::null_mut();
            check(apd_create(dev, &mut ctx));
            CTX = ctx;
        });
        CTX
    }
}

pub(crate) fn check(rc: c_int) {
    if rc != APD_OK {
        let text = unsafe { std::ffi::CStr::from_ptr(apd_status_string(rc)) }.to_string_lossy().into_owned();
        panic!("libapd_hip: {} ({})", text, rc);                       // the reference's feature code panics too (unwrap, index)
    }
}

impl NDSequence {
    /// spectrogram.rs:31-94.  `frames` = the cepstrum (Hamming window, DFT magnitudes, triangular filterbank, ln, DCT-I, drop 4,
    /// mean removal) of every fft_step-th window, computed by apd_cepstrum; n_bins from the library (= K - 4).
    pub fn new(fft_size: usize, fft_step: usize, filter_size: usize, raw_audio: &AudioData) -> NDSequence {
        let ctx = feature_context();
        let (mut n_frames, mut n_bins) = (0u64, 0u32);
        let samples = &raw_audio.data;                                  // Vec<i16> (audio.rs)
        unsafe {
            check(apd_cepstrum(ctx, samples.as_ptr(), samples.len() as u64, fft_size as u32, fft_step as u32, filter_size as u32, 0,
                               std::ptr::null_mut(), &mut n_frames, &mut n_bins));          // sizes first
        }
        let mut frames = vec![0f32; n_frames as usize * n_bins as usize];
        if !frames.is_empty() {
            unsafe {
                check(apd_cepstrum(ctx, samples.as_ptr(), samples.len() as u64, fft_size as u32, fft_step as u32, filter_size as u32, 0,
                                   frames.as_mut_ptr(), &mut n_frames, &mut n_bins));
            }
        }
        NDSequence { audio_id: raw_audio.id, n_bins: n_bins as usize, frames, dft_win: fft_size / 2 - 10, spectrogram: Vec::new() }
    }

    /// spectrogram.rs:99-101
    pub fn vec(&self, t: usize) -> &[f32] {
        &self.frames[t * self.n_bins..(t + 1) * self.n_bins]
    }

    /// spectrogram.rs:103-121: AutoEncoder::predict on every frame -- ONE apd_encode over the whole sequence.
    pub fn encoded(&self, nn: &AutoEncoder) -> NDSequence {
        let latent = nn.n_latent();
        let mut flat = vec![0f32; self.len() * latent];
        if !flat.is_empty() {
            unsafe {
                check(apd_encode(feature_context(), self.frames.as_ptr(), self.len() as u64, self.n_bins as u32, nn.w_encode.flat.as_ptr(),
                                 nn.b_encode.flat.as_ptr(), latent as u32, 0, flat.as_mut_ptr()));
            }
        }
        NDSequence { audio_id: self.audio_id, n_bins: latent, frames: flat, dft_win: self.dft_win, spectrogram: self.spectrogram.clone() }
    }

    /// spectrogram.rs:152-154
    pub fn len(&self) -> usize {
        self.frames.len() / self.n_bins
    }

    /// spectrogram.rs:167-169
    pub fn at(&self, t: usize, f: usize) -> f32 {
        self.frames[t * self.n_bins + f]
    }

    /// spectrogram.rs:192-216 (and `variance`, :174-187, inside the library): per-frame std, moving average, percentile threshold,
    /// runs longer than min_len.  The percentile of a too-short variance vector panics in the reference (numerics.rs:132):
    /// APD_ERR_INDEX panics here.
    pub fn interesting_ranges(&self, moving_average: usize, perc: f32, min_len: usize) -> Vec<Slice> {
        let mut ranges = vec![0u64; 2 * (self.len() / (min_len + 1) + 1)];
        let mut n = 0u64;
        unsafe {
            check(apd_interesting_ranges(feature_context(), self.frames.as_ptr(), self.len() as u64, self.n_bins as u32, moving_average as u32,
                                         perc, min_len as u64, 0, ranges.as_mut_ptr(), (ranges.len() / 2) as u64, &mut n));
        }
        (0..n as usize).map(|k| Slice::new(ranges[2 * k] as usize, ranges[2 * k + 1] as usize, self)).collect()
    }
}

/// spectrogram.rs:223-233 (the rest of `impl Slice` is unchanged reference code)
pub struct Slice<'a> {
    pub start: usize,
    pub stop: usize,
    pub sequence: &'a NDSequence,
}

impl<'a> Slice<'a> {
    pub fn new(start: usize, stop: usize, sequence: &'a NDSequence) -> Slice<'a> {
        Slice { start, stop, sequence }
    }
}

//! Drop-in for the path-facing part of src/neural.rs of dkohlsdorf/audio_pattern_discovery: `AutoEncoder` with the same public
//! fields (neural.rs:13-18) and `n_latent`, `from_file`, `save_file`, `predict` (neural.rs:21-71).  UNCOMPILED (no Rust toolchain in
//! the build image).
//!
//! `from_file` / `save_file` keep the reference's bincode calls when the crate still depends on bincode; the variants below go through
//! the library's own reader / writer of the same byte layout (apd_autoencoder_parse / _copy / _serialize: bincode 1.x defaults,
//! field order w_encode, w_decode, b_encode, b_decode, each {flat: Vec<f32>, cols: usize}), so that the weights a reference run saved
//! load without bincode.  Training (`new`, `step_decay`, `take_step`, `train`: neural.rs:26-28, 46-53, 73-94) is outside the path and
//! stays reference code.
use crate::apd_sys::*;
use crate::error::*;
use crate::numerics::Mat;
use crate::spectrogram::{check, feature_context};
use std::fs::File;
use std::io::prelude::*;

#[derive(Clone)]
pub struct AutoEncoder {
    pub w_encode: Mat,
    pub w_decode: Mat,
    pub b_encode: Mat,
    pub b_decode: Mat,
}

impl AutoEncoder {
    /// neural.rs:21-23
    pub fn n_latent(&self) -> usize {
        self.b_encode.cols
    }

    /// neural.rs:30-36.  The reference `unwrap`s a malformed file (panic); so does this.
    pub fn from_file(file: &str) -> Result<AutoEncoder> {
        let mut fp = File::open(file)?;
        let mut buf: Vec<u8> = vec![];
        let _ = fp.read_to_end(&mut buf)?;
        let mut view = apd_autoencoder_view::default();
        unsafe { check(apd_autoencoder_parse(buf.as_ptr() as *const _, buf.len() as u64, &mut view)); }
        let mat = |m: &apd_mat_view| -> Mat {
            let mut flat = vec![0f32; m.len as usize];
            unsafe { check(apd_autoencoder_copy(buf.as_ptr() as *const _, m, flat.as_mut_ptr())); }
            Mat { flat, cols: m.cols as usize }
        };
        Ok(AutoEncoder { w_encode: mat(&view.w_encode), w_decode: mat(&view.w_decode), b_encode: mat(&view.b_encode), b_decode: mat(&view.b_decode) })
    }

    /// neural.rs:39-44
    pub fn save_file(&self, file: &str) -> Result<()> {
        let (d_in, latent) = ((self.w_encode.flat.len() / self.w_encode.cols) as u32, self.w_encode.cols as u32);
        let mut n_bytes = 0u64;
        unsafe {
            check(apd_autoencoder_serialize(self.w_encode.flat.as_ptr(), self.w_decode.flat.as_ptr(), self.b_encode.flat.as_ptr(),
                                            self.b_decode.flat.as_ptr(), d_in, latent, std::ptr::null_mut(), 0, &mut n_bytes));
        }
        let mut encoded = vec![0u8; n_bytes as usize];
        unsafe {
            check(apd_autoencoder_serialize(self.w_encode.flat.as_ptr(), self.w_decode.flat.as_ptr(), self.b_encode.flat.as_ptr(),
                                            self.b_decode.flat.as_ptr(), d_in, latent, encoded.as_mut_ptr() as *mut _, n_bytes, &mut n_bytes));
        }
        let mut fp = File::create(file)?;
        fp.write_all(&encoded)?;
        Ok(())
    }

    /// neural.rs:55-71: z-score (sigma >= 1) of 255 * sigmoid(x W + b), one row.  The pipeline never calls this per frame any more
    /// (NDSequence::encoded hands the whole sequence to apd_encode); kept for callers that do.
    pub fn predict(&self, x: &Mat) -> Mat {
        let latent = self.n_latent();
        let rows = x.flat.len() / x.cols;
        let mut flat = vec![0f32; rows * latent];
        unsafe {
            check(apd_encode(feature_context(), x.flat.as_ptr(), rows as u64, x.cols as u32, self.w_encode.flat.as_ptr(),
                             self.b_encode.flat.as_ptr(), latent as u32, 0, flat.as_mut_ptr()));
        }
        Mat { flat, cols: latent }
    }
}

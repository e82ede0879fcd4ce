// Feature companions: AutoEncoder::predict (reference src/neural.rs:55-71) and the cepstrum branch of
// NDSequence::new (src/spectrogram.rs:31-80).
#include "apd_internal.h"

extern "C" int apd_encode(apd_context *ctx, const float *x, uint64_t t, uint32_t d_in, const float *w_encode,
                          const float *b_encode, uint32_t latent, int on_device, float *out)
{
    (void)ctx; (void)x; (void)t; (void)d_in; (void)w_encode; (void)b_encode; (void)latent; (void)on_device; (void)out;
    return APD_ERR_UNSUPPORTED;
}

extern "C" int apd_cepstrum(apd_context *ctx, const int16_t *samples, uint64_t n_samples, uint32_t fft_size,
                            uint32_t fft_step, uint32_t filter_size, int on_device, float *out, uint64_t *n_frames,
                            uint32_t *n_bins)
{
    (void)ctx; (void)samples; (void)n_samples; (void)fft_size; (void)fft_step; (void)filter_size; (void)on_device;
    (void)out; (void)n_frames; (void)n_bins;
    return APD_ERR_UNSUPPORTED;
}

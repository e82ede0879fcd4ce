// Feature companions on the GPU: AutoEncoder::predict (reference src/neural.rs:55-71) and the cepstrum branch of
// NDSequence::new (src/spectrogram.rs:31-80).  Both are streaming kernels far off the hot path's critical time
// (cfg 4: 4.2 M frames x 21 floats; cfg 5: 33.5 M frames of 256 samples); they exist so that the feature sequences
// the alignment consumes are produced in HBM and never cross PCIe.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "apd_internal.h"

using namespace apd;

namespace {

#define HIP_TRY(ctx, call)                                                             \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            (ctx)->last_error = std::string(#call) + ": " + hipGetErrorString(e_);     \
            return e_ == hipErrorOutOfMemory ? APD_ERR_OOM : APD_ERR_HIP;              \
        }                                                                              \
    } while (0)

// --------------------------------------------------------------------------------------- encoder
// One thread per frame.  Arithmetic order follows the reference: Mat::mul accumulates k ascending with a separate
// multiply and add (numerics.rs:310-316; this unit is built with -ffp-contract=off), add_col, sigmoid, scale(255),
// population mean / std over the latent values (numerics.rs:12-29), sigma floored at 1 (neural.rs:62), z-score.
__global__ __launch_bounds__(256) void encode_kernel(const float *__restrict__ x, uint64_t t, uint32_t d_in,
                                                     const float *__restrict__ w, const float *__restrict__ b,
                                                     uint32_t latent, float *__restrict__ out)
{
    extern __shared__ float wb[];                               // w[d_in][latent] then b[latent]
    for (uint32_t e = threadIdx.x; e < d_in * latent + latent; e += blockDim.x) wb[e] = e < d_in * latent ? w[e] : b[e - d_in * latent];
    __syncthreads();
    const float *bs = wb + d_in * latent;
    for (uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; f < t; f += (uint64_t)gridDim.x * blockDim.x) {
        const float *xf = x + f * d_in;
        float *pred = out + f * latent;
        float mean = 0.0f;
        for (uint32_t j = 0; j < latent; ++j) {
            float acc = 0.0f;
            for (uint32_t k = 0; k < d_in; ++k) acc = acc + xf[k] * wb[k * latent + j];
            acc = acc + bs[j];
            const float s = 1.0f / (1.0f + expf(-acc));        // numerics.rs:233
            const float v = s * 255.0f;
            pred[j] = v;
            mean = mean + v;
        }
        const float mu = mean / (float)latent;
        float sd = 0.0f;
        for (uint32_t j = 0; j < latent; ++j) { const float dv = pred[j] - mu; sd = sd + dv * dv; }
        const float sigma = fmaxf(sqrtf(sd / (float)latent), 1.0f);
        for (uint32_t j = 0; j < latent; ++j) pred[j] = (pred[j] - mu) / sigma;
    }
}

// -------------------------------------------------------------------------------------- cepstrum
// One wavefront per frame, four frames per workgroup.  Hamming window, radix-2 Stockham FFT of the N real samples in
// LDS (N a power of two), magnitudes of the first N/2 bins, triangular filterbank with stride L/2, ln(. + 1e-6),
// DCT-I as a K x K table product, drop 4, subtract the mean of what is left (spectrogram.rs:51-79).
struct CepsParams {
    const int16_t *samples;
    const uint64_t *sample_off;   // [n_seq+1] first sample of every recording
    const uint64_t *frame_off;    // [n_seq+1] first output frame of every recording
    uint32_t n_seq;
    uint64_t n_frames;
    uint64_t first_frame;      // frames first_frame .. of this launch (launches stay below 2^31 work-items)
    uint32_t fft, step, L, fstep, K, log2n;
    const float *hamming;      // [fft]
    const float *triag;        // [L]
    const float2 *twiddle;     // [fft/2]  exp(-2 pi i k / fft)
    const float *dct;          // [K][K]   DCT-I table incl. the 1/2 weights of the end points
    float *out;                // [n_frames][K-4]
};

__global__ __launch_bounds__(256) void cepstrum_kernel(const CepsParams P)
{
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t N = P.fft, half = N / 2;
    // per wave: two complex buffers of N (ping-pong), then mag[half], conv[K], ceps[K]
    float *base = lds + (size_t)wave * (4 * N + half + 2 * P.K);
    float2 *bufa = reinterpret_cast<float2 *>(base), *bufb = bufa + N;
    float *mag = base + 4 * N, *conv = mag + half, *ceps = conv + P.K;
    const uint64_t frame = P.first_frame + (uint64_t)blockIdx.x * 4 + wave;
    const bool live = frame < P.n_frames;
    uint32_t lo = 0, hi = P.n_seq;                                // recording holding this frame: largest s with frame_off[s] <= frame
    while (live && hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (P.frame_off[mid] <= frame) lo = mid; else hi = mid; }
    const uint64_t start = live ? P.sample_off[lo] + (frame - P.frame_off[lo]) * P.step : 0;   // i - fft_size, i = fft + t * step (:51-53)
    for (uint32_t s = lane; s < N; s += 64) {
        const float v = live ? (float)P.samples[start + s] * P.hamming[s] : 0.0f;   // :55-59
        bufa[s] = make_float2(v, 0.0f);
    }
    __syncthreads();
    float2 *src = bufa, *dst = bufb;
    for (uint32_t st = 0; st < P.log2n; ++st) {
        const uint32_t Ns = 1u << st;
        for (uint32_t j = lane; j < half; j += 64) {
            const uint32_t k = j & (Ns - 1);
            const float2 w = P.twiddle[k * (half / Ns)];           // exp(-2 pi i k / (2 Ns))
            const float2 p = src[j], q0 = src[j + half];
            const float2 q = make_float2(q0.x * w.x - q0.y * w.y, q0.x * w.y + q0.y * w.x);
            const uint32_t idx = ((j - k) << 1) + k;
            dst[idx] = make_float2(p.x + q.x, p.y + q.y);
            dst[idx + Ns] = make_float2(p.x - q.x, p.y - q.y);
        }
        __syncthreads();
        float2 *tmp = src; src = dst; dst = tmp;
    }
    for (uint32_t k = lane; k < half; k += 64) {
        const float2 v = src[k];
        const float nsq = v.x * v.x + v.y * v.y;                   // norm_sqr (:63)
        mag[k] = sqrtf(nsq);
    }
    __syncthreads();
    for (uint32_t c = lane; c < P.K; c += 64) {                    // convolve (numerics.rs:102-109)
        const uint32_t p = P.L + c * P.fstep;
        float dot = 0.0f;
        for (uint32_t q = 0; q < P.L; ++q) dot = dot + P.triag[q] * mag[p - P.L + q];
        conv[c] = logf(dot + 1e-6f);                               // :69
    }
    __syncthreads();
    for (uint32_t k = lane; k < P.K; k += 64) {                    // DCT-I (:71-73)
        float acc = 0.0f;
        for (uint32_t q = 0; q < P.K; ++q) acc = acc + P.dct[k * P.K + q] * conv[q];
        ceps[k] = acc;
    }
    __syncthreads();
    float mu = 0.0f;
    for (uint32_t k = 4; k < P.K; ++k) mu = mu + ceps[k];          // mean of cepstrum[4..] (:74, numerics.rs:12-18)
    mu = mu / (float)(P.K - 4);
    if (live)
        for (uint32_t k = 4 + lane; k < P.K; k += 64) P.out[frame * (P.K - 4) + (k - 4)] = ceps[k] - mu;   // :75-79
}

// ------------------------------------------------------------------------------------------ VAT
// NDSequence::variance (spectrogram.rs:174-187): per-frame population std (numerics.rs:12-29), then the mean of the k
// PREVIOUS values, summed in order.
__global__ void frame_std_kernel(const float *__restrict__ frames, uint64_t t, uint32_t n_bins, float *__restrict__ deltas)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < t; i += (uint64_t)gridDim.x * blockDim.x) {
        const float *v = frames + i * n_bins;
        float mean = 0.0f;
        for (uint32_t k = 0; k < n_bins; ++k) mean = mean + v[k];
        mean = mean / (float)n_bins;
        float sd = 0.0f;
        for (uint32_t k = 0; k < n_bins; ++k) { const float d = v[k] - mean; sd = sd + d * d; }
        deltas[i] = sqrtf(sd / (float)n_bins);
    }
}
__global__ void moving_mean_kernel(const float *__restrict__ deltas, uint64_t t, uint32_t k, float *__restrict__ out)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < t; i += (uint64_t)gridDim.x * blockDim.x) {
        float acc = 0.0f;
        if (i >= k) { for (uint32_t q = 0; q < k; ++q) acc = acc + deltas[i - k + q]; acc = acc / (float)k; }
        out[i] = acc;
    }
}

}  // namespace

// NDSequence::interesting_ranges (spectrogram.rs:192-216)
extern "C" int apd_interesting_ranges(apd_context *ctx, const float *frames, uint64_t t, uint32_t n_bins, uint32_t moving_average,
                                      float perc, uint64_t min_len, int on_device, uint64_t *ranges, uint64_t capacity,
                                      uint64_t *n_ranges)
{
    if (!ctx || !n_ranges || n_bins == 0 || (t && !frames) || (capacity && !ranges)) return APD_ERR_INVALID_ARG;
    *n_ranges = 0;
    const uint64_t kidx = apd::percentile_index(t, perc);
    if (t == 0 || kidx >= t) return APD_ERR_INDEX;                      // percentile of an empty / too short vector panics (numerics.rs:132)
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    char *pool = nullptr;
    const size_t in_bytes = on_device ? 0 : (size_t)t * n_bins * sizeof(float), v_bytes = ((size_t)t * sizeof(float) + 255) & ~(size_t)255;
    HIP_TRY(ctx, hipMalloc((void **)&pool, 2 * v_bytes + in_bytes + 256));
    float *d_deltas = (float *)pool, *d_var = (float *)(pool + v_bytes);
    const float *d_frames = on_device ? frames : (const float *)(pool + 2 * v_bytes);
    int rc = APD_OK;
    auto guard = [&](hipError_t e) { if (e != hipSuccess && rc == APD_OK) { ctx->last_error = hipGetErrorString(e); rc = APD_ERR_HIP; } };
    if (!on_device) guard(hipMemcpyAsync(pool + 2 * v_bytes, frames, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    const unsigned blocks = (unsigned)std::min<uint64_t>((t + 255) / 256, 8192);
    if (rc == APD_OK) {
        hipLaunchKernelGGL(frame_std_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_frames, t, n_bins, d_deltas);
        hipLaunchKernelGGL(moving_mean_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_deltas, t, moving_average, d_var);
        guard(hipGetLastError());
    }
    float th = 0.0f;
    if (rc == APD_OK) rc = apd::device_select(ctx, d_var, t, kidx, &th);                    // :198
    std::vector<float> var(t);
    if (rc == APD_OK) guard(hipMemcpyAsync(var.data(), d_var, (size_t)t * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    guard(hipStreamSynchronize(ctx->stream));
    hipFree(pool);
    if (rc != APD_OK) return rc;
    uint64_t start = 0, cnt = 0;
    bool recording = true;                                                                    // :202 (the scan starts "recording")
    for (uint64_t i = 0; i < t; ++i) {
        if (var[i] >= th && !recording) { start = i; recording = true; }                      // :204-207
        if (var[i] < th && recording) {                                                       // :208-213
            recording = false;
            if (i - start > min_len) { if (cnt < capacity) { ranges[2 * cnt] = start; ranges[2 * cnt + 1] = i; } ++cnt; }
        }
    }
    *n_ranges = cnt;
    return APD_OK;
}

extern "C" int apd_encode(apd_context *ctx, const float *x, uint64_t t, uint32_t d_in, const float *w_encode,
                          const float *b_encode, uint32_t latent, int on_device, float *out)
{
    if (!ctx || !w_encode || !b_encode || d_in == 0 || latent == 0 || (t && (!x || !out))) return APD_ERR_INVALID_ARG;
    const size_t wb_bytes = ((size_t)d_in * latent + latent) * sizeof(float);
    if (wb_bytes > 64 * 1024) return APD_ERR_UNSUPPORTED;
    if (t == 0) return APD_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    float *d_w = nullptr, *d_x = nullptr, *d_out = nullptr;
    HIP_TRY(ctx, hipMalloc((void **)&d_w, wb_bytes));
    int rc = APD_OK;
    auto guard = [&](hipError_t e) { if (e != hipSuccess && rc == APD_OK) { ctx->last_error = hipGetErrorString(e); rc = APD_ERR_HIP; } };
    guard(hipMemcpyAsync(d_w, w_encode, (size_t)d_in * latent * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    guard(hipMemcpyAsync(d_w + (size_t)d_in * latent, b_encode, latent * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    const float *xin = x;
    float *xout = out;
    if (!on_device && rc == APD_OK) {
        guard(hipMalloc((void **)&d_x, t * d_in * sizeof(float)));
        if (rc == APD_OK) guard(hipMalloc((void **)&d_out, t * latent * sizeof(float)));
        if (rc == APD_OK) guard(hipMemcpyAsync(d_x, x, t * d_in * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
        xin = d_x; xout = d_out;
    }
    if (rc == APD_OK) {
        const unsigned blocks = (unsigned)std::min<uint64_t>((t + 255) / 256, 8192);
        hipLaunchKernelGGL(encode_kernel, dim3(blocks), dim3(256), wb_bytes, ctx->stream, xin, t, d_in, d_w, d_w + (size_t)d_in * latent,
                           latent, xout);
        guard(hipGetLastError());
    }
    if (!on_device && rc == APD_OK) guard(hipMemcpyAsync(out, d_out, t * latent * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    guard(hipStreamSynchronize(ctx->stream));
    if (d_x) hipFree(d_x);
    if (d_out) hipFree(d_out);
    hipFree(d_w);
    return rc;
}

static int cepstrum_impl(apd_context *ctx, const int16_t *samples, const uint64_t *sample_off, uint32_t n_seq, uint32_t fft_size,
                         uint32_t fft_step, uint32_t filter_size, int on_device, float *out, uint64_t *frame_off, uint32_t *n_bins)
{
    if (!ctx || !sample_off || !frame_off || !n_bins || fft_size < 2 || fft_step == 0 || filter_size == 0) return APD_ERR_INVALID_ARG;
    const uint32_t L = fft_size / filter_size, half = fft_size / 2, fstep = L / 2;      // spectrogram.rs:38,64,67
    if (L == 0 || fstep == 0) return APD_ERR_INVALID_ARG;                               // step_by(0) panics in the reference
    uint32_t K = 0;
    for (uint32_t i = L; i < half; i += fstep) ++K;                                     // numerics.rs:105
    if (K < 5) return APD_ERR_INVALID_ARG;                                              // cepstrum[4..] of an empty tail
    frame_off[0] = 0;
    for (uint32_t s = 0; s < n_seq; ++s) {
        if (sample_off[s + 1] < sample_off[s]) return APD_ERR_INVALID_ARG;
        const uint64_t n = sample_off[s + 1] - sample_off[s];
        const uint64_t t = n > fft_size ? (n - fft_size + fft_step - 1) / fft_step : 0; // i in (fft_size..n).step_by(step), spectrogram.rs:51
        frame_off[s + 1] = frame_off[s] + t;
    }
    const uint64_t T = frame_off[n_seq], n_samples = sample_off[n_seq];
    *n_bins = K - 4;
    if (!out || T == 0) return APD_OK;
    if (!samples) return APD_ERR_INVALID_ARG;
    uint32_t log2n = 0;
    while ((1u << log2n) < fft_size) ++log2n;
    if ((1u << log2n) != fft_size || fft_size < 4 || fft_size > 2048 || K > 512) return APD_ERR_UNSUPPORTED;   // power-of-two windows only
    HIP_TRY(ctx, hipSetDevice(ctx->device));

    // tables, computed as the reference computes them
    std::vector<float> tab(fft_size + L + (size_t)K * K + fft_size);
    float *hamming = tab.data(), *triag = hamming + fft_size, *dct = triag + L;
    float2 *tw = reinterpret_cast<float2 *>(dct + (size_t)K * K);
    for (uint32_t i = 0; i < fft_size; ++i) {                                           // numerics.rs:60-66
        const float arg = (2.0f * 3.14159265358979323846f * (float)i) / (float)fft_size;
        hamming[i] = 0.54f + 0.46f * cosf(arg);
    }
    for (uint32_t i = 0; i < L; ++i) triag[i] = 0.0f;                                   // numerics.rs:78-86
    for (uint32_t i = 0; i <= (L - 1) / 2; ++i) { triag[i] = (float)i / (float)L; triag[L - 1 - i] = (float)i / (float)L; }
    for (uint32_t k = 0; k < K; ++k)                                                    // rustdct DCT-I definition
        for (uint32_t q = 0; q < K; ++q) {
            double c;
            if (q == 0) c = 0.5;
            else if (q == K - 1) c = (k & 1) ? -0.5 : 0.5;
            else c = std::cos(M_PI * (double)q * (double)k / (double)(K - 1));
            dct[(size_t)k * K + q] = (float)c;
        }
    for (uint32_t k = 0; k < half; ++k) {
        const double a = -2.0 * M_PI * (double)k / (double)fft_size;
        tw[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    const size_t tab_bytes = tab.size() * sizeof(float), off_bytes = 2 * ((size_t)n_seq + 1) * sizeof(uint64_t);
    char *pool = nullptr;
    const size_t in_bytes = on_device ? 0 : n_samples * sizeof(int16_t), out_bytes = on_device ? 0 : T * (K - 4) * sizeof(float);
    const size_t offs_off = (tab_bytes + 255) & ~(size_t)255, in_off = (offs_off + off_bytes + 255) & ~(size_t)255,
                 out_off = (in_off + in_bytes + 255) & ~(size_t)255;
    HIP_TRY(ctx, hipMalloc((void **)&pool, out_off + out_bytes + 256));
    int rc = APD_OK;
    auto guard = [&](hipError_t e) { if (e != hipSuccess && rc == APD_OK) { ctx->last_error = hipGetErrorString(e); rc = APD_ERR_HIP; } };
    guard(hipMemcpyAsync(pool, tab.data(), tab_bytes, hipMemcpyHostToDevice, ctx->stream));
    guard(hipMemcpyAsync(pool + offs_off, sample_off, off_bytes / 2, hipMemcpyHostToDevice, ctx->stream));
    guard(hipMemcpyAsync(pool + offs_off + off_bytes / 2, frame_off, off_bytes / 2, hipMemcpyHostToDevice, ctx->stream));
    if (!on_device) guard(hipMemcpyAsync(pool + in_off, samples, in_bytes, hipMemcpyHostToDevice, ctx->stream));
    CepsParams P{};
    P.samples = on_device ? samples : reinterpret_cast<const int16_t *>(pool + in_off);
    P.sample_off = reinterpret_cast<const uint64_t *>(pool + offs_off);
    P.frame_off = P.sample_off + n_seq + 1;
    P.n_seq = n_seq; P.n_frames = T; P.fft = fft_size; P.step = fft_step; P.L = L; P.fstep = fstep; P.K = K; P.log2n = log2n;
    const float *d_tab = reinterpret_cast<const float *>(pool);
    P.hamming = d_tab; P.triag = d_tab + fft_size; P.dct = d_tab + fft_size + L;
    P.twiddle = reinterpret_cast<const float2 *>(d_tab + fft_size + L + (size_t)K * K);
    P.out = on_device ? out : reinterpret_cast<float *>(pool + out_off);
    const size_t lds_bytes = 4 * (4 * (size_t)fft_size + half + 2 * K) * sizeof(float);
    if (lds_bytes > 160 * 1024) rc = APD_ERR_UNSUPPORTED;
    if (rc == APD_OK && lds_bytes > 64 * 1024)
        guard(hipFuncSetAttribute(reinterpret_cast<const void *>(cepstrum_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    if (rc == APD_OK) {
        constexpr uint64_t kFramesPerLaunch = 1ull << 24;                 // 2^22 workgroups of 256
        for (uint64_t f0 = 0; f0 < T && rc == APD_OK; f0 += kFramesPerLaunch) {
            P.first_frame = f0;
            const uint64_t cnt = std::min(kFramesPerLaunch, T - f0);
            hipLaunchKernelGGL(cepstrum_kernel, dim3((unsigned)((cnt + 3) / 4)), dim3(256), lds_bytes, ctx->stream, P);
            guard(hipGetLastError());
        }
    }
    if (!on_device && rc == APD_OK) guard(hipMemcpyAsync(out, pool + out_off, out_bytes, hipMemcpyDeviceToHost, ctx->stream));
    guard(hipStreamSynchronize(ctx->stream));
    hipFree(pool);
    return rc;
}

extern "C" int apd_cepstrum(apd_context *ctx, const int16_t *samples, uint64_t n_samples, uint32_t fft_size,
                            uint32_t fft_step, uint32_t filter_size, int on_device, float *out, uint64_t *n_frames,
                            uint32_t *n_bins)
{
    if (!n_frames) return APD_ERR_INVALID_ARG;
    const uint64_t sample_off[2] = {0, n_samples};
    uint64_t frame_off[2] = {0, 0};
    const int rc = cepstrum_impl(ctx, samples, sample_off, 1, fft_size, fft_step, filter_size, on_device, out, frame_off, n_bins);
    *n_frames = frame_off[1];
    return rc;
}

extern "C" int apd_cepstrum_batch(apd_context *ctx, const int16_t *samples, const uint64_t *sample_offsets, uint32_t n_seq,
                                  uint32_t fft_size, uint32_t fft_step, uint32_t filter_size, int on_device, float *out,
                                  uint64_t *frame_offsets, uint32_t *n_bins)
{
    return cepstrum_impl(ctx, samples, sample_offsets, n_seq, fft_size, fft_step, filter_size, on_device, out, frame_offsets, n_bins);
}

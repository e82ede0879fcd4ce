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
//
// The kernel is a stream (cfg 4: 218 MB in, 134 MB out, ~300 flops per frame), so what matters is how the bytes move: a
// wavefront takes 64 consecutive frames = one contiguous run of 64 * d_in floats, brings it into LDS with 16-byte coalesced
// loads, lets every lane read ITS frame from LDS (row stride padded to an odd number of floats: conflict-free), computes, and
// sends the 64 * latent results back through LDS as one contiguous run of 16-byte stores.  (The first version read frame
// rows straight from global memory, 52 bytes apart per lane: 13 load instructions touching 52 cache lines each, 4.8 % of
// the HBM rate.)
struct EncodeFrame {
    template <typename GetX, typename PutZ>
    __device__ static __forceinline__ void run(uint32_t d_in, uint32_t latent, const float *__restrict__ wb, GetX x, PutZ put, float *__restrict__ pred)
    {
        const float *bs = wb + d_in * latent;
        float mean = 0.0f;
        for (uint32_t j = 0; j < latent; ++j) {
            float acc = 0.0f;
            for (uint32_t k = 0; k < d_in; ++k) acc = acc + x(k) * wb[k * latent + j];
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
        for (uint32_t j = 0; j < latent; ++j) put(j, (pred[j] - mu) / sigma);
    }
};

__global__ __launch_bounds__(256) void encode_kernel(const float *__restrict__ x, uint64_t t, uint32_t d_in,
                                                     const float *__restrict__ w, const float *__restrict__ b,
                                                     uint32_t latent, float *__restrict__ out)
{
    extern __shared__ float wb[];                               // w[d_in][latent] then b[latent]
    for (uint32_t e = threadIdx.x; e < d_in * latent + latent; e += blockDim.x) wb[e] = e < d_in * latent ? w[e] : b[e - d_in * latent];
    __syncthreads();
    for (uint64_t f = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; f < t; f += (uint64_t)gridDim.x * blockDim.x) {
        const float *xf = x + f * d_in;
        float *pred = out + f * latent;                         // the un-normalised values are parked in the output row
        EncodeFrame::run(d_in, latent, wb, [&](uint32_t k) { return xf[k]; }, [&](uint32_t j, float v) { pred[j] = v; }, pred);
    }
}

// The staged form.  LDS: [w | b] then, per wavefront, 64 input rows of stride_in floats and 64 output rows of stride_out.
__global__ __launch_bounds__(256) void encode_staged_kernel(const float *__restrict__ x, uint64_t t, uint32_t d_in,
                                                            const float *__restrict__ w, const float *__restrict__ b,
                                                            uint32_t latent, float *__restrict__ out)
{
    extern __shared__ float lds[];
    const uint32_t n_wb = d_in * latent + latent, stride_in = d_in | 1u, stride_out = latent | 1u;
    float *wb = lds;
    for (uint32_t e = threadIdx.x; e < n_wb; e += blockDim.x) wb[e] = e < d_in * latent ? w[e] : b[e - d_in * latent];
    __syncthreads();
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float *xin = lds + ((n_wb + 3u) & ~3u) + wave * 64u * (stride_in + stride_out), *zout = xin + 64u * stride_in;
    const uint64_t n_chunks = (t + 63) / 64, waves = (uint64_t)gridDim.x * 4u;
    for (uint64_t chunk = (uint64_t)blockIdx.x * 4u + wave; chunk < n_chunks; chunk += waves) {
        const uint64_t f0 = chunk * 64;
        const uint32_t frames = (uint32_t)min<uint64_t>(64, t - f0), n_in = frames * d_in, n_out = frames * latent;
        const float *src = x + f0 * d_in;                       // 64 * d_in * 4 bytes per chunk: a multiple of 16, and hipMalloc aligns the base
        const bool vec_in = ((uintptr_t)src & 15u) == 0;
        for (uint32_t e = lane * 4u; e < n_in; e += 256u) {     // coalesced: 1 KB per wave instruction
            float v[4];
            if (vec_in && e + 4u <= n_in) { const float4 q = *reinterpret_cast<const float4 *>(src + e); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
            else for (uint32_t i = 0; i < 4; ++i) v[i] = e + i < n_in ? src[e + i] : 0.0f;
            uint32_t fr = e / d_in, k = e - fr * d_in;
            for (uint32_t i = 0; i < 4 && e + i < n_in; ++i) { xin[fr * stride_in + k] = v[i]; if (++k == d_in) { k = 0; ++fr; } }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // one wavefront: LDS runs in order, only the compiler must not reorder
        if (lane < frames) {
            const float *mine = xin + lane * stride_in;
            float *pred = zout + lane * stride_out;
            EncodeFrame::run(d_in, latent, wb, [&](uint32_t k) { return mine[k]; }, [&](uint32_t j, float v) { pred[j] = v; }, pred);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float *dst = out + f0 * latent;
        const bool vec_out = ((uintptr_t)dst & 15u) == 0;
        for (uint32_t e = lane * 4u; e < n_out; e += 256u) {
            float v[4];
            uint32_t fr = e / latent, j = e - fr * latent;
            for (uint32_t i = 0; i < 4; ++i) { v[i] = e + i < n_out ? zout[fr * stride_out + j] : 0.0f; if (++j == latent) { j = 0; ++fr; } }
            if (vec_out && e + 4u <= n_out) *reinterpret_cast<float4 *>(dst + e) = make_float4(v[0], v[1], v[2], v[3]);
            else for (uint32_t i = 0; i < 4 && e + i < n_out; ++i) dst[e + i] = v[i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the rows are reused by the next chunk
    }
}

// -------------------------------------------------------------------------------------- cepstrum
// One wavefront per frame, wavefronts loop over frames.  Hamming window, DFT of the N real samples, magnitudes of the
// first N/2 bins, triangular filterbank with stride L/2, ln(. + 1e-6), DCT-I as a K x K table product, drop 4, subtract
// the mean of what is left (spectrogram.rs:51-79).  The DFT is an autosort (Stockham) FFT in LDS, radix-4 passes, when N is a power of two and
// the defining sum X_k = sum_s x_s e^(-2 pi i k s / N) otherwise (rustfft plans any length, spectrogram.rs:44-48; the
// reference's shipped window is 256).  The tables live in LDS, loaded once per workgroup; the four wavefronts of a
// workgroup never exchange data, so the stages are ordered by the wavefront's own in-order LDS queue, not by barriers.
// Power-of-two windows: radix-4 passes, and each HALF of a wavefront (32 lanes) transforms a frame of its own.
struct CepsParams {
    const int16_t *samples;
    const uint64_t *sample_off;   // [n_seq+1] first sample of every recording
    const uint64_t *frame_off;    // [n_seq+1] first output frame of every recording
    uint32_t n_seq;
    uint64_t n_frames;
    uint32_t fft, step, L, fstep, K, log2n;   // log2n == 0: N is not a power of two, direct DFT
    uint32_t frames_per_wave;  // 2: each half of a wavefront transforms a frame of its own (power-of-two windows whose slots fit in LDS); else 1
    const float *hamming;      // [fft]
    const float *triag;        // [L]
    const float2 *twiddle;     // power of two: [fft/2] exp(-2 pi i k / fft); else [fft] exp(-2 pi i t / fft)
    const float *dct;          // [K][K]   DCT-I table incl. the 1/2 weights of the end points
    float *out;                // [n_frames][K-4]
};

#define APD_WAVE_LDS_FENCE() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

__device__ __forceinline__ float2 cmul(const float2 a, const float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// From the magnitudes of one frame (mag[0 .. N/2) in LDS) to its output row: triangular filterbank with stride L/2, ln(. + 1e-6),
// DCT-I as a K x K table product, drop 4, subtract the mean of what is left (spectrogram.rs:66-79).  Executed by `gsize`
// consecutive lanes (a whole wavefront, or one half of it), `gl` = lane within that group; the caller fences before and after.
__device__ __forceinline__ void finish_frame(const CepsParams &P, const float *t_tri, const float *t_dct, const float *mag, float *conv,
                                             float *ceps, uint32_t gl, uint32_t gsize, uint64_t frame, bool live)
{
    const uint32_t K = P.K;
    for (uint32_t c = gl; c < K; c += gsize) {                 // convolve (numerics.rs:102-109)
        const uint32_t p = P.L + c * P.fstep;
        float dot = 0.0f;
#pragma unroll 4
        for (uint32_t q = 0; q < P.L; ++q) dot = dot + t_tri[q] * mag[p - P.L + q];
        conv[c] = logf(dot + 1e-6f);                           // :69
    }
    APD_WAVE_LDS_FENCE();
    for (uint32_t k = gl; k < K; k += gsize) {                 // DCT-I (:71-73)
        float acc = 0.0f;
#pragma unroll 4
        for (uint32_t q = 0; q < K; ++q) acc = acc + t_dct[k * K + q] * conv[q];
        ceps[k] = acc;
    }
    APD_WAVE_LDS_FENCE();
    float mu = 0.0f;
#pragma unroll 4
    for (uint32_t k = 4; k < K; ++k) mu = mu + ceps[k];        // mean of cepstrum[4..] (:74, numerics.rs:12-18)
    mu = mu / (float)(K - 4);
    if (live)
        for (uint32_t k = 4 + gl; k < K; k += gsize) P.out[frame * (K - 4) + (k - 4)] = ceps[k] - mu;   // :75-79
}

__global__ __launch_bounds__(256) void cepstrum_kernel(const CepsParams P)
{
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t N = P.fft, half = N / 2, K = P.K, n_tw = P.log2n ? half : N;
    // tables, shared by the workgroup
    float *t_ham = lds, *t_tri = t_ham + N, *t_dct = t_tri + P.L;
    float2 *t_tw = reinterpret_cast<float2 *>(lds + ((N + P.L + K * K + 1u) & ~1u));        // 8-byte aligned
    for (uint32_t e = threadIdx.x; e < N; e += blockDim.x) t_ham[e] = P.hamming[e];
    for (uint32_t e = threadIdx.x; e < P.L; e += blockDim.x) t_tri[e] = P.triag[e];
    for (uint32_t e = threadIdx.x; e < K * K; e += blockDim.x) t_dct[e] = P.dct[e];
    for (uint32_t e = threadIdx.x; e < n_tw; e += blockDim.x) t_tw[e] = P.twiddle[e];
    __syncthreads();
    // per frame slot: two complex buffers of N/2 (ping-pong; the direct form uses the first as N real samples), mag[half], conv[K], ceps[K].
    // Power-of-two windows: TWO slots per wavefront -- each half of the wave (32 lanes) transforms a frame of its own; else one.
    const uint32_t slot_floats = (2 * N + half + 2 * K + 1u) & ~1u, slots = P.frames_per_wave;
    float *wave_base = reinterpret_cast<float *>(t_tw + n_tw) + (size_t)wave * slots * slot_floats;
    // A wavefront takes a CONTIGUOUS run of frames: consecutive frames of a recording overlap (dft_step < dft_win: their samples
    // come from L2 the second time) and the recording index only ever steps forward (one binary search per run, not per frame).
    const uint64_t n_waves = (uint64_t)gridDim.x * 4u, per_wave = (P.n_frames + n_waves - 1) / n_waves;
    const uint64_t f_begin = ((uint64_t)blockIdx.x * 4u + wave) * per_wave, f_end = min(f_begin + per_wave, P.n_frames);
    const uint32_t gsize = 64u / slots, gl = (uint32_t)lane % gsize, fs = (uint32_t)lane / gsize;   // lane group = frame slot
    float *base = wave_base + (size_t)fs * slot_floats;
    float2 *bufa = reinterpret_cast<float2 *>(base), *bufb = bufa + half;
    float *mag = base + 2 * N, *conv = mag + half, *ceps = conv + K;
    // recording holding this group's first frame: largest s with frame_off[s] <= frame
    uint32_t lo = 0;
    const uint64_t f_first = f_begin + fs;
    if (f_first < f_end) {
        uint32_t hi = P.n_seq;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (P.frame_off[mid] <= f_first) lo = mid; else hi = mid; }
    }
    uint64_t next_off = f_first < f_end ? P.frame_off[lo + 1] : 0, this_off = f_first < f_end ? P.frame_off[lo] : 0, samp0 = f_first < f_end ? P.sample_off[lo] : 0;
    for (uint64_t f0 = f_begin; f0 < f_end; f0 += slots) {
        const uint64_t frame = f0 + fs;
        const bool live = frame < f_end;                           // an odd run leaves the second half of the wave without a frame
        if (live)
            while (frame >= next_off) { ++lo; this_off = next_off; next_off = P.frame_off[lo + 1]; samp0 = P.sample_off[lo]; }   // recordings without frames are skipped
        const uint64_t start = live ? samp0 + (frame - this_off) * P.step : 0;   // i - fft_size, i = fft + t * step (:51-53)
        if (P.log2n) {
            // real input: the N samples are packed as M = N/2 complex values z[s] = x[2s] + i x[2s+1], one M-point FFT is
            // run (half the butterflies and half the LDS traffic of an N-point transform of zero-imaginary data), and the
            // N-point spectrum of the real signal follows from Z[k] and conj(Z[M-k]).  The M-point transform is an autosort
            // (Stockham) FFT in radix-4 passes with one radix-2 pass at the end when log2(M) is odd: M = 128 takes 4 LDS round
            // trips instead of 7, and a radix-4 pass of 32 butterflies is exactly one half-wave.
            const uint32_t M = half, m = P.log2n - 1u, n4 = m / 2u;
            for (uint32_t s = gl; s < M; s += gsize)
                bufa[s] = live ? make_float2((float)P.samples[start + 2 * s] * t_ham[2 * s], (float)P.samples[start + 2 * s + 1] * t_ham[2 * s + 1])
                               : make_float2(0.0f, 0.0f);           // :55-59
            APD_WAVE_LDS_FENCE();
            float2 *src = bufa, *dst = bufb;
            uint32_t Ns = 1;
            for (uint32_t st = 0; st < n4; ++st) {
                const uint32_t q = M / 4u;
                for (uint32_t j = gl; j < q; j += gsize) {
                    const uint32_t k = j & (Ns - 1u);
                    const float2 w1 = t_tw[k * (half / 2u / Ns)];   // exp(-2 pi i k / (4 Ns))
                    const float2 w2 = t_tw[k * (half / Ns)];        // its square
                    const float2 w3 = cmul(w1, w2);
                    const float2 v0 = src[j], v1 = cmul(src[j + q], w1), v2 = cmul(src[j + 2u * q], w2), v3 = cmul(src[j + 3u * q], w3);
                    const float2 a0 = make_float2(v0.x + v2.x, v0.y + v2.y), a1 = make_float2(v0.x - v2.x, v0.y - v2.y);
                    const float2 a2 = make_float2(v1.x + v3.x, v1.y + v3.y), a3 = make_float2(v1.y - v3.y, v3.x - v1.x);   // -i (v1 - v3)
                    const uint32_t idx = ((j - k) << 2) + k;
                    dst[idx] = make_float2(a0.x + a2.x, a0.y + a2.y);
                    dst[idx + Ns] = make_float2(a1.x + a3.x, a1.y + a3.y);
                    dst[idx + 2u * Ns] = make_float2(a0.x - a2.x, a0.y - a2.y);
                    dst[idx + 3u * Ns] = make_float2(a1.x - a3.x, a1.y - a3.y);
                }
                APD_WAVE_LDS_FENCE();
                float2 *tmp = src; src = dst; dst = tmp;
                Ns <<= 2;
            }
            if (m & 1u) {                                           // the last, radix-2 pass: Ns = M / 2
                const uint32_t hm = M / 2u;
                for (uint32_t j = gl; j < hm; j += gsize) {
                    const uint32_t k = j & (Ns - 1u);
                    const float2 p = src[j], qv = cmul(src[j + hm], t_tw[k * (half / Ns)]);   // exp(-2 pi i k / (2 Ns))
                    const uint32_t idx = ((j - k) << 1) + k;
                    dst[idx] = make_float2(p.x + qv.x, p.y + qv.y);
                    dst[idx + Ns] = make_float2(p.x - qv.x, p.y - qv.y);
                }
                APD_WAVE_LDS_FENCE();
                float2 *tmp = src; src = dst; dst = tmp;
            }
            for (uint32_t k = gl; k < M; k += gsize) {
                const float2 zk = src[k], zm = src[(M - k) & (M - 1)], w = t_tw[k];   // w = exp(-2 pi i k / N)
                const float ax = 0.5f * (zk.x + zm.x), ay = 0.5f * (zk.y - zm.y);     // (Z[k] + conj Z[M-k]) / 2
                const float bx = zk.x - zm.x, by = zk.y + zm.y;                       //  Z[k] - conj Z[M-k]
                const float xr = ax + 0.5f * (w.x * by + w.y * bx), xi = ay - 0.5f * (w.x * bx - w.y * by);
                mag[k] = sqrtf(xr * xr + xi * xi);                                    // norm_sqr().sqrt() (:63)
            }
        } else {
            float *win = base;                                     // N windowed samples
            for (uint32_t s = lane; s < N; s += 64) win[s] = (float)P.samples[start + s] * t_ham[s];
            APD_WAVE_LDS_FENCE();
            for (uint32_t k = lane; k < half; k += 64) {           // X_k = sum_s x_s (cos - i sin)(2 pi k s / N)
                float re = 0.0f, im = 0.0f;
                uint32_t idx = 0;                                  // (k * s) mod N
                for (uint32_t s = 0; s < N; ++s) {
                    const float2 w = t_tw[idx];
                    const float x = win[s];
                    re = fmaf(x, w.x, re);
                    im = fmaf(x, w.y, im);
                    idx += k;
                    if (idx >= N) idx -= N;
                }
                mag[k] = sqrtf(re * re + im * im);
            }
        }
        APD_WAVE_LDS_FENCE();
        finish_frame(P, t_tri, t_dct, mag, conv, ceps, gl, gsize, frame, live);
        APD_WAVE_LDS_FENCE();                                      // the buffers are reused by this lane group's next frame
    }
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
    HIP_TRY(ctx, apd::bind_device(ctx));
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

// ---- the feature stage as RESIDENT objects + enqueue-only calls -----------------------------------------------------------------
// The reference builds its features once per recording with par_iter (main.rs:150-161); a host that recomputes them per step (the
// benchmark's cfg 4 / cfg 5 step, or a pipeline over many corpora) calls these: the encoder weights / the cepstrum tables and
// offsets are uploaded ONCE into an object that lives on the context's GPU, and apd_encode_async / apd_cepstrum_batch_async only
// enqueue the kernel on the context's stream -- no allocation, no table building, no synchronisation.  With several GPUs every
// device runs the whole corpus' feature kernel concurrently (replicated, not sharded: the kernels take 0.15 ms (cfg 4) and 19 ms
// (cfg 5) for the WHOLE corpus, less than an all-gather of their 134 MB / 1.74 GB output over xGMI would).

struct apd_encoder {
    apd_context *ctx = nullptr;
    float *d_w = nullptr;             // [d_in][latent] weights, then [latent] bias
    uint32_t d_in = 0, latent = 0;
};

struct apd_cepstrum_plan {
    apd_context *ctx = nullptr;
    char *pool = nullptr;             // tables | sample offsets | frame offsets
    CepsParams P{};                   // samples / out filled per call
    size_t lds_bytes = 0;
    unsigned blocks = 0;
    uint64_t n_samples = 0;
};

namespace apd {
void orphan_encoder(apd_encoder *e) { if (e->d_w) hipFree(e->d_w); e->d_w = nullptr; e->ctx = nullptr; }
void orphan_cepstrum_plan(apd_cepstrum_plan *p) { if (p->pool) hipFree(p->pool); p->pool = nullptr; p->ctx = nullptr; }
}  // namespace apd

extern "C" int apd_encoder_create(apd_context *ctx, const float *w_encode, const float *b_encode, uint32_t d_in, uint32_t latent, apd_encoder **out)
{
    if (!ctx || !w_encode || !b_encode || d_in == 0 || latent == 0 || !out) return APD_ERR_INVALID_ARG;
    *out = nullptr;
    const size_t wb_bytes = ((size_t)d_in * latent + latent) * sizeof(float);
    if (wb_bytes > 64 * 1024) return APD_ERR_UNSUPPORTED;
    HIP_TRY(ctx, apd::bind_device(ctx));
    apd_encoder *e = new (std::nothrow) apd_encoder();
    if (!e) return APD_ERR_OOM;
    e->ctx = ctx; e->d_in = d_in; e->latent = latent;
    hipError_t err = hipMalloc((void **)&e->d_w, wb_bytes);
    if (err == hipSuccess) err = hipMemcpyAsync(e->d_w, w_encode, (size_t)d_in * latent * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (err == hipSuccess) err = hipMemcpyAsync(e->d_w + (size_t)d_in * latent, b_encode, latent * sizeof(float), hipMemcpyHostToDevice, ctx->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);      // the host arrays are the caller's: done with them on return
    if (err != hipSuccess) {
        ctx->last_error = std::string("apd_encoder_create: ") + hipGetErrorString(err);
        if (e->d_w) hipFree(e->d_w);
        delete e;
        return err == hipErrorOutOfMemory ? APD_ERR_OOM : APD_ERR_HIP;
    }
    ctx->encoders.insert(e);
    *out = e;
    return APD_OK;
}

extern "C" int apd_encoder_destroy(apd_encoder *e)
{
    if (!e) return APD_ERR_INVALID_ARG;
    if (e->ctx) {                                                         // else: orphaned by apd_destroy, device side already gone
        apd::bind_device(e->ctx);
        hipStreamSynchronize(e->ctx->stream);
        if (e->d_w) hipFree(e->d_w);
        e->ctx->encoders.erase(e);
    }
    delete e;
    return APD_OK;
}

extern "C" int apd_encode_async(apd_context *ctx, const apd_encoder *e, const float *d_x, uint64_t t, float *d_out)
{
    if (!ctx || !e || e->ctx != ctx || (t && (!d_x || !d_out))) return APD_ERR_INVALID_ARG;
    if (t == 0) return APD_OK;
    HIP_TRY(ctx, apd::bind_device(ctx));
    APD_AFFINITY(ctx, "encoder launch");
    const uint32_t d_in = e->d_in, latent = e->latent;
    const unsigned blocks = (unsigned)std::min<uint64_t>((t + 255) / 256, 8192);
    // staged through LDS when the 4 x 64 staged rows fit next to the weights (they do for every shape the reference produces)
    const size_t staged_bytes = (((size_t)d_in * latent + latent + 3) & ~(size_t)3) * sizeof(float) +
                                4 * 64 * (size_t)((d_in | 1u) + (latent | 1u)) * sizeof(float);
    if (staged_bytes <= 64 * 1024)
        hipLaunchKernelGGL(encode_staged_kernel, dim3(blocks), dim3(256), staged_bytes, ctx->stream, d_x, t, d_in, e->d_w,
                           e->d_w + (size_t)d_in * latent, latent, d_out);
    else
        hipLaunchKernelGGL(encode_kernel, dim3(blocks), dim3(256), ((size_t)d_in * latent + latent) * sizeof(float), ctx->stream, d_x, t, d_in, e->d_w,
                           e->d_w + (size_t)d_in * latent, latent, d_out);
    HIP_TRY(ctx, hipGetLastError());
    return APD_OK;
}

// The one-call form (host or device arrays): an encoder object made, used and dropped inside the call; blocking.
extern "C" int apd_encode(apd_context *ctx, const float *x, uint64_t t, uint32_t d_in, const float *w_encode,
                          const float *b_encode, uint32_t latent, int on_device, float *out)
{
    if (!ctx || !w_encode || !b_encode || d_in == 0 || latent == 0 || (t && (!x || !out))) return APD_ERR_INVALID_ARG;
    if (((size_t)d_in * latent + latent) * sizeof(float) > 64 * 1024) return APD_ERR_UNSUPPORTED;
    if (t == 0) return APD_OK;
    apd_encoder *e = nullptr;
    int rc = apd_encoder_create(ctx, w_encode, b_encode, d_in, latent, &e);
    if (rc != APD_OK) return rc;
    float *d_x = nullptr, *d_out = nullptr;
    auto guard = [&](hipError_t err) { if (err != hipSuccess && rc == APD_OK) { ctx->last_error = hipGetErrorString(err); rc = err == hipErrorOutOfMemory ? APD_ERR_OOM : APD_ERR_HIP; } };
    if (!on_device) {
        guard(hipMalloc((void **)&d_x, t * d_in * sizeof(float)));
        if (rc == APD_OK) guard(hipMalloc((void **)&d_out, t * latent * sizeof(float)));
        if (rc == APD_OK) guard(hipMemcpyAsync(d_x, x, t * d_in * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    }
    if (rc == APD_OK) rc = apd_encode_async(ctx, e, on_device ? x : d_x, t, on_device ? out : d_out);
    if (!on_device && rc == APD_OK) guard(hipMemcpyAsync(out, d_out, t * latent * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    guard(hipStreamSynchronize(ctx->stream));
    if (d_x) hipFree(d_x);
    if (d_out) hipFree(d_out);
    apd_encoder_destroy(e);
    return rc;
}

// Frame offsets and bin count of a corpus (spectrogram.rs:51, numerics.rs:105): pure host arithmetic.
static int cepstrum_geometry(const uint64_t *sample_off, uint32_t n_seq, uint32_t fft_size, uint32_t fft_step, uint32_t filter_size,
                             uint64_t *frame_off, uint32_t *n_bins, uint32_t *K_out)
{
    if (!sample_off || !frame_off || !n_bins || fft_size < 2 || fft_step == 0 || filter_size == 0) return APD_ERR_INVALID_ARG;
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
    *n_bins = K - 4;
    *K_out = K;
    return APD_OK;
}

extern "C" int apd_cepstrum_plan_create(apd_context *ctx, const uint64_t *sample_off, uint32_t n_seq, uint32_t fft_size, uint32_t fft_step,
                                        uint32_t filter_size, uint64_t *frame_off, uint32_t *n_bins, apd_cepstrum_plan **out)
{
    if (!ctx || !out) return APD_ERR_INVALID_ARG;
    *out = nullptr;
    uint32_t K = 0;
    int rc = cepstrum_geometry(sample_off, n_seq, fft_size, fft_step, filter_size, frame_off, n_bins, &K);
    if (rc != APD_OK) return rc;
    const uint32_t L = fft_size / filter_size, half = fft_size / 2, fstep = L / 2;
    const uint64_t T = frame_off[n_seq];
    uint32_t log2n = 0;
    while ((1u << log2n) < fft_size) ++log2n;
    if ((1u << log2n) != fft_size) log2n = 0;                                           // not a power of two: the defining sum
    if (fft_size < 4 || fft_size > 4096 || K > 512) return APD_ERR_UNSUPPORTED;
    const uint32_t n_tw = log2n ? half : fft_size;
    HIP_TRY(ctx, apd::bind_device(ctx));

    // tables, computed as the reference computes them
    const size_t tw_off = ((size_t)fft_size + L + (size_t)K * K + 1) & ~(size_t)1;      // float2 table: 8-byte aligned
    std::vector<float> tab(tw_off + 2 * (size_t)n_tw);
    float *hamming = tab.data(), *triag = hamming + fft_size, *dct = triag + L;
    float2 *tw = reinterpret_cast<float2 *>(tab.data() + tw_off);
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
    for (uint32_t k = 0; k < n_tw; ++k) {
        const double a = -2.0 * M_PI * (double)k / (double)fft_size;
        tw[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    apd_cepstrum_plan *plan = new (std::nothrow) apd_cepstrum_plan();
    if (!plan) return APD_ERR_OOM;
    plan->ctx = ctx;
    plan->n_samples = sample_off[n_seq];
    const size_t tab_bytes = tab.size() * sizeof(float), off_bytes = 2 * ((size_t)n_seq + 1) * sizeof(uint64_t);
    const size_t offs_off = (tab_bytes + 255) & ~(size_t)255;
    hipError_t err = hipMalloc((void **)&plan->pool, offs_off + off_bytes + 256);
    if (err == hipSuccess) err = hipMemcpyAsync(plan->pool, tab.data(), tab_bytes, hipMemcpyHostToDevice, ctx->stream);
    if (err == hipSuccess) err = hipMemcpyAsync(plan->pool + offs_off, sample_off, off_bytes / 2, hipMemcpyHostToDevice, ctx->stream);
    if (err == hipSuccess) err = hipMemcpyAsync(plan->pool + offs_off + off_bytes / 2, frame_off, off_bytes / 2, hipMemcpyHostToDevice, ctx->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);                     // `tab` and the caller's offsets are free after this
    CepsParams &P = plan->P;
    P.sample_off = reinterpret_cast<const uint64_t *>(plan->pool + offs_off);
    P.frame_off = P.sample_off + n_seq + 1;
    P.n_seq = n_seq; P.n_frames = T; P.fft = fft_size; P.step = fft_step; P.L = L; P.fstep = fstep; P.K = K; P.log2n = log2n;
    const float *d_tab = reinterpret_cast<const float *>(plan->pool);
    P.hamming = d_tab; P.triag = d_tab + fft_size; P.dct = d_tab + fft_size + L;
    P.twiddle = reinterpret_cast<const float2 *>(d_tab + tw_off);
    const size_t table_floats = tw_off + 2 * (size_t)n_tw;                               // the kernel lays its LDS out the same way
    const size_t slot_floats = (2 * (size_t)fft_size + half + 2 * K + 1) & ~(size_t)1;
    // two frames per wavefront for power-of-two windows, if eight slots fit beside the tables (windows up to 1024 do)
    P.frames_per_wave = (log2n && (table_floats + 8 * slot_floats) * sizeof(float) <= 96 * 1024) ? 2u : 1u;
    plan->lds_bytes = (table_floats + 4 * P.frames_per_wave * slot_floats) * sizeof(float);
    // wavefronts loop over frames: enough workgroups to fill the GPU several times over, tables loaded once per workgroup
    plan->blocks = (unsigned)std::min<uint64_t>((T + 4 * P.frames_per_wave - 1) / (4 * P.frames_per_wave), 256 * 16);
    int status = APD_OK;
    if (err != hipSuccess) { ctx->last_error = std::string("apd_cepstrum_plan_create: ") + hipGetErrorString(err); status = err == hipErrorOutOfMemory ? APD_ERR_OOM : APD_ERR_HIP; }
    else if (plan->lds_bytes > 160 * 1024) status = APD_ERR_UNSUPPORTED;
    else if (plan->lds_bytes > 64 * 1024 &&
             hipFuncSetAttribute(reinterpret_cast<const void *>(cepstrum_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plan->lds_bytes) != hipSuccess) {
        ctx->last_error = "apd_cepstrum_plan_create: the runtime refused the kernel its LDS";
        status = APD_ERR_HIP;
    }
    if (status != APD_OK) { if (plan->pool) hipFree(plan->pool); delete plan; return status; }
    ctx->cepstrum_plans.insert(plan);
    *out = plan;
    return APD_OK;
}

extern "C" int apd_cepstrum_plan_destroy(apd_cepstrum_plan *plan)
{
    if (!plan) return APD_ERR_INVALID_ARG;
    if (plan->ctx) {
        apd::bind_device(plan->ctx);
        hipStreamSynchronize(plan->ctx->stream);
        if (plan->pool) hipFree(plan->pool);
        plan->ctx->cepstrum_plans.erase(plan);
    }
    delete plan;
    return APD_OK;
}

extern "C" int apd_cepstrum_batch_async(apd_context *ctx, const apd_cepstrum_plan *plan, const int16_t *d_samples, float *d_out)
{
    if (!ctx || !plan || plan->ctx != ctx) return APD_ERR_INVALID_ARG;
    if (plan->P.n_frames == 0) return APD_OK;
    if (!d_samples || !d_out) return APD_ERR_INVALID_ARG;
    HIP_TRY(ctx, apd::bind_device(ctx));
    APD_AFFINITY(ctx, "cepstrum launch");
    CepsParams P = plan->P;
    P.samples = d_samples; P.out = d_out;
    hipLaunchKernelGGL(cepstrum_kernel, dim3(plan->blocks), dim3(256), plan->lds_bytes, ctx->stream, P);
    HIP_TRY(ctx, hipGetLastError());
    return APD_OK;
}

// The one-call forms: a plan made, used and dropped inside the call; host or device arrays; blocking.
static int cepstrum_impl(apd_context *ctx, const int16_t *samples, const uint64_t *sample_off, uint32_t n_seq, uint32_t fft_size,
                         uint32_t fft_step, uint32_t filter_size, int on_device, float *out, uint64_t *frame_off, uint32_t *n_bins)
{
    if (!ctx) return APD_ERR_INVALID_ARG;
    uint32_t K = 0;
    int rc = cepstrum_geometry(sample_off, n_seq, fft_size, fft_step, filter_size, frame_off, n_bins, &K);
    if (rc != APD_OK) return rc;
    const uint64_t T = frame_off[n_seq], n_samples = sample_off[n_seq];
    if (!out || T == 0) return APD_OK;
    if (!samples) return APD_ERR_INVALID_ARG;
    apd_cepstrum_plan *plan = nullptr;
    rc = apd_cepstrum_plan_create(ctx, sample_off, n_seq, fft_size, fft_step, filter_size, frame_off, n_bins, &plan);
    if (rc != APD_OK) return rc;
    int16_t *d_in = nullptr;
    float *d_o = nullptr;
    auto guard = [&](hipError_t e) { if (e != hipSuccess && rc == APD_OK) { ctx->last_error = hipGetErrorString(e); rc = e == hipErrorOutOfMemory ? APD_ERR_OOM : APD_ERR_HIP; } };
    if (!on_device) {
        guard(hipMalloc((void **)&d_in, n_samples * sizeof(int16_t)));
        if (rc == APD_OK) guard(hipMalloc((void **)&d_o, T * (K - 4) * sizeof(float)));
        if (rc == APD_OK) guard(hipMemcpyAsync(d_in, samples, n_samples * sizeof(int16_t), hipMemcpyHostToDevice, ctx->stream));
    }
    if (rc == APD_OK) rc = apd_cepstrum_batch_async(ctx, plan, on_device ? samples : d_in, on_device ? out : d_o);
    if (!on_device && rc == APD_OK) guard(hipMemcpyAsync(out, d_o, T * (K - 4) * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    guard(hipStreamSynchronize(ctx->stream));
    if (d_in) hipFree(d_in);
    if (d_o) hipFree(d_o);
    apd_cepstrum_plan_destroy(plan);
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

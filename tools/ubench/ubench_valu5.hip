// Micro-benchmark 5 (a measured ASIDE, see DESIGN.md: the north star rules MFMA out for this path): does the f32 matrix pipe
// issue beside a saturated vector pipe on gfx950?  v_mfma_f32_16x16x4_f32 runs at the f32 vector rate (MI355X_MICROARCH.md),
// so moving the 13-term dot products of the local distances onto it could only pay through CO-issue.  Modes: a VALU-only
// stream of v_fmac, the same stream with one MFMA per 24 / 12 / 6 VALU instructions, and MFMA only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define REP4(x) x x x x
#define FMAC6 "v_fmac_f32 %0, %1, %2\n v_fmac_f32 %3, %1, %2\n v_fmac_f32 %0, %2, %1\n v_fmac_f32 %3, %2, %1\n v_fmac_f32 %0, %1, %1\n v_fmac_f32 %3, %2, %2\n"
template <int VALU6_PER_MFMA, bool MFMA>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    float a = threadIdx.x, b = 1.0f + threadIdx.x * 1e-6f, c = 0.999f, d = 0.5f;
    f4 acc0 = {0, 0, 0, 0}, acc1 = {1, 1, 1, 1};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int v = 0; v < VALU6_PER_MFMA; ++v) asm volatile(FMAC6 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
            if (MFMA) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, c, acc0, 0, 0, 0);
                asm volatile("" : "+v"(acc0));
                if (VALU6_PER_MFMA == 0) { acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c, b, acc1, 0, 0, 0); asm volatile("" : "+v"(acc1)); }
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + acc0.x + acc0.y + acc0.z + acc0.w + acc1.x;
}
template <int V, bool M>
double run(float *d, int w, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<V, M>), dim3(256 * w), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<V, M>), dim3(256 * w), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms * 1e-3 / ((double)iters * 4) * 2.4e9 / w;      // SIMD cycles (2.4 GHz assumed) per inner group, per wave slot
}
int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 4}) {
        printf("waves/SIMD=%d: SIMD cycles (2.4 GHz assumed) per group\n", w);
        printf("  24 v_fmac                 %.1f\n", run<4, false>(d, w, 4000));
        printf("  24 v_fmac + 1 mfma16x16x4 %.1f\n", run<4, true>(d, w, 4000));
        printf("  12 v_fmac                 %.1f\n", run<2, false>(d, w, 4000));
        printf("  12 v_fmac + 1 mfma16x16x4 %.1f\n", run<2, true>(d, w, 4000));
        printf("   6 v_fmac                 %.1f\n", run<1, false>(d, w, 4000));
        printf("   6 v_fmac + 1 mfma16x16x4 %.1f\n", run<1, true>(d, w, 4000));
        printf("   2 mfma16x16x4 (independent) %.1f\n", run<0, true>(d, w, 4000));
    }
    return 0;
}

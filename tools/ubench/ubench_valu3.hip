// Micro-benchmark 3: compare/select encodings (VOPC/VOP2 via VCC vs VOP3 via SGPR pairs) on gfx950, inline asm so
// that the encodings are exactly the ones named.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    float a = threadIdx.x, b = threadIdx.x * 2.0f, c = 1.0f, d = 3.0f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) asm volatile(REP16("v_add_f32 %0, %0, %1\n v_add_f32 %2, %2, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (MODE == 1) asm volatile(REP16("v_cmp_eq_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "vcc");
        if (MODE == 2) asm volatile(REP16("v_cmp_eq_f32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s20", "s21");
        if (MODE == 3) asm volatile(REP16("v_cmp_eq_f32_e64 s[20:21], %0, %1\n s_or_b64 s[22:23], s[20:21], s[24:25]\n v_cndmask_b32_e64 %2, %2, %3, s[22:23]\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s20", "s21", "s22", "s23", "scc");
        if (MODE == 4) asm volatile(REP16("v_min3_f32 %0, %0, %1, %2\n v_min3_f32 %3, %3, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (MODE == 5) asm volatile(REP16("v_min_f32 %0, %0, %1\n v_min_f32 %0, %0, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (MODE == 6) asm volatile(REP16("v_cmp_eq_f32_e64 s[20:21], %0, %1\n v_cmp_lt_f32_e64 s[22:23], %2, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : : "s20", "s21", "s22", "s23");
        if (MODE == 7) asm volatile(REP16("v_cndmask_b32_e64 %0, %0, %1, s[20:21]\n v_cndmask_b32_e64 %2, %2, %3, s[20:21]\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (MODE == 8) asm volatile(REP16("v_fmac_f32 %0, %1, %2\n v_fmac_f32 %3, %1, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (MODE == 9) asm volatile(REP16("v_fma_f32 %0, %1, %2, %3\n v_fma_f32 %3, %1, %2, %0\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d;
}
template <int MODE>
double run(float *d, int w, int iters, int valu_per_rep)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * w), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * w), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms * 1e-3 / ((double)iters * 16 * valu_per_rep) * 2.4e9 / w;
}
int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    const char *names[] = {"v_add x2", "cmp_e32(vcc)+cndmask_e32", "cmp_e64+cndmask_e64", "cmp_e64+s_or+cndmask_e64", "v_min3 x2", "v_min x2 (dependent)",
                           "v_cmp_e64 x2", "v_cndmask_e64 x2", "v_fmac (VOP2) x2", "v_fma (VOP3) x2"};
    for (int w : {2, 4}) {
        printf("waves/SIMD=%d: SIMD cycles per VALU instruction (2.4 GHz assumed)\n", w);
        double t[10];
        t[0] = run<0>(d, w, 4000, 2); t[1] = run<1>(d, w, 4000, 2); t[2] = run<2>(d, w, 4000, 2); t[3] = run<3>(d, w, 4000, 2);
        t[4] = run<4>(d, w, 4000, 2); t[5] = run<5>(d, w, 4000, 2); t[6] = run<6>(d, w, 4000, 2); t[7] = run<7>(d, w, 4000, 2);
        t[8] = run<8>(d, w, 4000, 2); t[9] = run<9>(d, w, 4000, 2);
        for (int m = 0; m < 10; ++m) printf("  %-28s %.2f\n", names[m], t[m]);
    }
    return 0;
}

// Micro-benchmark 4: packed f32 (VOP3P) issue cost on gfx950 next to the scalar-lane forms, inline asm.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(x) x x x x x x x x x x x x x x x x
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    f2 a = {(float)threadIdx.x, 1.0f}, b = {threadIdx.x * 2.0f, 0.5f}, c = {1.0f, 2.0f}, d = {3.0f, 4.0f};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) asm volatile(REP16("v_fmac_f32 %0, %1, %2\n v_fmac_f32 %3, %1, %2\n") : "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(d.x));
        if (MODE == 1) asm volatile(REP16("v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %3, %1, %2, %3\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (MODE == 2) asm volatile(REP16("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %3, %3, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (MODE == 3) asm volatile(REP16("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %3, %3, %2\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (MODE == 4) asm volatile(REP16("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n v_pk_fma_f32 %3, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (MODE == 5) asm volatile(REP16("v_pk_fma_f32 %0, %1, %2, %0\n v_fmac_f32 %3, %4, %5\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d.x), "+v"(d.y), "+v"(b.x));
        if (MODE == 6) asm volatile(REP16("v_pk_fma_f32 %0, %1, %2, %0 neg_lo:[0,1,0] neg_hi:[0,1,0]\n v_pk_fma_f32 %3, %1, %2, %3 neg_lo:[0,1,0] neg_hi:[0,1,0]\n") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
        if (MODE == 7) asm volatile(REP16("v_pk_min_f16 %0, %0, %1\n v_pk_min_f16 %2, %2, %3\n") : "+v"(a.x), "+v"(b.x), "+v"(c.x), "+v"(d.x));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a.x + b.x + c.x + d.x + a.y + b.y + c.y + d.y;
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
    const char *names[] = {"v_fmac_f32 x2", "v_pk_fma_f32 x2", "v_pk_add_f32 x2", "v_pk_mul_f32 x2", "v_pk_fma_f32 op_sel bcast x2", "v_pk_fma + v_fmac", "v_pk_fma_f32 neg x2", "v_pk_min_f16 x2"};
    for (int w : {1, 2, 4}) {
        printf("waves/SIMD=%d: SIMD cycles per VALU instruction (2.4 GHz assumed)\n", w);
        double t[8];
        t[0] = run<0>(d, w, 4000, 2); t[1] = run<1>(d, w, 4000, 2); t[2] = run<2>(d, w, 4000, 2); t[3] = run<3>(d, w, 4000, 2);
        t[4] = run<4>(d, w, 4000, 2); t[5] = run<5>(d, w, 4000, 2); t[6] = run<6>(d, w, 4000, 2); t[7] = run<7>(d, w, 4000, 2);
        for (int m = 0; m < 8; ++m) printf("  %-32s %.2f\n", names[m], t[m]);
    }
    return 0;
}

// Micro-benchmark 6: the DTW node update's tie rule  r = (l == u) ? m : min3(l, u, m);  r += d  as
//   mode 0: v_min3 + v_cmp_eq + v_cndmask + v_add                (what the compiler emits for the C++ form)
//   mode 1: v_cmpx_neq + v_min3 (masked, in place on m) + s_mov exec + v_add
//   mode 2: mode 1 with the EXEC mask saved inside every block (s_mov tmp, exec first)
//   mode 3: mode 1 followed by s_nop 0
// two independent chains (the two DPs of a fused pair), 8 cells per repetition.  SIMD cycles per cell PAIR.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
#define SEL0 "v_min3_f32 %4, %0, %2, %1\n v_cmp_eq_f32 vcc, %0, %2\n v_cndmask_b32 %4, %4, %1, vcc\n v_add_f32 %0, %4, %3\n" \
             "v_min3_f32 %9, %5, %7, %6\n v_cmp_eq_f32 vcc, %5, %7\n v_cndmask_b32 %9, %9, %6, vcc\n v_add_f32 %5, %9, %8\n"
#define SEL1 "v_cmpx_neq_f32 vcc, %0, %2\n v_min3_f32 %1, %0, %2, %1\n s_mov_b64 exec, %10\n v_add_f32 %0, %1, %3\n" \
             "v_cmpx_neq_f32 vcc, %5, %7\n v_min3_f32 %6, %5, %7, %6\n s_mov_b64 exec, %10\n v_add_f32 %5, %6, %8\n"
#define SEL2 "s_mov_b64 %11, exec\n v_cmpx_neq_f32 vcc, %0, %2\n v_min3_f32 %1, %0, %2, %1\n s_mov_b64 exec, %11\n v_add_f32 %0, %1, %3\n" \
             "s_mov_b64 %11, exec\n v_cmpx_neq_f32 vcc, %5, %7\n v_min3_f32 %6, %5, %7, %6\n s_mov_b64 exec, %11\n v_add_f32 %5, %6, %8\n"
#define SEL3 "v_cmpx_neq_f32 vcc, %0, %2\n v_min3_f32 %1, %0, %2, %1\n s_mov_b64 exec, %10\n v_add_f32 %0, %1, %3\n s_nop 0\n" \
             "v_cmpx_neq_f32 vcc, %5, %7\n v_min3_f32 %6, %5, %7, %6\n s_mov_b64 exec, %10\n v_add_f32 %5, %6, %8\n s_nop 0\n"
// mode 4: the tie rule in integer arithmetic, no lane mask: x = (l ^ u) - 1 is negative iff l == u (non-negative floats);
//         k = min_i32(x, 0) is all ones iff tie; r = min(min(l, u) | k, m) -- the OR makes a NaN, which v_min drops; r += d
#define SEL4 "v_xad_u32 %4, %0, %2, -1\n v_min_f32 %0, %0, %2\n v_min_i32 %4, %4, 0\n v_or_b32 %0, %0, %4\n v_min_f32 %0, %0, %1\n v_add_f32 %0, %0, %3\n" \
             "v_xad_u32 %9, %5, %7, -1\n v_min_f32 %5, %5, %7\n v_min_i32 %9, %9, 0\n v_or_b32 %5, %5, %9\n v_min_f32 %5, %5, %6\n v_add_f32 %5, %5, %8\n"
// mode 5: the same with a guard value folded into the integer minimum (k = min3_i32(x, 0, g))
#define SEL5 "v_xad_u32 %4, %0, %2, -1\n v_min_f32 %0, %0, %2\n v_min3_i32 %4, %4, 0, %1\n v_or_b32 %0, %0, %4\n v_min_f32 %0, %0, %1\n v_add_f32 %0, %0, %3\n" \
             "v_xad_u32 %9, %5, %7, -1\n v_min_f32 %5, %5, %7\n v_min3_i32 %9, %9, 0, %6\n v_or_b32 %5, %5, %9\n v_min_f32 %5, %5, %6\n v_add_f32 %5, %5, %8\n"
// mode 6: no lane mask, no integer detour: t = min(l, u); e = l - u (0 iff tie); T = fma(|e|, -BIG, m) (m on a tie, hugely
//         negative otherwise); r = med3(t, m, T) (m on a tie, min(t, m) otherwise); r += d.  %10/%11 hold -BIG in VGPRs.
#define SEL6 "v_min_f32 %4, %0, %2\n v_sub_f32 %0, %0, %2\n v_fma_f32 %0, |%0|, %10, %1\n v_med3_f32 %4, %4, %1, %0\n v_add_f32 %0, %4, %3\n" \
             "v_min_f32 %9, %5, %7\n v_sub_f32 %5, %5, %7\n v_fma_f32 %5, |%5|, %10, %6\n v_med3_f32 %9, %9, %6, %5\n v_add_f32 %5, %9, %8\n"
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    float l1 = threadIdx.x & 3, m1 = 1.5f, u1 = (threadIdx.x >> 1) & 3, d1 = 0.25f, t1 = 0;
    float l2 = threadIdx.x & 1, m2 = 2.5f, u2 = (threadIdx.x >> 2) & 1, d2 = 0.125f, t2 = 0;
    const uint64_t ex = __builtin_amdgcn_read_exec();
    uint64_t tmp = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) asm volatile(REP8(SEL0) : "+v"(l1), "+v"(m1), "+v"(u1), "+v"(d1), "+v"(t1), "+v"(l2), "+v"(m2), "+v"(u2), "+v"(d2), "+v"(t2) : "s"(ex) : "vcc");
        if (MODE == 1) asm volatile(REP8(SEL1) : "+v"(l1), "+v"(m1), "+v"(u1), "+v"(d1), "+v"(t1), "+v"(l2), "+v"(m2), "+v"(u2), "+v"(d2), "+v"(t2) : "s"(ex) : "vcc");
        if (MODE == 2) asm volatile(REP8(SEL2) : "+v"(l1), "+v"(m1), "+v"(u1), "+v"(d1), "+v"(t1), "+v"(l2), "+v"(m2), "+v"(u2), "+v"(d2), "+v"(t2), "+s"(tmp) : "s"(ex) : "vcc");
        if (MODE == 4) asm volatile(REP8(SEL4) : "+v"(l1), "+v"(m1), "+v"(u1), "+v"(d1), "+v"(t1), "+v"(l2), "+v"(m2), "+v"(u2), "+v"(d2), "+v"(t2) : "s"(ex) : "vcc");
        if (MODE == 5) asm volatile(REP8(SEL5) : "+v"(l1), "+v"(m1), "+v"(u1), "+v"(d1), "+v"(t1), "+v"(l2), "+v"(m2), "+v"(u2), "+v"(d2), "+v"(t2) : "s"(ex) : "vcc");
        if (MODE == 6) { float nb = -3.0e38f; asm volatile(REP8(SEL6) : "+v"(l1), "+v"(m1), "+v"(u1), "+v"(d1), "+v"(t1), "+v"(l2), "+v"(m2), "+v"(u2), "+v"(d2), "+v"(t2) : "v"(nb)); }
        if (MODE == 3) asm volatile(REP8(SEL3) : "+v"(l1), "+v"(m1), "+v"(u1), "+v"(d1), "+v"(t1), "+v"(l2), "+v"(m2), "+v"(u2), "+v"(d2), "+v"(t2) : "s"(ex) : "vcc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = l1 + m1 + u1 + t1 + l2 + m2 + u2 + t2 + (float)tmp;
}
template <int MODE>
double run(float *d, int w, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<MODE>), dim3(256 * w), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL((k<MODE>), dim3(256 * w), dim3(256), 0, 0, d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms * 1e-3 / ((double)iters * 8) * 2.4e9 / w;      // SIMD cycles (2.4 GHz assumed) per cell pair, per wave slot
}
int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 4}) {
        printf("waves/SIMD=%d: SIMD cycles (2.4 GHz assumed) per cell pair (two node updates)\n", w);
        printf("  min3 + cmp_eq + cndmask + add          %.1f\n", run<0>(d, w, 20000));
        printf("  cmpx_neq + masked min3 + s_mov + add   %.1f\n", run<1>(d, w, 20000));
        printf("  ... with exec saved in every block     %.1f\n", run<2>(d, w, 20000));
        printf("  ... mode 1 + s_nop 0                   %.1f\n", run<3>(d, w, 20000));
        printf("  integer form: xad, min, min_i32, or, min, add  %.1f\n", run<4>(d, w, 20000));
        printf("  integer form with a guard (min3_i32)   %.1f\n", run<5>(d, w, 20000));
        printf("  min + sub + fma(|e|,-BIG,m) + med3 + add  %.1f\n", run<6>(d, w, 20000));
    }
    return 0;
}

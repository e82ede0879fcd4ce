// Micro-benchmark 2: the exact instruction forms of the DTW inner loop (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_REG 16
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters, float pen)
{
    float r[N_REG], t[N_REG];
    for (int q = 0; q < N_REG; ++q) { r[q] = (float)(threadIdx.x + q); t[q] = (float)(threadIdx.x * 3 + q); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int q = 0; q < N_REG; ++q) {
                if (MODE == 0) r[q] = r[q] + t[q];                                        // v_add 2 src
                if (MODE == 1) r[q] = __builtin_fmaf(t[q], t[q], r[q]);                   // v_fmac acc, t, t
                if (MODE == 2) r[q] = __builtin_fmaf(t[q], t[(q + 1) % N_REG], r[q]);     // v_fmac acc, a, b (3 distinct)
                if (MODE == 3) r[q] = __builtin_fmaf(pen, t[q], r[q]);                    // v_fmac acc, sgpr, t
                if (MODE == 4) r[q] = (t[q] == t[(q + 1) % N_REG]) ? r[q] : r[(q + 1) % N_REG];   // cmp + cndmask
                if (MODE == 5) r[q] = fminf(r[q], t[q]);                                  // v_min 2 src
                if (MODE == 6) r[q] = t[q] - t[(q + 1) % N_REG];                          // v_sub into other reg
                if (MODE == 7) { float d = t[q] - r[q]; r[q] = __builtin_fmaf(d, d, r[(q + 1) % N_REG]); }  // sub+fma (non-accumulating form)
            }
        }
    }
    float s = 0;
    for (int q = 0; q < N_REG; ++q) s += r[q] + t[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
double run(float *d, int w, int iters, int per)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * w), dim3(256), 0, 0, d, 10, 1.5f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * w), dim3(256), 0, 0, d, iters, 1.5f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms * 1e-3 / ((double)iters * 4 * N_REG * per) * 2.4e9 / w;
}
int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    const char *names[] = {"v_add a+=b", "v_fmac acc,t,t", "v_fmac acc,a,b", "v_fmac acc,s,t", "cmp+cndmask (2 instr)", "v_min a,b", "v_sub c=a-b", "sub+fma (2 instr)"};
    const int per[] = {1, 1, 1, 1, 2, 1, 1, 2};
    for (int w : {2, 4}) {
        printf("waves/SIMD=%d: SIMD cycles per instruction (2.4 GHz assumed)\n", w);
        double t[8];
        t[0] = run<0>(d, w, 4000, per[0]); t[1] = run<1>(d, w, 4000, per[1]); t[2] = run<2>(d, w, 4000, per[2]); t[3] = run<3>(d, w, 4000, per[3]);
        t[4] = run<4>(d, w, 4000, per[4]); t[5] = run<5>(d, w, 4000, per[5]); t[6] = run<6>(d, w, 4000, per[6]); t[7] = run<7>(d, w, 4000, per[7]);
        for (int m = 0; m < 8; ++m) printf("  %-24s %.2f\n", names[m], t[m]);
    }
    return 0;
}

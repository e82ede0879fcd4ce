// Micro-benchmark: issue cost of cross-lane primitives relative to a plain VALU op on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/ubench_valu tools/ubench/ubench_valu.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_REG 16
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    float r[N_REG];
    for (int q = 0; q < N_REG; ++q) r[q] = (float)(threadIdx.x + q);
    const int idx = ((threadIdx.x + 1) & 63) * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
#pragma unroll
            for (int q = 0; q < N_REG; ++q) {
                if (MODE == 0) r[q] = r[q] + 1.0f;                                                   // v_add
                if (MODE == 1) r[q] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, r[q]), __builtin_bit_cast(int, r[(q + 1) % N_REG]), 0x101, 0xf, 0xf, false));  // row_shl:1
                if (MODE == 2) r[q] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, r[q]), __builtin_bit_cast(int, r[(q + 1) % N_REG]), 0x130, 0xf, 0xf, false));  // wave_shl:1
                if (MODE == 3) r[q] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(idx, __builtin_bit_cast(int, r[(q + 1) % N_REG])));
                if (MODE == 4) r[q] = __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, r[(q + 1) % N_REG]), 0x8001 /* rotate? */));
                if (MODE == 5) r[q] = __builtin_amdgcn_sqrtf(r[q]);
                if (MODE == 6) r[q] = __builtin_fmaf(r[q], 1.0001f, r[(q + 1) % N_REG]);
                if (MODE == 7) r[q] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, r[q]), __builtin_bit_cast(int, r[(q + 1) % N_REG]), 0x111, 0xf, 0xf, false));  // row_shr:1
                if (MODE == 8) r[q] = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, r[(q + 1) % N_REG]), 0xB1 /*quad_perm [1,0,3,2]*/, 0xf, 0xf, true));
                if (MODE == 9) r[q] = fminf(fminf(r[q], r[(q + 1) % N_REG]), r[(q + 2) % N_REG]);  // v_min3
            }
        }
    }
    float s = 0;
    for (int q = 0; q < N_REG; ++q) s += r[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
double run(float *d, int waves_per_simd, int iters)
{
    const int blocks = 256 * waves_per_simd;   // 256 CUs x (4 waves per block = 1 wave per SIMD) x waves_per_simd
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double instr = (double)iters * 4 * N_REG;          // per wave
    return ms * 1e-3 / instr * 2.4e9;                        // cycles (at 2.4 GHz) per wave-instruction per SIMD-resident wave set
}

int main()
{
    float *d;
    hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    const char *names[] = {"v_add_f32", "dpp row_shl:1", "dpp wave_shl:1", "ds_bpermute", "ds_swizzle", "v_sqrt_f32", "v_fma_f32", "dpp row_shr:1", "dpp quad_perm", "v_min3_f32"};
    for (int w : {1, 2, 4}) {
        printf("waves/SIMD=%d: wall cycles per wave-instruction (2.4 GHz assumed), i.e. cycles between issues of ONE wave\n", w);
        double t[10];
        t[0] = run<0>(d, w, 4000); t[1] = run<1>(d, w, 4000); t[2] = run<2>(d, w, 4000); t[3] = run<3>(d, w, 2000);
        t[4] = run<4>(d, w, 2000); t[5] = run<5>(d, w, 4000); t[6] = run<6>(d, w, 4000); t[7] = run<7>(d, w, 4000);
        t[8] = run<8>(d, w, 4000); t[9] = run<9>(d, w, 4000);
        for (int m = 0; m < 10; ++m) printf("  %-16s %.2f  (SIMD throughput: %.2f cyc/instr)\n", names[m], t[m], t[m] / w);
    }
    return 0;
}

// Microbenchmark: sustained f32-MFMA rate on gfx950 under different issue mixes (what is the real ceiling the
// network kernels should be priced against once DVFS has settled?).   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: 32x32x2 x2 acc; 1: 16x16x4 x5 acc; 2: 16x16x4 x5 acc + 2 VALU per MFMA; 3: 32x32x2 x2 + 2 VALU
__global__ __launch_bounds__(256) void k(float *out, int iters, float s) {
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    if constexpr (MODE == 0 || MODE == 3) {
        f32x16 c0 = {0}, c1 = {0};
        float x = a, y = b;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, b, c0, 0, 0, 0);
                if (MODE == 3) { x = x * s + 1.0f; __builtin_amdgcn_sched_barrier(0); }
                c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b, c1, 0, 0, 0);
                if (MODE == 3) { y = y * s + 1.0f; __builtin_amdgcn_sched_barrier(0); }
            }
        }
        float r = 0;
        for (int j = 0; j < 16; ++j) r += c0[j] + c1[j];
        out[blockIdx.x * 256 + threadIdx.x] = r;
    } else {
        f32x4 c[5] = {{0}, {0}, {0}, {0}, {0}};
        float x[5] = {a, a + 1, a + 2, a + 3, a + 4};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    c[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[t], b, c[t], 0, 0, 0);
                    if (MODE == 2) { x[t] = (float)(__float_as_uint(x[t]) & 0xff) * s; __builtin_amdgcn_sched_barrier(0); }
                }
        }
        float r = 0;
        for (int t = 0; t < 5; ++t) r += c[t][0] + c[t][1] + c[t][2] + c[t][3];
        out[blockIdx.x * 256 + threadIdx.x] = r;
    }
}

template <int MODE>
void run(const char *name, int blocks_per_cu, double flops_per_iter_per_wave) {
    const int blocks = 256 * blocks_per_cu, iters = 20000;
    float *out;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, 100, 1.0001f);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        k<MODE><<<blocks, 256>>>(out, iters, 1.0001f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)blocks * 4 * iters * flops_per_iter_per_wave;
        printf("%-44s blocks/CU %d  %8.2f ms  %7.1f TFLOP/s\n", name, blocks_per_cu, ms, fl / ms / 1e9);
    }
    hipFree(out);
}

int main() {
    for (int bpc = 1; bpc <= 3; ++bpc) {
        run<0>("32x32x2 f32, 2 acc, MFMA only", bpc, 16 * 4096.0);
        run<3>("32x32x2 f32, 2 acc, + 1 VALU(fma) / MFMA", bpc, 16 * 4096.0);
        run<1>("16x16x4 f32, 5 acc, MFMA only", bpc, 20 * 2048.0);
        run<2>("16x16x4 f32, 5 acc, + 3 VALU / MFMA", bpc, 20 * 2048.0);
    }
    return 0;
}

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


// MODE 4: the k-loop of sf_nn_img.h in isolation — 16x16x4, 4 accumulators, A fragments from one ds_read_b128 per 4
// MFMAs (double-buffered, issued mid-group), B from 36 distinct register quads (compile-time indices), random data.
// MODE 5: the same without the LDS reads (A from registers).
template <int MODE>
__global__ __launch_bounds__(256) void k2(float *out, const float *src, int iters) {
    __shared__ __attribute__((aligned(16))) float img[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) img[i] = src[i];
    f32x4 breg[36];
#pragma unroll
    for (int g = 0; g < 36; ++g) breg[g] = *reinterpret_cast<const f32x4 *>(src + 8192 + (g * 256 + threadIdx.x) * 4 % 8192);
    __syncthreads();
    f32x4 c[4] = {{0}, {0}, {0}, {0}};
    const int lane = threadIdx.x & 63;
    const float *base[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) base[f] = img + ((f * 16 + (lane & 15)) * 4 + (lane >> 4) * 1296) % 4096;
    f32x4 a[2][4];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int f = 0; f < 4; ++f) a[0][f] = *reinterpret_cast<const f32x4 *>(base[f]);
#pragma unroll
        for (int g = 0; g < 36; ++g) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int f = 0; f < 4; ++f) c[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g & 1][f][j], breg[g][j], c[f], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < 36) {
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    if (MODE == 4) a[(g + 1) & 1][f] = *reinterpret_cast<const f32x4 *>(base[f] + (g + 1) * 100 % 4000 / 4 * 4);
                    else a[(g + 1) & 1][f] = a[g & 1][f];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 2; j < 4; ++j)
#pragma unroll
                for (int f = 0; f < 4; ++f) c[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[g & 1][f][j], breg[g][j], c[f], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float r = 0;
    for (int t = 0; t < 4; ++t) r += c[t][0] + c[t][1] + c[t][2] + c[t][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
void run2(const char *name, int blocks_per_cu, bool zeros) {
    const int blocks = 256 * blocks_per_cu, iters = 600;
    float *out, *src;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&src, 16384 * sizeof(float));
    float *h = (float *)malloc(16384 * sizeof(float));
    for (int i = 0; i < 16384; ++i) h[i] = zeros ? 0.f : (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(src, h, 16384 * sizeof(float), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k2<MODE><<<blocks, 256>>>(out, src, 10);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        k2<MODE><<<blocks, 256>>>(out, src, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)blocks * 4 * iters * 36 * 16 * 2048.0;
        printf("%-44s blocks/CU %d %s %8.2f ms  %7.1f TFLOP/s\n", name, blocks_per_cu, zeros ? "zeros " : "random", ms, fl / ms / 1e9);
    }
    hipFree(out); hipFree(src); free(h);
}

int main() {
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run2<4>("16x16x4, 4 acc, 36 B quads, LDS A (img loop)", bpc, false);
        run2<4>("16x16x4, 4 acc, 36 B quads, LDS A (img loop)", bpc, true);
        run2<5>("16x16x4, 4 acc, 36 B quads, A in regs", bpc, false);
    }
    for (int bpc = 1; bpc <= 3; ++bpc) {
        run<0>("32x32x2 f32, 2 acc, MFMA only", bpc, 16 * 4096.0);
        run<3>("32x32x2 f32, 2 acc, + 1 VALU(fma) / MFMA", bpc, 16 * 4096.0);
        run<1>("16x16x4 f32, 5 acc, MFMA only", bpc, 20 * 2048.0);
        run<2>("16x16x4 f32, 5 acc, + 3 VALU / MFMA", bpc, 20 * 2048.0);
    }
    return 0;
}

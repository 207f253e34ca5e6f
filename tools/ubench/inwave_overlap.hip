// Does VALU work overlap with MFMAs INSIDE one wave on gfx950 (one wave per SIMD, program-order interleave)?
//   hipcc --offload-arch=gfx950 -O3 inwave_overlap.hip -o inwave && ./inwave
// kernel<MODE>: 1 = MFMAs only (v_mfma_f32_16x16x32_bf16: 4 passes), 2 = VALU only (3 independent v_fma per MFMA slot),
// 3 = both, alternating in program order.  Same for the 8-pass f32 MFMA (16x16x4) with 6 VALU per slot.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, bool BF, int NV>
__global__ __launch_bounds__(256, 1) void k(float *out, int iters) {
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + i); b[i] = (__bf16)(0.5f * i); }
    const float fa = 1.0001f, fb = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE & 1) {
                if (BF) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j & 3], 0, 0, 0);
                else acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, acc[j & 3], 0, 0, 0);
            }
            if (MODE & 2) {
#pragma unroll
                for (int q = 0; q < NV; ++q) v[(j * NV + q) & 7] = __builtin_fmaf(v[(j * NV + q) & 7], fa, fb);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + s;
}

template <int MODE, bool BF, int NV>
float run(float *out, int iters) {
    k<MODE, BF, NV><<<256, 256>>>(out, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE, BF, NV><<<256, 256>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *out;
    (void)hipMalloc(&out, 256 * 256 * 4);
    const int iters = 4000;
    printf("bf16 16x16x32 (4 passes) + 3 VALU per MFMA: mfma %.3f ms, valu %.3f ms, interleaved %.3f ms\n",
           run<1, true, 3>(out, iters), run<2, true, 3>(out, iters), run<3, true, 3>(out, iters));
    printf("bf16 16x16x32 (4 passes) + 2 VALU per MFMA: mfma %.3f ms, valu %.3f ms, interleaved %.3f ms\n",
           run<1, true, 2>(out, iters), run<2, true, 2>(out, iters), run<3, true, 2>(out, iters));
    printf("f32 16x16x4 (8 passes) + 6 VALU per MFMA:   mfma %.3f ms, valu %.3f ms, interleaved %.3f ms\n",
           run<1, false, 6>(out, iters), run<2, false, 6>(out, iters), run<3, false, 6>(out, iters));
    printf("f32 16x16x4 (8 passes) + 3 VALU per MFMA:   mfma %.3f ms, valu %.3f ms, interleaved %.3f ms\n",
           run<1, false, 3>(out, iters), run<2, false, 3>(out, iters), run<3, false, 3>(out, iters));
    return 0;
}

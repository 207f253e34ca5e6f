// Does VALU work overlap with MFMAs INSIDE one wave on gfx950 (one wave per SIMD, program-order interleave)?
//   hipcc --offload-arch=gfx950 -O3 inwave_overlap.hip -o inwave && ./inwave
// kernel<MODE>: 1 = MFMAs only (v_mfma_f32_16x16x32_bf16: 4 passes), 2 = VALU only (3 independent v_fma per MFMA slot),
// 3 = both, alternating in program order.  Same for the 8-pass f32 MFMA (16x16x4) with 6 VALU per slot.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, bool BF, int NV>
__global__ __launch_bounds__(256, 1) void k(float *out, int iters, long long *cyc) {
    const long long t0 = __builtin_readcyclecounter();  // s_memtime: shader-clock ticks (tools/ubench/wt_drain.hip: 2.36 GHz)
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
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + s;
}

static double last_cycles_per_slot = 0;
template <int MODE, bool BF, int NV>
float run(float *out, int iters) {
    static long long *cyc = nullptr;
    if (!cyc) (void)hipMalloc(&cyc, 1024 * sizeof(long long));
    k<MODE, BF, NV><<<256, 256>>>(out, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    k<MODE, BF, NV><<<256, 256>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    long long h[1024];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < 1024; ++i) sum += (double)h[i];
    last_cycles_per_slot = sum / 1024.0 / ((double)iters * 16.0);
    return ms;
}

template <bool BF, int NV>
void row(const char *name, float *out, int iters) {
    const float m = run<1, BF, NV>(out, iters); const double cm = last_cycles_per_slot;
    const float v = run<2, BF, NV>(out, iters); const double cv = last_cycles_per_slot;
    const float b = run<3, BF, NV>(out, iters); const double cb = last_cycles_per_slot;
    printf("%s: mfma %.3f ms (%.1f shader cycles per slot), valu %.3f ms (%.1f), interleaved %.3f ms (%.1f)\n", name, m, cm, v, cv, b, cb);
}

int main() {
    float *out;
    (void)hipMalloc(&out, 256 * 256 * 4);
    const int iters = 4000;
    row<true, 3>("bf16 16x16x32 + 3 VALU per MFMA", out, iters);
    row<true, 2>("bf16 16x16x32 + 2 VALU per MFMA", out, iters);
    row<false, 6>("f32 16x16x4 + 6 VALU per MFMA  ", out, iters);
    row<false, 3>("f32 16x16x4 + 3 VALU per MFMA  ", out, iters);
    return 0;
}

// Two questions about one CU of gfx950, answered by measurement (hipcc --offload-arch=gfx950 -O3 simd_overlap.hip):
//  1. which SIMD do the 8 waves of a 512-thread work-group land on (HW_ID.simd_id per wave);
//  2. do the MFMAs of one wave and the VALU work (expf / tanhf chains) of ANOTHER wave on the same SIMD overlap?
//     modes: 0 = both waves of a SIMD multiply, 1 = both run VALU, 2 = one multiplies while the other runs VALU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512, 1) void k_ids(unsigned *out) {
    __shared__ float pad[32 * 1024];
    pad[threadIdx.x] = 0.f;
    const unsigned id = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_REG_HW_ID
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}

__global__ __launch_bounds__(512, 1) void k_mix(float *out, int iters, int mode, long long *cycles) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // role by SIMD partner: waves w and w + 4 (question 1 says whether they share a SIMD)
    const bool second = wave >= 4;
    const bool do_mfma = mode == 0 || (mode == 2 && !second) || (mode == 3 && !second);
    const bool do_valu = mode == 1 || (mode == 2 && second) || (mode == 4 && second);
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    float a = 0.5f + lane * 1e-3f, b = 0.25f, v = 0.1f * lane;
    const long long t0 = __builtin_readcyclecounter();
    if (do_mfma) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc1, 0, 0, 0);
            }
        }
    }
    if (do_valu) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v = tanhf(v * 0.9f + 0.01f) + 1.0f / (1.0f + expf(-v));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = acc0[0] + acc1[1] + v;
}

int main() {
    unsigned *ids; float *out; long long *cyc;
    hipMalloc(&ids, 256 * 8 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    k_ids<<<256, 512>>>(ids);
    unsigned h[256 * 8];
    hipMemcpy(h, ids, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 3; ++b) {
        printf("block %d: simd of waves 0..7:", b);
        for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3);
        printf("   cu %u se %u\n", (h[b * 8] >> 8) & 15, (h[b * 8] >> 13) & 7);
    }
    int bad = 0;
    for (int b = 0; b < 256; ++b) for (int w = 0; w < 4; ++w) bad += ((h[b * 8 + w] >> 4) & 3) != ((h[b * 8 + w + 4] >> 4) & 3);
    printf("blocks x waves where wave w and w+4 are NOT on the same SIMD: %d of 1024\n", bad);
    const int iters = 2000;
    for (int mode = 0; mode <= 4; ++mode) {
        k_mix<<<256, 512>>>(out, iters, mode, cyc);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        k_mix<<<256, 512>>>(out, iters, mode, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const char *names[] = {"both waves MFMA", "both waves VALU", "wave w MFMA + wave w+4 VALU", "only wave w MFMA", "only wave w+4 VALU"};
        printf("mode %d (%s): %.3f ms\n", mode, names[mode], ms);
    }
    return 0;
}

// Shared helpers for libsf_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "sf_hip.h"

extern thread_local char sf_err_buf[512];

#define SF_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) {                                          \
            snprintf(sf_err_buf, sizeof(sf_err_buf), __VA_ARGS__); \
            return SF_ERR_ARG;                                  \
        }                                                       \
    } while (0)

static inline int sf_launch_status(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(sf_err_buf, sizeof(sf_err_buf), "%s: %s", what, hipGetErrorString(e));
        return SF_ERR_LAUNCH;
    }
    return SF_OK;
}

static inline int sf_hip_status(hipError_t e, const char *what) {
    if (e != hipSuccess) {
        snprintf(sf_err_buf, sizeof(sf_err_buf), "%s: %s", what, hipGetErrorString(e));
        return SF_ERR_LAUNCH;
    }
    return SF_OK;
}

constexpr int SF_WAVE = 64;  // CDNA wavefront

__device__ __forceinline__ double sf_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ float sf_wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_down(v, off, 64));
    return v;
}

// Block-wide sum of NV doubles for a 256-thread block (4 waves); result valid in thread 0.
template <int NV>
__device__ __forceinline__ void sf_block_sum(double (&v)[NV], double *lds /* >= 4*NV doubles */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = sf_wave_sum(v[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) lds[wave * NV + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            double s = 0.0;
            for (int w = 0; w < nw; ++w) s += lds[w * NV + i];
            v[i] = s;
        }
    }
}

// Philox4x32-10 (Salmon et al., Random123); known-answer vectors are checked by the test-suite.
__host__ __device__ __forceinline__ void sf_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                         uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

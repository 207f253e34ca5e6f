// Microbenchmark for DESIGN.md §8.8 item 1: how fast is an f32 x f32 GEMM on the bf16 matrix pipe with EXACT products?
//   out[M][N] = A[M][K] * W[N][K]^T,  A f32 (split into 3 bf16 terms in the loader), W pre-split into 3 bf16 planes
//   (once per optimiser step in a real engine), all 9 partial products of the two splits (each exact in f32), f32
//   accumulation in v_mfma_f32_16x16x32_bf16.  Shape = the fc layer of the C2 network at a training minibatch
//   (32768 x 3136 x 512; the f32-MFMA kernel k_fwd_glds<128,128> does it in 0.82-0.85 ms = 124-128 TFLOP/s).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/gemm_x9.hip -o tools/ubench/gemm_x9 && tools/ubench/gemm_x9
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int PLANE = 128 * BK;            // elements of one operand plane in a stage
constexpr int STAGE = 6 * PLANE;           // A planes 0..2, B planes 3..5

// element offset of 16-byte chunk c (8 elements) of row r inside a plane: rows are 64 B, chunk position swizzled so
// that the 8 rows one ds_read_b128 cycle touches (8 lanes = 8 consecutive rows, same chunk) cover all 32 banks
__device__ __forceinline__ int swz(int r, int c) { return r * BK + ((c ^ ((r >> 1) & 3)) << 3); }

template <int NPROD, int NSTAGE>  // 9: every partial product (exact); 6: without the three terms of relative size <= 2^-24
__global__ __launch_bounds__(256, 2) void k_gemm_x(const float *__restrict__ A, const uint16_t *__restrict__ Wp,
                                                float *__restrict__ out, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) uint16_t lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int i16 = lane & 15, kg = lane >> 4;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    // loader roles
    const int arow = tid >> 1, ahalf = tid & 1;                       // A: 16 consecutive k of one row
    const float *ap = A + (size_t)(m0 + arow) * K + ahalf * 16;
    f32x4 areg[4];
    u32x4 breg[6];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) areg[j] = *reinterpret_cast<const f32x4 *>(ap + k0 + 4 * j);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int idx = tid + 256 * j, p = idx >> 9, rem = idx & 511, row = rem >> 2, ch = rem & 3;
            breg[j] = *reinterpret_cast<const u32x4 *>(Wp + ((size_t)p * N + n0 + row) * K + k0 + ch * 8);
        }
    };
    auto lstore = [&](int stage) {
        uint16_t *st = lds + stage * STAGE;
        uint32_t h[8], m[8], l[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {  // pairs of consecutive k: exact 3-way split, one v_perm per term and pair
            const float v0 = areg[q >> 1][(q & 1) * 2], v1 = areg[q >> 1][(q & 1) * 2 + 1];
            const uint32_t b0 = __float_as_uint(v0), b1 = __float_as_uint(v1);
            const float r0 = v0 - __uint_as_float(b0 & 0xFFFF0000u), r1 = v1 - __uint_as_float(b1 & 0xFFFF0000u);
            const uint32_t c0 = __float_as_uint(r0), c1 = __float_as_uint(r1);
            const float s0 = r0 - __uint_as_float(c0 & 0xFFFF0000u), s1 = r1 - __uint_as_float(c1 & 0xFFFF0000u);
            h[q] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
            m[q] = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
            l[q] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int off = swz(arow, ahalf * 2 + c);
            *reinterpret_cast<u32x4 *>(st + 0 * PLANE + off) = u32x4{h[4 * c], h[4 * c + 1], h[4 * c + 2], h[4 * c + 3]};
            *reinterpret_cast<u32x4 *>(st + 1 * PLANE + off) = u32x4{m[4 * c], m[4 * c + 1], m[4 * c + 2], m[4 * c + 3]};
            *reinterpret_cast<u32x4 *>(st + 2 * PLANE + off) = u32x4{l[4 * c], l[4 * c + 1], l[4 * c + 2], l[4 * c + 3]};
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int idx = tid + 256 * j, p = idx >> 9, rem = idx & 511, row = rem >> 2, ch = rem & 3;
            *reinterpret_cast<u32x4 *>(st + (3 + p) * PLANE + swz(row, ch)) = breg[j];
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    gload(0);
    lstore(0);
    __syncthreads();
    int stage = 0;
    for (int k0 = 0; k0 < K; k0 += BK, stage ^= (NSTAGE - 1)) {
        if (k0 + BK < K) gload(k0 + BK);
        const uint16_t *st = lds + stage * STAGE;
        s16x8 af[3][4], bf[3][4];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[p][t] = *reinterpret_cast<const s16x8 *>(st + p * PLANE + swz(wm * 64 + t * 16 + i16, kg));
                bf[p][t] = *reinterpret_cast<const s16x8 *>(st + (3 + p) * PLANE + swz(wn * 64 + t * 16 + i16, kg));
            }
        // small terms first; plane index = 0 hi, 1 mid, 2 lo; product (pa, pb) has relative size 2^-8(pa+pb)
        constexpr int order[9][2] = {{2, 2}, {2, 1}, {1, 2}, {2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};
#pragma unroll
        for (int o = 9 - NPROD; o < 9; ++o)
#pragma unroll
            for (int ta = 0; ta < 4; ++ta)
#pragma unroll
                for (int tb = 0; tb < 4; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, af[order[o][0]][ta]), __builtin_bit_cast(bf16x8, bf[order[o][1]][tb]),
                        acc[ta][tb], 0, 0, 0);
        if (NSTAGE == 1) __syncthreads();  // single stage: everybody is done reading before it is overwritten
        if (k0 + BK < K) lstore(stage ^ (NSTAGE - 1));
        __syncthreads();
    }
    // C layout: col = lane & 15, row = 4*kg + r
#pragma unroll
    for (int ta = 0; ta < 4; ++ta)
#pragma unroll
        for (int tb = 0; tb < 4; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                out[(size_t)(m0 + wm * 64 + ta * 16 + 4 * kg + r) * N + n0 + wn * 64 + tb * 16 + i16] = acc[ta][tb][r];
}

static void split3(float w, uint16_t &h, uint16_t &m, uint16_t &l) {
    uint32_t b; memcpy(&b, &w, 4);
    h = b >> 16;
    uint32_t hb = b & 0xFFFF0000u; float hf; memcpy(&hf, &hb, 4);
    float r1 = w - hf; uint32_t b1; memcpy(&b1, &r1, 4);
    m = b1 >> 16;
    uint32_t mb = b1 & 0xFFFF0000u; float mf; memcpy(&mf, &mb, 4);
    float r2 = r1 - mf; uint32_t b2; memcpy(&b2, &r2, 4);
    l = b2 >> 16;
}

template <int NPROD, int NSTAGE>
static void run(const float *dA, const uint16_t *dW, float *dO, int M, int N, int K, const std::vector<float> &hA,
                const std::vector<float> &hW) {
    const size_t ldsb = NSTAGE * STAGE * sizeof(uint16_t);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k_gemm_x<NPROD, NSTAGE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    dim3 grid(M / BM, N / BN);
    k_gemm_x<NPROD, NSTAGE><<<grid, 256, ldsb>>>(dA, dW, dO, M, N, K);
    hipDeviceSynchronize();
    std::vector<float> hO((size_t)64 * N);
    hipMemcpy(hO.data(), dO + (size_t)(M - 64) * N, hO.size() * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int r = 0; r < 64; r += 7)
        for (int n = 0; n < N; n += 13) {
            double s = 0;
            for (int k = 0; k < K; ++k) s += (double)hA[(size_t)(M - 64 + r) * K + k] * (double)hW[(size_t)n * K + k];
            maxerr = fmax(maxerr, fabs(s - hO[(size_t)r * N + n]));
            maxref = fmax(maxref, fabs(s));
        }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, sum = 0;
    for (int rep = 0; rep < 10; ++rep) {
        hipEventRecord(e0);
        k_gemm_x<NPROD, NSTAGE><<<grid, 256, ldsb>>>(dA, dW, dO, M, N, K);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = fminf(best, ms); sum += ms;
    }
    printf("%d products, %d LDS stage(s): best %.3f ms, mean %.3f ms = %.1f TFLOP/s f32-equivalent (2MNK), max err vs f64 %.2e of max |ref| %.2f -> %.2e relative\n",
           NPROD, NSTAGE, best, sum / 10, 2.0 * M * N * K / (sum / 10 * 1e-3) / 1e12, maxerr, maxref, maxerr / maxref);
}

int main() {
    const int M = 32768, K = 3136, N = 512;
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
    srand(1);
    auto rnd = []() { return (float)((rand() / (double)RAND_MAX + rand() / (double)RAND_MAX + rand() / (double)RAND_MAX - 1.5) * 1.4); };
    for (auto &v : hA) v = fmaxf(0.f, rnd());   // post-ReLU activations
    for (auto &v : hW) v = rnd() / 56.f;
    std::vector<uint16_t> hWp((size_t)3 * N * K);
    for (size_t i = 0; i < hW.size(); ++i) split3(hW[i], hWp[i], hWp[(size_t)N * K + i], hWp[(size_t)2 * N * K + i]);
    float *dA, *dO; uint16_t *dW;
    hipMalloc(&dA, hA.size() * 4); hipMalloc(&dO, (size_t)M * N * 4); hipMalloc(&dW, hWp.size() * 2);
    hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dW, hWp.data(), hWp.size() * 2, hipMemcpyHostToDevice);
    run<9, 2>(dA, dW, dO, M, N, K, hA, hW);
    run<9, 1>(dA, dW, dO, M, N, K, hA, hW);
    run<6, 2>(dA, dW, dO, M, N, K, hA, hW);
    run<6, 1>(dA, dW, dO, M, N, K, hA, hW);
    return 0;
}

// Go / no-go microbenchmark for moving the f32 x f32 layers (conv2 / conv3 / fc) onto the bf16 matrix pipe with EXACT
// products: out[M][N] = A[M][K] * W[N][K]^T where BOTH operands arrive PRE-SPLIT into three bf16
// planes (A: written that way by the producing layer's epilogue, 1.5x the bytes of f32; W: split once per optimiser
// step), all 9 partial products of the two 3-term splits (each exact in f32), f32 accumulation in
// v_mfma_f32_16x16x32_bf16.  No VALU in the k-loop: both operands go global -> LDS by DMA (global_load_lds_dwordx4),
// fragments by ds_read_b128.  512-thread work-groups (two waves per SIMD), double-buffered LDS, one barrier per chunk.
//
// Shapes: the fc layer of a C2 training minibatch (32768 x 3136 x 512) and the conv2 forward as a dense GEMM over its
// im2col rows (32768*81 x 512 x 64: an UPPER bound for the implicit-GEMM kernel, whose gather costs extra).
// Gate: >= 1.35x over k_fwd_glds at both shapes with float64 error <= the f32-MFMA kernel's.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/gemm_x9_dma.hip -o tools/ubench/gemm_x9_dma && tools/ubench/gemm_x9_dma
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

#define GLDS16(gsrc, ldst)                                                                       \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc),    \
                                     (__attribute__((address_space(3))) void *)(ldst), 16, 0, 0)

constexpr int BK = 32;  // reduction elements per chunk: 64-byte LDS rows, one 16-row fragment = one 1-KiB DMA instruction

// A-planes [3][M][K], W-planes [3][N][K] (bf16 bit patterns).  WGM x WGN waves, wave tile (BM/WGM) x (BN/WGN).
template <int BM, int BN, int WGM, int WGN, int NPROD>
__global__ __launch_bounds__(64 * WGM * WGN) void k_gemm_x9(const uint16_t *__restrict__ Ap, const uint16_t *__restrict__ Wp,
                                                            float *__restrict__ out, int M, int N, int K,
                                                            unsigned long long *__restrict__ clk) {
    // clock probe (round 6): the work-group in the middle of the grid records s_memtime ticks (shader cycles) and the
    // 100 MHz wall clock over its own life time: ticks / (10 ns units) * 0.1 = the shader clock in GHz WHILE THIS KERNEL RUNS
    const bool probe = clk && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && threadIdx.x == 0;
    unsigned long long t0 = 0, w0 = 0;
    if (probe) { t0 = __builtin_amdgcn_s_memtime(); w0 = wall_clock64(); }
    constexpr int NW = WGM * WGN, TM = BM / WGM / 16, TN = BN / WGN / 16;
    constexpr int FA = BM / 16, FB = BN / 16;               // 16-row fragments per plane
    constexpr int NFR = 3 * (FA + FB);                      // DMA instructions per chunk and work-group
    constexpr int NI = (NFR + NW - 1) / NW;                 // ... per wave
    constexpr int STAGE = NFR * 1024;                       // bytes per pipeline stage
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int i16 = lane & 15, kg = lane >> 4;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    // ---- DMA sources.  Instruction q of the chunk = fragment q of the stage: q < 3*FA: A plane q / FA, fragment q % FA;
    // else W.  Lane l lands at LDS position l of the fragment = (row l >> 2, slot l & 3); slot s of row r holds k-chunk
    // s ^ ((r >> 1) & 3) (bank swizzle applied on the SOURCE side, the same involution on the ds_read address).
    const int lr = lane >> 2, lc = (lane & 3) ^ ((lr >> 1) & 3);
    const uint16_t *src[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int q = wave + NW * j;
        if (q < 3 * FA) {
            const int p = q / FA, f = q % FA;
            int64_t m = m0 + 16 * f + lr;
            m = m < M ? m : M - 1;
            src[j] = Ap + ((int64_t)p * M + m) * K + 8 * lc;
        } else {
            const int qq = (q < NFR ? q : NFR - 1) - 3 * FA, p = qq / FB, f = qq % FB;
            src[j] = Wp + ((int64_t)p * N + n0 + 16 * f + lr) * K + 8 * lc;
        }
    }
    auto issue = [&](int stage, int k0) {
        char *st = lds + stage * STAGE;
#pragma unroll
        for (int j = 0; j < NI; ++j)
            if (wave + NW * j < NFR) GLDS16(src[j] + k0, st + (wave + NW * j) * 1024);
    };
    f32x4 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment read offset of this lane inside a 1-KiB fragment
    const int foff = i16 * 64 + ((kg ^ ((i16 >> 1) & 3)) << 4);
    issue(0, 0);
    const int KT = K / BK;
    for (int kt = 0; kt < KT; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 1 < KT) issue((kt + 1) & 1, (kt + 1) * BK);
        const char *st = lds + (kt & 1) * STAGE;
        s16x8 af[3][TM], bf[3][TN];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int t = 0; t < TM; ++t)
                af[p][t] = *reinterpret_cast<const s16x8 *>(st + (p * FA + wm * TM + t) * 1024 + foff);
#pragma unroll
            for (int t = 0; t < TN; ++t)
                bf[p][t] = *reinterpret_cast<const s16x8 *>(st + (3 * FA + p * FB + wn * TN + t) * 1024 + foff);
        }
        // small terms first; plane 0 = hi, 1 = mid, 2 = lo; product (pa, pb) has relative size 2^-8(pa+pb)
        constexpr int order[9][2] = {{2, 2}, {2, 1}, {1, 2}, {2, 0}, {0, 2}, {1, 1}, {1, 0}, {0, 1}, {0, 0}};
#pragma unroll
        for (int o = 9 - NPROD; o < 9; ++o)
#pragma unroll
            for (int ta = 0; ta < TM; ++ta)
#pragma unroll
                for (int tb = 0; tb < TN; ++tb)
                    acc[ta][tb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(bf16x8, af[order[o][0]][ta]), __builtin_bit_cast(bf16x8, bf[order[o][1]][tb]),
                        acc[ta][tb], 0, 0, 0);
    }
    // C layout of 16x16x32: col = lane & 15, rows 4*kg + r
#pragma unroll
    for (int ta = 0; ta < TM; ++ta)
#pragma unroll
        for (int tb = 0; tb < TN; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t m = m0 + (wm * TM + ta) * 16 + 4 * kg + r;
                if (m < M) out[m * N + n0 + (wn * TN + tb) * 16 + i16] = acc[ta][tb][r];
            }
    if (probe) { clk[0] = __builtin_amdgcn_s_memtime() - t0; clk[1] = wall_clock64() - w0; }
}

// ---- operands generated and split on the device (the conv2 shape is 1.36 G elements)
__device__ __forceinline__ uint32_t mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float gen(uint64_t i, uint32_t seed, int relu, float scale) {
    const uint32_t h = mix((uint32_t)i * 2654435761u ^ mix((uint32_t)(i >> 32) + seed));
    const uint32_t h2 = mix(h + 0x9e3779b9u);
    float v = ((h >> 8) * (1.0f / 16777216.0f) + (h2 >> 8) * (1.0f / 16777216.0f) - 1.0f) * scale;
    return relu ? fmaxf(v, 0.f) : v;
}
__global__ void k_gen_split(uint16_t *planes, float *f32copy, int64_t n, uint32_t seed, int relu, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = gen((uint64_t)i, seed, relu, scale);
        const uint32_t b = __float_as_uint(v);
        const float r1 = v - __uint_as_float(b & 0xFFFF0000u);
        const uint32_t b1 = __float_as_uint(r1);
        const float r2 = r1 - __uint_as_float(b1 & 0xFFFF0000u);
        planes[i] = (uint16_t)(b >> 16);
        planes[n + i] = (uint16_t)(b1 >> 16);
        planes[2 * n + i] = (uint16_t)(__float_as_uint(r2) >> 16);
        if (f32copy) f32copy[i] = v;
    }
}
static float hgen(uint64_t i, uint32_t seed, int relu, float scale) {  // host copy of gen()
    auto mixh = [](uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; };
    const uint32_t h = mixh((uint32_t)i * 2654435761u ^ mixh((uint32_t)(i >> 32) + seed));
    const uint32_t h2 = mixh(h + 0x9e3779b9u);
    float v = ((h >> 8) * (1.0f / 16777216.0f) + (h2 >> 8) * (1.0f / 16777216.0f) - 1.0f) * scale;
    return relu ? fmaxf(v, 0.f) : v;
}

template <int BM, int BN, int WGM, int WGN, int NPROD>
static void run(const char *label, const uint16_t *dA, const uint16_t *dW, float *dO, int M, int N, int K, double ref_ms) {
    constexpr int NFR = 3 * (BM / 16 + BN / 16);
    const size_t ldsb = 2 * NFR * 1024;
    auto kern = k_gemm_x9<BM, BN, WGM, WGN, NPROD>;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) {
        printf("%s: cannot set %zu B of LDS\n", label, ldsb);
        return;
    }
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 64 * WGM * WGN, ldsb);
    dim3 grid((M + BM - 1) / BM, N / BN);
    kern<<<grid, 64 * WGM * WGN, ldsb>>>(dA, dW, dO, M, N, K, nullptr);
    if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed: %s\n", label, hipGetErrorString(hipGetLastError())); return; }
    // float64 check on sampled entries of the LAST 64 rows and the first 64 rows
    double maxerr = 0, maxref = 0;
    std::vector<float> hO((size_t)64 * N);
    for (int part = 0; part < 2; ++part) {
        const int64_t r0 = part ? (int64_t)M - 64 : 0;
        hipMemcpy(hO.data(), dO + r0 * N, hO.size() * 4, hipMemcpyDeviceToHost);
        for (int r = 0; r < 64; r += 7)
            for (int n = 0; n < N; n += 13) {
                double s = 0;
                for (int k = 0; k < K; ++k)
                    s += (double)hgen((uint64_t)(r0 + r) * K + k, 1u, 1, 1.4f) * (double)hgen((uint64_t)n * K + k, 2u, 0, 1.4f / sqrtf((float)K));
                maxerr = fmax(maxerr, fabs(s - hO[(size_t)r * N + n]));
                maxref = fmax(maxref, fabs(s));
            }
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f, sum = 0;
    const int reps = 12;
    unsigned long long *dclk, hclk[2];
    hipMalloc(&dclk, 16);
    double ghz_sum = 0;
    for (int rep = 0; rep < reps; ++rep) {
        hipEventRecord(e0);
        kern<<<grid, 64 * WGM * WGN, ldsb>>>(dA, dW, dO, M, N, K, dclk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = fminf(best, ms); sum += ms;
        hipMemcpy(hclk, dclk, 16, hipMemcpyDeviceToHost);
        ghz_sum += (double)hclk[0] / ((double)hclk[1] * 10.0);   // ticks per ns
    }
    hipFree(dclk);
    const double mean = sum / reps, tf = 2.0 * M * N * K / (mean * 1e-3) / 1e12;
    printf("%-34s %d products tile %dx%d waves %dx%d occ %d: best %.3f ms mean %.3f ms = %.1f TFLOP/s f32-equivalent; "
           "vs k_fwd_glds %.3f ms: %.2fx; err vs f64 %.2e relative to max|ref| %.3f; shader clock inside the kernel %.2f GHz "
           "-> bf16 pipe %.2f of its rate at that clock\n",
           label, NPROD, BM, BN, WGM, WGN, occ, best, mean, tf, ref_ms, ref_ms / mean, maxerr / maxref, maxref, ghz_sum / reps,
           // NPROD MFMAs of 16384 FLOP per 16x16x32 output block at 1024 FLOP per cycle and SIMD, 1024 SIMDs
           (2.0 * M * N * K * NPROD / (mean * 1e-3)) / (1024.0 * 1024.0 * (ghz_sum / reps) * 1e9));
}

static void shape(const char *label, int M, int N, int K, double ref_ms) {
    uint16_t *dA, *dW;
    float *dO;
    const int64_t na = (int64_t)M * K, nw = (int64_t)N * K;
    if (hipMalloc(&dA, (size_t)na * 6) != hipSuccess || hipMalloc(&dW, (size_t)nw * 6) != hipSuccess ||
        hipMalloc(&dO, (size_t)M * N * 4) != hipSuccess) { printf("%s: allocation failed\n", label); return; }
    k_gen_split<<<4096, 256>>>(dA, nullptr, na, 1u, 1, 1.4f);                       // post-ReLU activations
    k_gen_split<<<1024, 256>>>(dW, nullptr, nw, 2u, 0, 1.4f / sqrtf((float)K));
    hipDeviceSynchronize();
    printf("== %s: M=%d N=%d K=%d (%.1f GFLOP)\n", label, M, N, K, 2.0 * M * N * K / 1e9);
    if (N % 128 == 0) {
        run<256, 128, 4, 2, 9>(label, dA, dW, dO, M, N, K, ref_ms);
        run<128, 128, 2, 2, 9>(label, dA, dW, dO, M, N, K, ref_ms);
        run<128, 128, 4, 2, 9>(label, dA, dW, dO, M, N, K, ref_ms);
        run<256, 128, 4, 2, 6>(label, dA, dW, dO, M, N, K, ref_ms);
    } else {
        run<256, 64, 4, 2, 9>(label, dA, dW, dO, M, N, K, ref_ms);
        run<256, 64, 8, 1, 9>(label, dA, dW, dO, M, N, K, ref_ms);
        run<128, 64, 2, 2, 9>(label, dA, dW, dO, M, N, K, ref_ms);
        run<128, 64, 4, 1, 9>(label, dA, dW, dO, M, N, K, ref_ms);
        run<256, 64, 4, 2, 6>(label, dA, dW, dO, M, N, K, ref_ms);
    }
    hipFree(dA); hipFree(dW); hipFree(dO);
}

int main(int argc, char **argv) {
    // reference times: k_fwd_glds at the same shapes inside the C2 step (profiles/r02_b_*: fc 126 TFLOP/s = 0.833 ms,
    // conv2 forward n = 32768: 113-118 TFLOP/s = 1.50 ms); override with argv for a same-box number from tools/kbench.py
    const double fc_ms = argc > 1 ? atof(argv[1]) : 0.833, c2_ms = argc > 2 ? atof(argv[2]) : 1.50;
    shape("fc 32768x3136x512", 32768, 512, 3136, fc_ms);
    shape("conv2-as-GEMM 2654208x512x64", 32768 * 81, 64, 512, c2_ms);
    return 0;
}

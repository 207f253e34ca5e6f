// sf_nn.hip — actor-critic network kernels for gfx950 (MI355X): one fp32-MFMA implicit-GEMM family
//   out[M,N] = epilogue( gather(A)[M,K] x B[K,N] )
// instantiated as conv/linear FORWARD, WEIGHT-GRADIENT (split over the reduction = sample/pixel axis, deterministic
// two-stage reduce) and DATA-GRADIENT (gather form, decomposed by stride-parity class so no MFMA work is spent on
// structurally-zero taps).  A dense layer is the 1x1 conv on a 1x1 image, so six reference ops share three kernels.
//
// Numerics: v_mfma_f32_32x32x2_f32 — f32 in, f32 accumulate, bit-equal to an fmaf chain (MI355X_MICROARCH.md), i.e.
// the same precision class as the reference's fp32 MIOpen/rocBLAS path.  No reduced precision anywhere.
//
// Data layout: activations NHWC ([sample][oh][ow][c] == row-major [M, C]); weights K-major [K, Cout] with
// k = (kh*KW + kw)*Cin + c (NHWC input) or k = (c*KH + kh)*KW + kw (raw NCHW u8 observation input, so that four
// consecutive k are four consecutive bytes).  u8 observations are converted ((x - mean) * 1/scale) inside the
// loader: the f32 copy of the observations that the reference materialises (utils/normalize.py:40-70) never exists.
//
// Tile: 256 threads = 4 wavefronts (one per SIMD), block tile BM x BN x 32, LDS image As[32][BM+pad], Bs[32][BN+pad]
// (reduction-major => both MFMA fragment reads are 32 consecutive words, conflict-free), register prefetch of the next
// K-chunk while the current one is in the matrix pipe.
#include "sf_common.h"

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------- exact n/d for n<2^31
struct FastDiv {
    uint32_t d, mul, shr;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    if (d <= 1) { f.mul = 0; f.shr = 0; return f; }
    uint32_t s = 0;
    while ((1u << s) < d) ++s;  // s = ceil(log2 d) >= 1
    f.mul = (uint32_t)((((uint64_t)1) << (31 + s)) / d + 1);
    f.shr = s - 1;
    return f;
}
__host__ __device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv &f) {
    if (f.d <= 1) return n;
    return (uint32_t)(((uint64_t)n * f.mul) >> 32) >> f.shr;
}

extern "C" int sf_selftest_host(void) {  // exercised by the CPU test-suite: the index math everything rests on
    const uint32_t ds[] = {1, 2, 3, 4, 6, 7, 9, 20, 32, 33, 49, 64, 81, 84, 128, 400, 512, 576, 3136, 7056, 28224, 65535, 1000003};
    for (uint32_t d : ds) {
        const FastDiv f = make_fastdiv(d);
        const uint32_t ns[] = {0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, 12345678, 0x7FFFFFFFu, 0x7FFFFFFFu - d, 13107200, 13107199};
        for (uint32_t n : ns) if (fdiv(n, f) != n / d) return -1;
        for (uint32_t n = 0; n < 200000; n += 7) if (fdiv(n, f) != n / d) return -2;
        for (uint32_t n = 0x7FFFFFFFu; n > 0x7FFFFFFFu - 100000; n -= 13) if (fdiv(n, f) != n / d) return -3;
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------- geometry
struct ConvG {
    int Cin, H, W, Cout, KH, KW, S, OH, OW;
    int in_u8, relu, traj_T;
    float sub_mean, inv_scale;
    int K;          // KH*KW*Cin
    int vecA, vecB; // vector (16-byte / 4-byte-of-u8) loads legal for the activation / weight operand
    FastDiv dOHOW, dOW, dCin, dKW, dKHKW, dT, dCout;
};

static ConvG make_geom(const sf_conv_desc *d) {
    ConvG g;
    g.Cin = d->Cin; g.H = d->H; g.W = d->W; g.Cout = d->Cout; g.KH = d->KH; g.KW = d->KW; g.S = d->stride;
    g.OH = d->OH; g.OW = d->OW; g.in_u8 = d->in_u8; g.relu = d->relu; g.traj_T = d->traj_T;
    g.sub_mean = d->sub_mean; g.inv_scale = d->inv_scale;
    g.K = d->KH * d->KW * d->Cin;
    g.vecA = d->in_u8 ? (d->KW % 4 == 0 && d->stride % 4 == 0 && d->W % 4 == 0) : (d->Cin % 4 == 0);
    g.vecB = d->Cout % 4 == 0;
    g.dOHOW = make_fastdiv((uint32_t)(d->OH * d->OW));
    g.dOW = make_fastdiv((uint32_t)d->OW);
    g.dCin = make_fastdiv((uint32_t)d->Cin);
    g.dKW = make_fastdiv((uint32_t)d->KW);
    g.dKHKW = make_fastdiv((uint32_t)(d->KH * d->KW));
    g.dT = make_fastdiv((uint32_t)(d->traj_T > 0 ? d->traj_T : 1));
    g.dCout = make_fastdiv((uint32_t)d->Cout);
    return g;
}

static int check_desc(const sf_conv_desc *d, const char *who) {
    SF_REQUIRE(d, "%s: null descriptor", who);
    SF_REQUIRE(d->Cin > 0 && d->H > 0 && d->W > 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0 && d->stride > 0,
               "%s: bad geometry", who);
    SF_REQUIRE(d->OH == (d->H - d->KH) / d->stride + 1 && d->OW == (d->W - d->KW) / d->stride + 1,
               "%s: OH/OW do not match a VALID (no padding) convolution", who);
    return SF_OK;
}

// input-sample base offset (elements) of logical sample `smp`: optional index gather, optional dataset->trajectory
// slab row mapping (flat index e*T+t  ->  slab row e*(T+1)+t, learner.py:1005-1012 drops column T by *copy*; we
// read the slab in place instead).
__device__ __forceinline__ int64_t sample_base(const ConvG &g, const int32_t *__restrict__ index, int64_t offset,
                                               int64_t stride, uint32_t smp) {
    int64_t d = index ? (int64_t)index[smp] : offset + (int64_t)smp;
    if (g.traj_T > 0) {
        const uint32_t e = fdiv((uint32_t)d, g.dT);
        d = d + (int64_t)e;  // e*(T+1) + (d - e*T)
    }
    return d * stride;
}

// offset (elements) of im2col column k inside one input sample, relative to the patch origin
__device__ __forceinline__ int tap_offset(const ConvG &g, uint32_t k) {
    if (g.in_u8) {  // k = (c*KH + kh)*KW + kw over NCHW bytes
        const uint32_t c = fdiv(k, g.dKHKW), r = k - c * (uint32_t)(g.KH * g.KW);
        const uint32_t kh = fdiv(r, g.dKW), kw = r - kh * (uint32_t)g.KW;
        return (int)((c * (uint32_t)g.H + kh) * (uint32_t)g.W + kw);
    }
    const uint32_t tap = fdiv(k, g.dCin), c = k - tap * (uint32_t)g.Cin;  // k = (kh*KW + kw)*Cin + c over NHWC
    const uint32_t kh = fdiv(tap, g.dKW), kw = tap - kh * (uint32_t)g.KW;
    return (int)((kh * (uint32_t)g.W + kw) * (uint32_t)g.Cin + c);
}
// offset (elements) of output pixel `pix` patch origin inside one input sample
__device__ __forceinline__ int patch_origin(const ConvG &g, uint32_t pix) {
    const uint32_t oh = fdiv(pix, g.dOW), ow = pix - oh * (uint32_t)g.OW;
    if (g.in_u8) return (int)(oh * (uint32_t)g.S * (uint32_t)g.W + ow * (uint32_t)g.S);
    return (int)((oh * (uint32_t)g.S * (uint32_t)g.W + ow * (uint32_t)g.S) * (uint32_t)g.Cin);
}

__device__ __forceinline__ void load_act4(const ConvG &g, const void *__restrict__ in, int64_t base, uint32_t k,
                                          float (&v)[4]) {
    // four consecutive im2col columns k..k+3 (k % 4 == 0) of the patch whose origin is `base`
    if (g.in_u8) {
        const uint8_t *p = reinterpret_cast<const uint8_t *>(in);
        if (g.vecA) {
            const uint32_t w = *reinterpret_cast<const uint32_t *>(p + base + tap_offset(g, k));
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ((float)((w >> (8 * j)) & 0xFFu) - g.sub_mean) * g.inv_scale;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                v[j] = (k + j < (uint32_t)g.K) ? ((float)p[base + tap_offset(g, k + j)] - g.sub_mean) * g.inv_scale : 0.f;
        }
    } else {
        const float *p = reinterpret_cast<const float *>(in);
        if (g.vecA) {
            const float4 w = *reinterpret_cast<const float4 *>(p + base + tap_offset(g, k));
            v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (k + j < (uint32_t)g.K) ? p[base + tap_offset(g, k + j)] : 0.f;
        }
    }
}

// four consecutive columns n..n+3 of row `row` of a row-major [*, N] f32 matrix (weights [K,N], or dY [M,N])
__device__ __forceinline__ void load_row4(const float *__restrict__ p, int64_t row, int n, int N, bool vec,
                                          float (&v)[4]) {
    if (vec && n + 3 < N) {
        const float4 w = *reinterpret_cast<const float4 *>(p + row * N + n);
        v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (n + j < N) ? p[row * N + n + j] : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------- MFMA tile compute
// As: [32][LDA] (reduction-major), Bs: [32][LDB].  Wave (wm, wn) owns rows wm*TM*32.. and cols wn*TN*32..
template <int TM, int TN, int LDA, int LDB>
__device__ __forceinline__ void mma_chunk(const float *__restrict__ As, const float *__restrict__ Bs, int arow0,
                                          int bcol0, int lane, f32x16 (&acc)[TM][TN]) {
    const int i = lane & 31, kh = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 32; kk += 2) {
        float a[TM], b[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) a[tm] = As[(kk + kh) * LDA + arow0 + tm * 32 + i];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) b[tn] = Bs[(kk + kh) * LDB + bcol0 + tn * 32 + i];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
    }
}

// C/D fragment: reg r of lane l holds (row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31)  (cdna_hip_programming.md §3)
#define FRAG_ROW(r, lane) (((r) & 3) + 8 * ((r) >> 2) + 4 * ((lane) >> 5))

template <int BM, int BN, int WM, int WN>
struct Tile {
    static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static constexpr int SA = BM / 32, SB = BN / 32;  // 16-byte load slots per thread per 32-deep chunk
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 waves per block");
};

// ============================================================================================== FORWARD
// rows m = (sample, oh, ow); A reduction-major loads (4 consecutive k per slot), B = weights free-axis-major.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void k_conv_fwd(ConvG g, const void *__restrict__ in, int64_t in_stride,
                                                  const int32_t *__restrict__ index, int64_t offset,
                                                  const float *__restrict__ w, const float *__restrict__ bias,
                                                  float *__restrict__ out, int64_t Mtot) {
    using T = Tile<BM, BN, WM, WN>;
    constexpr int LDA = BM + 1, LDB = BN + 4;
    __shared__ __attribute__((aligned(16))) float As[32 * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[32 * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int N = g.Cout, K = g.K;

    // A slots: row = tid/8 + 32*s, k-quad = tid%8
    const int kq = (tid & 7) * 4;
    int64_t abase[T::SA];
    bool aval[T::SA];
#pragma unroll
    for (int s = 0; s < T::SA; ++s) {
        const int64_t m = m0 + (tid >> 3) + 32 * s;
        aval[s] = m < Mtot;
        abase[s] = 0;
        if (aval[s]) {
            const uint32_t smp = fdiv((uint32_t)m, g.dOHOW), pix = (uint32_t)m - smp * (uint32_t)(g.OH * g.OW);
            abase[s] = sample_base(g, index, offset, in_stride, smp) + patch_origin(g, pix);
        }
    }
    // B slots: F-major: column quad cg, reduction row kk0 + s*(1024/BN)
    constexpr int BG = BN / 4, BROWS = 256 / BG;
    const int bcg = (tid % BG) * 4, bkk0 = tid / BG;

    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int a = 0; a < T::TM; ++a)
#pragma unroll
        for (int b = 0; b < T::TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float ra[T::SA][4], rb[T::SB][4];
    auto gload = [&](int k0) {
#pragma unroll
        for (int s = 0; s < T::SA; ++s) {
            if (aval[s] && k0 + kq < K) load_act4(g, in, abase[s], (uint32_t)(k0 + kq), ra[s]);
            else { ra[s][0] = ra[s][1] = ra[s][2] = ra[s][3] = 0.f; }
        }
#pragma unroll
        for (int s = 0; s < T::SB; ++s) {
            const int k = k0 + bkk0 + s * BROWS;
            if (k < K) load_row4(w, k, n0 + bcg, N, g.vecB, rb[s]);
            else { rb[s][0] = rb[s][1] = rb[s][2] = rb[s][3] = 0.f; }
        }
    };
    gload(0);
    for (int k0 = 0; k0 < K; k0 += 32) {
        __syncthreads();
#pragma unroll
        for (int s = 0; s < T::SA; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) As[(kq + j) * LDA + (tid >> 3) + 32 * s] = ra[s][j];
#pragma unroll
        for (int s = 0; s < T::SB; ++s)
            *reinterpret_cast<float4 *>(&Bs[(bkk0 + s * BROWS) * LDB + bcg]) =
                make_float4(rb[s][0], rb[s][1], rb[s][2], rb[s][3]);
        __syncthreads();
        if (k0 + 32 < K) gload(k0 + 32);
        mma_chunk<T::TM, T::TN, LDA, LDB>(As, Bs, wm * T::TM * 32, wn * T::TN * 32, lane, acc);
    }
    // epilogue: bias + ReLU, NHWC store
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < T::TN; ++tn) {
            const int n = n0 + wn * T::TN * 32 + tn * 32 + (lane & 31);
            const float bv = (n < N && bias) ? bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * T::TM * 32 + tm * 32 + FRAG_ROW(r, lane);
                if (m < Mtot && n < N) {
                    float v = acc[tm][tn][r] + bv;
                    if (g.relu) v = fmaxf(v, 0.f);
                    out[m * N + n] = v;
                }
            }
        }
}

// ============================================================================================== WEIGHT GRADIENT
// dW[k][n] = sum_m col(in)[m][k] * dY[m][n].  GEMM rows = k (BM), cols = n (BN), reduction = m, split over
// gridDim.z contiguous m-ranges; partial tiles go to workspace[z][K][N], k_reduce_partials sums them in fixed order.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void k_conv_wgrad(ConvG g, const void *__restrict__ in, int64_t in_stride,
                                                    const int32_t *__restrict__ index, int64_t offset,
                                                    const float *__restrict__ dy, float *__restrict__ partial,
                                                    int64_t Mtot, int64_t m_per_split) {
    using T = Tile<BM, BN, WM, WN>;
    constexpr int LDA = BM + 4, LDB = BN + 4;
    __shared__ __attribute__((aligned(16))) float As[32 * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[32 * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int k0row = blockIdx.x * BM;  // first weight row (im2col column) of this block
    const int n0 = blockIdx.y * BN;
    const int N = g.Cout, K = g.K;
    const int64_t mbeg = (int64_t)blockIdx.z * m_per_split;
    const int64_t mend = (mbeg + m_per_split < Mtot) ? mbeg + m_per_split : Mtot;

    // A' slots (free-axis-major): 4 consecutive weight rows k for one reduction index m
    constexpr int AG = BM / 4, AROWS = 256 / AG;
    const int acg = (tid % AG) * 4, akk0 = tid / AG;
    const uint32_t ak = (uint32_t)(k0row + acg);
    const bool akval = (int)ak < K;
    const int atap = (akval && g.vecA) ? tap_offset(g, ak) : 0;
    constexpr int BG = BN / 4, BROWS = 256 / BG;
    const int bcg = (tid % BG) * 4, bkk0 = tid / BG;

    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int a = 0; a < T::TM; ++a)
#pragma unroll
        for (int b = 0; b < T::TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float ra[T::SA][4], rb[T::SB][4];
    auto gload = [&](int64_t mc) {
#pragma unroll
        for (int s = 0; s < T::SA; ++s) {
            const int64_t m = mc + akk0 + s * AROWS;
            ra[s][0] = ra[s][1] = ra[s][2] = ra[s][3] = 0.f;
            if (akval && m < mend) {
                const uint32_t smp = fdiv((uint32_t)m, g.dOHOW), pix = (uint32_t)m - smp * (uint32_t)(g.OH * g.OW);
                const int64_t base = sample_base(g, index, offset, in_stride, smp) + patch_origin(g, pix);
                if (g.vecA) {
                    if (g.in_u8) {
                        const uint32_t wv = *reinterpret_cast<const uint32_t *>(
                            reinterpret_cast<const uint8_t *>(in) + base + atap);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            ra[s][j] = ((float)((wv >> (8 * j)) & 0xFFu) - g.sub_mean) * g.inv_scale;
                    } else {
                        const float4 wv = *reinterpret_cast<const float4 *>(
                            reinterpret_cast<const float *>(in) + base + atap);
                        ra[s][0] = wv.x; ra[s][1] = wv.y; ra[s][2] = wv.z; ra[s][3] = wv.w;
                    }
                } else {
                    load_act4(g, in, base, ak, ra[s]);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < T::SB; ++s) {
            const int64_t m = mc + bkk0 + s * BROWS;
            if (m < mend) load_row4(dy, m, n0 + bcg, N, g.vecB, rb[s]);
            else { rb[s][0] = rb[s][1] = rb[s][2] = rb[s][3] = 0.f; }
        }
    };
    if (mbeg < mend) gload(mbeg);
    for (int64_t mc = mbeg; mc < mend; mc += 32) {
        __syncthreads();
#pragma unroll
        for (int s = 0; s < T::SA; ++s)
            *reinterpret_cast<float4 *>(&As[(akk0 + s * AROWS) * LDA + acg]) =
                make_float4(ra[s][0], ra[s][1], ra[s][2], ra[s][3]);
#pragma unroll
        for (int s = 0; s < T::SB; ++s)
            *reinterpret_cast<float4 *>(&Bs[(bkk0 + s * BROWS) * LDB + bcg]) =
                make_float4(rb[s][0], rb[s][1], rb[s][2], rb[s][3]);
        __syncthreads();
        if (mc + 32 < mend) gload(mc + 32);
        mma_chunk<T::TM, T::TN, LDA, LDB>(As, Bs, wm * T::TM * 32, wn * T::TN * 32, lane, acc);
    }
    float *dst = partial + (int64_t)blockIdx.z * K * N;
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < T::TN; ++tn) {
            const int n = n0 + wn * T::TN * 32 + tn * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0row + wm * T::TM * 32 + tm * 32 + FRAG_ROW(r, lane);
                if (k < K && n < N) dst[(int64_t)k * N + n] = acc[tm][tn][r];
            }
        }
}

// column sums of dY [Mtot, N] over the same m-splits -> partial_b[z][N]   (bias gradient)
__global__ __launch_bounds__(256) void k_colsum_partial(const float *__restrict__ dy, float *__restrict__ partial_b,
                                                        int64_t Mtot, int N, int64_t m_per_split) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.y * 64 + lane;
    const int64_t mbeg = (int64_t)blockIdx.x * m_per_split;
    const int64_t mend = (mbeg + m_per_split < Mtot) ? mbeg + m_per_split : Mtot;
    float acc = 0.f;
    if (n < N)
        for (int64_t m = mbeg + wave; m < mend; m += 4) acc += dy[m * N + n];
    red[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && n < N) partial_b[(int64_t)blockIdx.x * N + n] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// out[i] = sum_z partial[z][i], z ascending (deterministic)
__global__ __launch_bounds__(256) void k_reduce_partials(const float *__restrict__ partial, float *__restrict__ out,
                                                         int64_t n, int Z) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < Z; ++z) s += partial[(int64_t)z * n + i];
        out[i] = s;
    }
}

// ============================================================================================== DATA GRADIENT
// din[sample, ih, iw, c] = mask * sum_{kh,kw,n} dY[sample, (ih-kh)/S, (iw-kw)/S, n] * W[(kh,kw,c), n], only taps with
// kh = ih mod S (+ a*S), kw = iw mod S (+ b*S) contribute -> one GEMM per parity class (ph, pw) = blockIdx.z:
// rows m' = (sample, ihh, iww) with ih = ihh*S+ph, cols = c, reduction k' = ((a*KWs + b)*Cout + n).
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void k_conv_dgrad(ConvG g, const float *__restrict__ dy,
                                                    const float *__restrict__ w, const float *__restrict__ in_act,
                                                    float *__restrict__ din, int64_t nsamples) {
    using T = Tile<BM, BN, WM, WN>;
    constexpr int LDA = BM + 1, LDB = BN + 1;
    __shared__ __attribute__((aligned(16))) float As[32 * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[32 * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ph = blockIdx.z / g.S, pw = blockIdx.z % g.S;
    const int Hc = (g.H - ph + g.S - 1) / g.S, Wc = (g.W - pw + g.S - 1) / g.S;
    const int KHs = (g.KH - ph + g.S - 1) / g.S, KWs = (g.KW - pw + g.S - 1) / g.S;
    const int64_t Mc = nsamples * Hc * Wc;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    if (m0 >= Mc || Hc <= 0 || Wc <= 0) return;
    const int n0 = blockIdx.y * BN;  // input-channel tile
    const int Cin = g.Cin, Cout = g.Cout;
    const int Kp = KHs * KWs * Cout;  // reduction length of this class (0 if the class has no taps)
    const bool vec = (Cout % 4) == 0;

    const int kq = (tid & 7) * 4;
    int64_t arow[T::SA];  // first dY row (sample * OH*OW) of the slot's sample
    int aih[T::SA], aiw[T::SA];
    bool aval[T::SA];
    const uint32_t HcWc = (uint32_t)(Hc * Wc);
#pragma unroll
    for (int s = 0; s < T::SA; ++s) {
        const int64_t m = m0 + (tid >> 3) + 32 * s;
        aval[s] = m < Mc;
        arow[s] = 0; aih[s] = 0; aiw[s] = 0;
        if (aval[s]) {
            const uint32_t smp = (uint32_t)m / HcWc;  // m < 2^31 (checked by the launcher)
            const uint32_t pix = (uint32_t)m - smp * HcWc;
            aih[s] = (int)(pix / (uint32_t)Wc);
            aiw[s] = (int)pix - aih[s] * Wc;
            arow[s] = (int64_t)smp * g.OH * g.OW;
        }
    }
    // B' slots (reduction-major): column c = tid/8 + 32*s, 4 consecutive n
    f32x16 acc[T::TM][T::TN];
#pragma unroll
    for (int a = 0; a < T::TM; ++a)
#pragma unroll
        for (int b = 0; b < T::TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    float ra[T::SA][4], rb[T::SB][4];
    auto gload = [&](int k0) {
        const int k = k0 + kq;
        int a = 0, b = 0, n = 0;
        const bool kval = k < Kp;
        if (kval) {
            const int tap = (int)fdiv((uint32_t)k, g.dCout);
            n = k - tap * Cout;
            a = tap / KWs;
            b = tap - a * KWs;
        }
#pragma unroll
        for (int s = 0; s < T::SA; ++s) {
            ra[s][0] = ra[s][1] = ra[s][2] = ra[s][3] = 0.f;
            const int oh = aih[s] - a, ow = aiw[s] - b;
            if (kval && aval[s] && oh >= 0 && oh < g.OH && ow >= 0 && ow < g.OW)
                load_row4(dy, arow[s] + (int64_t)oh * g.OW + ow, n, Cout, vec, ra[s]);
        }
        const int kh = ph + a * g.S, kw = pw + b * g.S;
#pragma unroll
        for (int s = 0; s < T::SB; ++s) {
            const int c = n0 + (tid >> 3) + 32 * s;
            if (kval && c < Cin) load_row4(w, (int64_t)(kh * g.KW + kw) * Cin + c, n, Cout, vec, rb[s]);
            else { rb[s][0] = rb[s][1] = rb[s][2] = rb[s][3] = 0.f; }
        }
    };
    if (Kp > 0) gload(0);
    for (int k0 = 0; k0 < Kp; k0 += 32) {
        __syncthreads();
#pragma unroll
        for (int s = 0; s < T::SA; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) As[(kq + j) * LDA + (tid >> 3) + 32 * s] = ra[s][j];
#pragma unroll
        for (int s = 0; s < T::SB; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) Bs[(kq + j) * LDB + (tid >> 3) + 32 * s] = rb[s][j];
        __syncthreads();
        if (k0 + 32 < Kp) gload(k0 + 32);
        mma_chunk<T::TM, T::TN, LDA, LDB>(As, Bs, wm * T::TM * 32, wn * T::TN * 32, lane, acc);
    }
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < T::TN; ++tn) {
            const int c = n0 + wn * T::TN * 32 + tn * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * T::TM * 32 + tm * 32 + FRAG_ROW(r, lane);
                if (m < Mc && c < Cin) {
                    const uint32_t smp = (uint32_t)m / HcWc, pix = (uint32_t)m - smp * HcWc;
                    const int ihh = (int)(pix / (uint32_t)Wc), iww = (int)pix - ihh * Wc;
                    const int64_t o = (((int64_t)smp * g.H + (ihh * g.S + ph)) * g.W + (iww * g.S + pw)) * Cin + c;
                    float v = acc[tm][tn][r];
                    if (in_act && !(in_act[o] > 0.f)) v = 0.f;
                    din[o] = v;
                }
            }
        }
}

__global__ __launch_bounds__(256) void k_relu_mask(float *__restrict__ gsrc, const float *__restrict__ act, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (!(act[i] > 0.f)) gsrc[i] = 0.f;
}

// ============================================================================================== host launchers
static inline unsigned cdiv64(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

template <int BM, int BN, int WM, int WN>
static void launch_fwd(const ConvG &g, const void *in, int64_t stride, const int32_t *index, int64_t offset,
                       const float *w, const float *bias, float *out, int64_t Mtot, hipStream_t st) {
    dim3 grid(cdiv64(Mtot, BM), cdiv64(g.Cout, BN), 1);
    k_conv_fwd<BM, BN, WM, WN><<<grid, dim3(256), 0, st>>>(g, in, stride, index, offset, w, bias, out, Mtot);
}

extern "C" int sf_conv_fwd(const void *in, int64_t in_sample_stride, const int32_t *index, int64_t offset,
                           const float *w, const float *bias, float *out, int64_t n, const sf_conv_desc *h_desc,
                           void *stream) {
    int rc = check_desc(h_desc, "sf_conv_fwd");
    if (rc) return rc;
    SF_REQUIRE(in && w && out && n > 0, "sf_conv_fwd: bad args");
    const ConvG g = make_geom(h_desc);
    const int64_t Mtot = n * g.OH * g.OW;
    SF_REQUIRE(Mtot < (1LL << 31), "sf_conv_fwd: M=%lld exceeds 2^31 rows; split the batch", (long long)Mtot);
    hipStream_t st = STREAM(stream);
    if (g.Cout <= 32) launch_fwd<128, 32, 4, 1>(g, in, in_sample_stride, index, offset, w, bias, out, Mtot, st);
    else if (Mtot * ((g.Cout + 63) / 64) < 128LL * 1024)
        launch_fwd<64, 64, 2, 2>(g, in, in_sample_stride, index, offset, w, bias, out, Mtot, st);
    else launch_fwd<128, 64, 2, 2>(g, in, in_sample_stride, index, offset, w, bias, out, Mtot, st);
    return sf_launch_status("sf_conv_fwd");
}

// split plan shared by the workspace query and the launcher
struct SplitPlan {
    int Z;
    int64_t m_per_split;
};
static SplitPlan plan_splits(int64_t Mtot, int K, int N, int BM, int BN) {
    const int64_t tiles = (int64_t)((K + BM - 1) / BM) * ((N + BN - 1) / BN);
    const int64_t chunks = (Mtot + 31) / 32;
    int64_t Z = (2048 + tiles - 1) / tiles;  // aim for ~2k blocks
    const int64_t zmax = (chunks + 7) / 8;   // at least 8 chunks (256 reduction rows) per split
    if (Z > zmax) Z = zmax;
    if (Z < 1) Z = 1;
    if (Z > 1024) Z = 1024;
    SplitPlan p;
    p.m_per_split = ((chunks + Z - 1) / Z) * 32;
    p.Z = (int)((Mtot + p.m_per_split - 1) / p.m_per_split);
    return p;
}
static inline int wgrad_bn(int N) { return N <= 32 ? 32 : 64; }

extern "C" int64_t sf_conv_wgrad_workspace(int64_t n, const sf_conv_desc *h_desc) {
    if (!h_desc || n <= 0) return 0;
    const int K = h_desc->KH * h_desc->KW * h_desc->Cin, N = h_desc->Cout;
    const int64_t Mtot = n * h_desc->OH * h_desc->OW;
    const SplitPlan p = plan_splits(Mtot, K, N, 128, wgrad_bn(N));
    return (int64_t)sizeof(float) * p.Z * ((int64_t)K * N + N) + 256;
}

extern "C" int sf_conv_wgrad(const void *in, int64_t in_sample_stride, const int32_t *index, int64_t offset,
                             const float *dout, float *dw, float *db, int64_t n, const sf_conv_desc *h_desc,
                             void *workspace, void *stream) {
    int rc = check_desc(h_desc, "sf_conv_wgrad");
    if (rc) return rc;
    SF_REQUIRE(in && dout && dw && workspace && n > 0, "sf_conv_wgrad: bad args");
    SF_REQUIRE(((uintptr_t)workspace & 15) == 0, "sf_conv_wgrad: workspace must be 16-byte aligned");
    const ConvG g = make_geom(h_desc);
    const int64_t Mtot = n * g.OH * g.OW;
    SF_REQUIRE(Mtot < (1LL << 31), "sf_conv_wgrad: M=%lld exceeds 2^31 rows; split the batch", (long long)Mtot);
    const int K = g.K, N = g.Cout;
    const int BN = wgrad_bn(N);
    const SplitPlan p = plan_splits(Mtot, K, N, 128, BN);
    float *partial_w = reinterpret_cast<float *>(workspace);
    float *partial_b = partial_w + (int64_t)p.Z * K * N;
    hipStream_t st = STREAM(stream);
    dim3 grid(cdiv64(K, 128), cdiv64(N, BN), (unsigned)p.Z);
    if (BN == 32)
        k_conv_wgrad<128, 32, 4, 1><<<grid, dim3(256), 0, st>>>(g, in, in_sample_stride, index, offset, dout, partial_w,
                                                               Mtot, p.m_per_split);
    else
        k_conv_wgrad<128, 64, 2, 2><<<grid, dim3(256), 0, st>>>(g, in, in_sample_stride, index, offset, dout, partial_w,
                                                               Mtot, p.m_per_split);
    const int64_t KN = (int64_t)K * N;
    k_reduce_partials<<<dim3(cdiv64(KN, 256) < 2048 ? cdiv64(KN, 256) : 2048), dim3(256), 0, st>>>(partial_w, dw, KN,
                                                                                                    p.Z);
    if (db) {
        k_colsum_partial<<<dim3((unsigned)p.Z, cdiv64(N, 64)), dim3(256), 0, st>>>(dout, partial_b, Mtot, N,
                                                                                   p.m_per_split);
        k_reduce_partials<<<dim3(cdiv64(N, 256)), dim3(256), 0, st>>>(partial_b, db, N, p.Z);
    }
    return sf_launch_status("sf_conv_wgrad");
}

extern "C" int sf_conv_dgrad(const float *dout, const float *w, const float *in_act, float *din, int64_t n,
                             const sf_conv_desc *h_desc, void *stream) {
    int rc = check_desc(h_desc, "sf_conv_dgrad");
    if (rc) return rc;
    SF_REQUIRE(dout && w && din && n > 0, "sf_conv_dgrad: bad args");
    SF_REQUIRE(!h_desc->in_u8, "sf_conv_dgrad: the observation layer has no data gradient");
    SF_REQUIRE(h_desc->Cout % 4 == 0 || (h_desc->KH == 1 && h_desc->KW == 1),
               "sf_conv_dgrad: Cout must be a multiple of 4 for spatial kernels");
    const ConvG g = make_geom(h_desc);
    const int Hc = (g.H + g.S - 1) / g.S, Wc = (g.W + g.S - 1) / g.S;  // largest parity class
    const int64_t Mc = n * Hc * Wc;
    SF_REQUIRE(n * g.H * g.W < (1LL << 31), "sf_conv_dgrad: too many rows; split the batch");
    hipStream_t st = STREAM(stream);
    const unsigned classes = (unsigned)(g.S * g.S);
    if (g.Cin <= 32) {
        dim3 grid(cdiv64(Mc, 128), cdiv64(g.Cin, 32), classes);
        k_conv_dgrad<128, 32, 4, 1><<<grid, dim3(256), 0, st>>>(g, dout, w, in_act, din, n);
    } else if (Mc * ((g.Cin + 63) / 64) < 128LL * 1024) {
        dim3 grid(cdiv64(Mc, 64), cdiv64(g.Cin, 64), classes);
        k_conv_dgrad<64, 64, 2, 2><<<grid, dim3(256), 0, st>>>(g, dout, w, in_act, din, n);
    } else {
        dim3 grid(cdiv64(Mc, 128), cdiv64(g.Cin, 64), classes);
        k_conv_dgrad<128, 64, 2, 2><<<grid, dim3(256), 0, st>>>(g, dout, w, in_act, din, n);
    }
    return sf_launch_status("sf_conv_dgrad");
}

// ---- dense layers = 1x1 conv on a 1x1 image
static sf_conv_desc linear_desc(int K, int N, int relu) {
    sf_conv_desc d;
    d.Cin = K; d.H = 1; d.W = 1; d.Cout = N; d.KH = 1; d.KW = 1; d.stride = 1; d.OH = 1; d.OW = 1;
    d.in_u8 = 0; d.relu = relu; d.traj_T = 0; d.sub_mean = 0.f; d.inv_scale = 1.f;
    return d;
}

extern "C" int sf_linear_fwd(const float *in, const float *w, const float *bias, float *out, int64_t M, int K, int N,
                             int relu, void *stream) {
    SF_REQUIRE(M > 0 && K > 0 && N > 0, "sf_linear_fwd: bad shape");
    const sf_conv_desc d = linear_desc(K, N, relu);
    return sf_conv_fwd(in, K, nullptr, 0, w, bias, out, M, &d, stream);
}
extern "C" int64_t sf_linear_wgrad_workspace(int64_t M, int K, int N) {
    const sf_conv_desc d = linear_desc(K, N, 0);
    return sf_conv_wgrad_workspace(M, &d);
}
extern "C" int sf_linear_wgrad(const float *in, const float *dout, float *dw, float *db, int64_t M, int K, int N,
                               void *workspace, void *stream) {
    SF_REQUIRE(M > 0 && K > 0 && N > 0, "sf_linear_wgrad: bad shape");
    const sf_conv_desc d = linear_desc(K, N, 0);
    return sf_conv_wgrad(in, K, nullptr, 0, dout, dw, db, M, &d, workspace, stream);
}
extern "C" int sf_linear_dgrad(const float *dout, const float *w, const float *in_act, float *din, int64_t M, int K,
                               int N, void *stream) {
    SF_REQUIRE(M > 0 && K > 0 && N > 0, "sf_linear_dgrad: bad shape");
    const sf_conv_desc d = linear_desc(K, N, 0);
    return sf_conv_dgrad(dout, w, in_act, din, M, &d, stream);
}

extern "C" int sf_relu_mask(float *g, const float *act, int64_t n, void *stream) {
    SF_REQUIRE(g && act && n >= 0, "sf_relu_mask: bad args");
    if (n == 0) return SF_OK;
    const unsigned blocks = cdiv64(n, 256) < 4096 ? cdiv64(n, 256) : 4096;
    k_relu_mask<<<dim3(blocks), dim3(256), 0, STREAM(stream)>>>(g, act, n);
    return sf_launch_status("sf_relu_mask");
}

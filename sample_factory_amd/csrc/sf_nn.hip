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
//
// Loaders are compile-time specialised (MODE) and BRANCH-FREE: out-of-range rows / columns / reduction indices load
// from a clamped, always-valid address and are zeroed with a select, so the compiler issues every global load of a
// chunk back-to-back and waits once (the first version branched per load and hipcc serialised them with vmcnt(0)).
#include "sf_common.h"
#include <stdlib.h>
#ifndef SF_GLDS_ABLATE
#define SF_GLDS_ABLATE 0  // timing experiments only (sf_nn_glds.h)
#endif
#include <type_traits>

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
    const uint32_t q = (uint32_t)(((uint64_t)n * f.mul) >> 32) >> f.shr;
    return f.d <= 1 ? n : q;  // select, not a branch: keeps the loaders straight-line
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
    int vecA, vecB; // vector (16-byte / 4-byte-of-u8) loads legal for the activation / [*,Cout] operands
    const float *nmu, *nrstd;  // sf_conv_fwd_norm / sf_conv_wgrad_norm: the observation normaliser's f32 tables, else NULL
    FastDiv dOHOW, dOW, dCin, dKW, dKHKW, dT, dCout;
    FastDiv dHcWc[16], dWc[16], dKWs[16];  // data-gradient stride-parity classes (S*S <= 16)
};

static ConvG make_geom(const sf_conv_desc *d) {
    ConvG g;
    g.Cin = d->Cin; g.H = d->H; g.W = d->W; g.Cout = d->Cout; g.KH = d->KH; g.KW = d->KW; g.S = d->stride;
    g.OH = d->OH; g.OW = d->OW; g.in_u8 = d->in_u8; g.relu = d->relu; g.traj_T = d->traj_T;
    g.sub_mean = d->sub_mean; g.inv_scale = d->inv_scale;
    g.nmu = nullptr; g.nrstd = nullptr;
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
    for (int z = 0; z < 16; ++z) {
        const int S = d->stride, ph = z / S, pw = z % S;
        const int Hc = (d->H - ph + S - 1) / S, Wc = (d->W - pw + S - 1) / S, KWs = (d->KW - pw + S - 1) / S;
        const bool ok = z < S * S && Hc > 0 && Wc > 0;
        g.dHcWc[z] = make_fastdiv(ok ? (uint32_t)(Hc * Wc) : 1u);
        g.dWc[z] = make_fastdiv(ok ? (uint32_t)Wc : 1u);
        g.dKWs[z] = make_fastdiv(ok && KWs > 0 ? (uint32_t)KWs : 1u);
    }
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

// loader specialisations
constexpr int MODE_F32 = 0;      // f32 NHWC activations, Cin % 4 == 0, Cout % 4 == 0: 16-byte loads everywhere
constexpr int MODE_U8 = 1;       // raw u8 NCHW observation, KW/stride/W % 4 == 0: 4-byte loads of 4 pixels
constexpr int MODE_GENERIC = 2;  // any geometry: scalar, bounds-checked (slow; odd shapes only)

// input-sample base offset (elements) of logical sample `smp`: optional index gather, optional dataset->trajectory
// slab row mapping (flat index e*T+t  ->  slab row e*(T+1)+t, learner.py:1005-1012 drops column T by *copy*; we
// read the slab in place instead).
__device__ __forceinline__ int64_t sample_base(const ConvG &g, const int32_t *__restrict__ index, int64_t offset,
                                               int64_t stride, uint32_t smp) {
    int64_t d = index ? (int64_t)index[smp] : offset + (int64_t)smp;
    if (g.traj_T > 0) d += (int64_t)fdiv((uint32_t)d, g.dT);  // e*(T+1) + (d - e*T)
    return d * stride;
}

// offset (elements) of im2col column k inside one input sample, relative to the patch origin
template <bool U8>
__device__ __forceinline__ int tap_offset(const ConvG &g, uint32_t k) {
    if (U8) {  // k = (c*KH + kh)*KW + kw over NCHW bytes
        const uint32_t c = fdiv(k, g.dKHKW), r = k - c * (uint32_t)(g.KH * g.KW);
        const uint32_t kh = fdiv(r, g.dKW), kw = r - kh * (uint32_t)g.KW;
        return (int)((c * (uint32_t)g.H + kh) * (uint32_t)g.W + kw);
    }
    const uint32_t tap = fdiv(k, g.dCin), c = k - tap * (uint32_t)g.Cin;  // k = (kh*KW + kw)*Cin + c over NHWC
    const uint32_t kh = fdiv(tap, g.dKW), kw = tap - kh * (uint32_t)g.KW;
    return (int)((kh * (uint32_t)g.W + kw) * (uint32_t)g.Cin + c);
}
// offset (elements) of output pixel `pix` patch origin inside one input sample
template <bool U8>
__device__ __forceinline__ int patch_origin(const ConvG &g, uint32_t pix) {
    const uint32_t oh = fdiv(pix, g.dOW), ow = pix - oh * (uint32_t)g.OW;
    const uint32_t o = oh * (uint32_t)g.S * (uint32_t)g.W + ow * (uint32_t)g.S;
    return (int)(U8 ? o : o * (uint32_t)g.Cin);
}

// activations (desc.relu carries the kind): 0 none, 1 ReLU, 2 tanh, 3 ELU(alpha=1) — model/model_utils.py:27-35
__device__ __forceinline__ float act_fwd(float x, int kind) {
    if (kind == 1) return fmaxf(x, 0.f);
    if (kind == 2) return tanhf(x);
    if (kind == 3) return x > 0.f ? x : expm1f(x);
    return x;
}
// derivative expressed through the activation OUTPUT y (what the backward chain has at hand)
__device__ __forceinline__ float act_bwd(float y, int kind) {
    if (kind == 1) return y > 0.f ? 1.f : 0.f;
    if (kind == 2) return 1.f - y * y;
    if (kind == 3) return y > 0.f ? 1.f : y + 1.f;
    return 1.f;
}

// v * act'(y) with the kind known at compile time (KIND 0: no activation, < 0: run-time kind)
template <int KIND>
__device__ __forceinline__ float act_bwd_mul(float v, float y, int kind) {
    if (KIND == 0) return v;
    if (KIND == 1) return y > 0.f ? v : 0.f;
    return v * act_bwd(y, kind);
}

// Raw (unconverted, unmasked) operand quads.  The value is NOT touched between the global load and the LDS store of
// the next iteration, so the loads stay in flight across the whole MFMA phase (a select right after the load made
// hipcc wait for the data before the first MFMA — no overlap at all).
template <int MODE>
struct ARaw {
    float4 v;
};
template <>
struct ARaw<MODE_U8> {
    uint32_t v;
};

// four consecutive im2col columns k..k+3 (k % 4 == 0, k < K) of the patch whose origin is `base`
template <int MODE>
__device__ __forceinline__ ARaw<MODE> load_act_raw(const ConvG &g, const void *__restrict__ in, int64_t base,
                                                   uint32_t k, bool ok) {
    ARaw<MODE> r;
    if constexpr (MODE == MODE_U8) {
        r.v = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(in) + base + tap_offset<true>(g, k));
    } else if constexpr (MODE == MODE_F32) {
        r.v = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(in) + base + tap_offset<false>(g, k));
    } else {
        float x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool okj = ok && (k + j < (uint32_t)g.K);
            const uint32_t kj = okj ? k + j : 0u;
            if (g.in_u8)
                x[j] = ((float)reinterpret_cast<const uint8_t *>(in)[base + tap_offset<true>(g, kj)] - g.sub_mean) * g.inv_scale;
            else
                x[j] = reinterpret_cast<const float *>(in)[base + tap_offset<false>(g, kj)];
            x[j] = okj ? x[j] : 0.f;
        }
        r.v = make_float4(x[0], x[1], x[2], x[3]);
    }
    return r;
}
template <int MODE>
__device__ __forceinline__ float4 act_finish(const ConvG &g, const ARaw<MODE> &r, bool ok) {
    if constexpr (MODE == MODE_U8) {
        float x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float t = ((float)((r.v >> (8 * j)) & 0xFFu) - g.sub_mean) * g.inv_scale;
            x[j] = ok ? t : 0.f;
        }
        return make_float4(x[0], x[1], x[2], x[3]);
    } else {
        return make_float4(ok ? r.v.x : 0.f, ok ? r.v.y : 0.f, ok ? r.v.z : 0.f, ok ? r.v.w : 0.f);
    }
}

// four consecutive columns n..n+3 of row `row` of a row-major [*, N] f32 matrix (row must be a valid row index even
// when the result is going to be masked).  VEC: N % 4 == 0 and n % 4 == 0 -> one 16-byte load; n >= N reads column 0.
template <bool VEC>
__device__ __forceinline__ float4 load_row_raw(const float *__restrict__ p, int64_t row, int n, int N) {
    if constexpr (VEC) {
        return *reinterpret_cast<const float4 *>(p + row * N + (n < N ? n : 0));
    } else {
        float x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = p[row * N + (n + j < N ? n + j : 0)];
        return make_float4(x[0], x[1], x[2], x[3]);
    }
}
__device__ __forceinline__ float4 row_finish(const float4 &r, bool ok, int n, int N) {
    return make_float4((ok && n < N) ? r.x : 0.f, (ok && n + 1 < N) ? r.y : 0.f, (ok && n + 2 < N) ? r.z : 0.f,
                       (ok && n + 3 < N) ? r.w : 0.f);
}

// ---------------------------------------------------------------------------------------------- MFMA tile compute
// As: [32][LDA] (reduction-major), Bs: [32][LDB].  Wave (wm, wn) owns rows wm*TM*32.. and cols wn*TN*32..
template <int TM, int TN, int LDA, int LDB>
__device__ __forceinline__ void mma_chunk(const float *__restrict__ As, const float *__restrict__ Bs, int arow0,
                                          int bcol0, int lane, f32x16 (&acc)[TM][TN]) {
    // Fragments are double-buffered in registers, two k-steps ahead: the ds_reads of step kk+2 are issued before the
    // MFMAs of step kk, so the LDS latency hides under 2*TM*TN*64 cycles of matrix-pipe time instead of stalling
    // every MFMA pair behind an lgkmcnt(0) (what hipcc emitted for the naive loop).
    const int i = lane & 31, kh = lane >> 5;
    const float *ap = As + kh * LDA + arow0 + i;
    const float *bp = Bs + kh * LDB + bcol0 + i;
    float a[3][TM], b[3][TN];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) a[p][tm] = ap[(2 * p) * LDA + tm * 32];
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) b[p][tn] = bp[(2 * p) * LDB + tn * 32];
    }
#pragma unroll
    for (int st = 0; st < 16; ++st) {
        const int cur = st % 3, nxt = (st + 2) % 3;
        if (st + 2 < 16) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) a[nxt][tm] = ap[(2 * (st + 2)) * LDA + tm * 32];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) b[nxt][tn] = bp[(2 * (st + 2)) * LDB + tn * 32];
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the prefetch reads ABOVE this step's MFMAs
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[cur][tm], b[cur][tn], acc[tm][tn], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// C/D fragment: reg r of lane l holds (row = (r&3) + 8*(r>>2) + 4*(l>>5), col = l&31)  (cdna_hip_programming.md §3)
#define FRAG_ROW(r, lane) (((r) & 3) + 8 * ((r) >> 2) + 4 * ((lane) >> 5))

// act_fwd with the kind known at compile time (KIND < 0: run-time kind)
template <int KIND>
__device__ __forceinline__ float act_fwd_c(float x, int kind) {
    if (KIND == 0) return x;
    if (KIND == 1) return fmaxf(x, 0.f);
    return act_fwd(x, kind);
}

// Store one wave's TM x TN accumulator fragments (32x32x2 layout: row = (r&3) + 8*(r>>2) + 4*(lane>>5), col =
// lane&31) as act(acc + bias) into a row-major [rows][N] matrix.  ob = uniform pointer to the tile's (0, 0);
// voff = this lane's (4*(lane>>5))*N + (lane&31); rows_left / cols_left = number of existing rows / columns counted
// from THIS LANE's first row / column (FULL: the whole tile exists, no per-element test); bias0 = bias + tile column,
// lcol = lane&31.
template <int TM, int TN, int KIND, bool FULL>
__device__ __forceinline__ void store_fwd_tile(const f32x16 (&acc)[TM][TN], float *__restrict__ ob, uint32_t voff, int N,
                                               int rows_left, int cols_left, const float *__restrict__ bias0, int lcol,
                                               int kind) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const bool cok = FULL || tn * 32 < cols_left;
        const float bv = (bias0 && cok) ? bias0[tn * 32 + lcol] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = tm * 32 + (r & 3) + 8 * (r >> 2);
                float *p = ob + (int64_t)rr * N + tn * 32;  // uniform
                const float v = act_fwd_c<KIND>(acc[tm][tn][r] + bv, kind);
#if SF_GLDS_ABLATE & 4
                if (v == 1.2345e30f) p[voff] = v;  // timing experiment: the value is computed, the store never happens
#else
                if (FULL || (cok && rr < rows_left)) p[voff] = v;
#endif
            }
    }
}

// The same tile as v * act'(y): the data gradient of a linear layer is this forward GEMM on dY with the (untransposed)
// weight matrix, finished with the derivative of the activation that produced the layer's input y (mk: pointer to the
// tile's (0, 0) inside y, same [rows][N] layout).  KIND 0: no mask, 1: ReLU, < 0: run-time kind.
template <int TM, int TN, int KIND, bool FULL>
__device__ __forceinline__ void store_dgrad_tile(const f32x16 (&acc)[TM][TN], float *__restrict__ ob,
                                                 const float *__restrict__ mk, uint32_t voff, int N, int rows_left,
                                                 int cols_left, int kind) {
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const bool cok = FULL || tn * 32 < cols_left;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            float y[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {  // all 16 loads in flight before the first store
                const int rr = tm * 32 + (r & 3) + 8 * (r >> 2);
                const bool ok = FULL || (cok && rr < rows_left);
                y[r] = (KIND != 0 && ok) ? (mk + (int64_t)rr * N + tn * 32)[voff] : 1.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = tm * 32 + (r & 3) + 8 * (r >> 2);
                const float v = act_bwd_mul<KIND>(acc[tm][tn][r], y[r], kind);
                if (FULL || (cok && rr < rows_left)) (ob + (int64_t)rr * N + tn * 32)[voff] = v;
            }
        }
    }
}


template <int BM, int BN, int WM, int WN>
struct Tile {
    static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    static constexpr int SA = BM / 32, SB = BN / 32;  // 16-byte load slots per thread per 32-deep chunk
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 waves per block");
};

#define ZERO_ACC(acc)                                                                                         \
    _Pragma("unroll") for (int a_ = 0; a_ < T::TM; ++a_) _Pragma("unroll") for (int b_ = 0; b_ < T::TN; ++b_) \
        _Pragma("unroll") for (int r_ = 0; r_ < 16; ++r_) acc[a_][b_][r_] = 0.f

// ============================================================================================== FORWARD
// rows m = (sample, oh, ow); A reduction-major loads (4 consecutive k per slot), B = weights free-axis-major.
template <int BM, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__(256) void k_conv_fwd(ConvG g, const void *__restrict__ in, int64_t in_stride,
                                                  const int32_t *__restrict__ index, int64_t offset,
                                                  const float *__restrict__ w, const float *__restrict__ bias,
                                                  float *__restrict__ out, int64_t Mtot, int k_per_split,
                                                  float *__restrict__ partial) {
    using T = Tile<BM, BN, WM, WN>;
    constexpr bool U8 = MODE == MODE_U8;
    constexpr bool VECB = MODE != MODE_GENERIC;
    constexpr int LDA = BM + 1, LDB = BN + 4;
    __shared__ __attribute__((aligned(16))) float As[32 * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[32 * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int N = g.Cout, K = g.K;
    // split-K (small grids only): this block reduces k in [kbeg, kend) and writes a raw partial tile
    const int kbeg = blockIdx.z * k_per_split;
    const int kend = (kbeg + k_per_split < K) ? kbeg + k_per_split : K;

    // A slots.  f32 NHWC input: k-quad fastest across lanes (row = tid/8 + 32*s, k-quad = tid%8): 8 lanes read one
    // 128-byte run of channels.  Raw u8 NCHW input: row fastest (row = tid%32 + 32*s, k-quad = tid/32): consecutive
    // output pixels are `stride` bytes apart, so 32 lanes x 4 bytes cover one contiguous 128-byte span of the frame.
    const int arow = U8 ? (tid & 31) : (tid >> 3);
    const int kq = (U8 ? (tid >> 5) : (tid & 7)) * 4;
    int64_t abase[T::SA];
    bool aval[T::SA];
#pragma unroll
    for (int s = 0; s < T::SA; ++s) {
        const int64_t m = m0 + arow + 32 * s;
        aval[s] = m < Mtot;
        const uint32_t mm = aval[s] ? (uint32_t)m : 0u;
        const uint32_t smp = fdiv(mm, g.dOHOW), pix = mm - smp * (uint32_t)(g.OH * g.OW);
        const bool u8 = MODE == MODE_GENERIC ? g.in_u8 != 0 : U8;
        abase[s] = sample_base(g, index, offset, in_stride, smp) +
                   (u8 ? patch_origin<true>(g, pix) : patch_origin<false>(g, pix));
    }
    // B slots: F-major: column quad cg, reduction row kk0 + s*(1024/BN)
    constexpr int BG = BN / 4, BROWS = 256 / BG;
    const int bcg = (tid % BG) * 4, bkk0 = tid / BG;

    f32x16 acc[T::TM][T::TN];
    ZERO_ACC(acc);

    ARaw<MODE> ra[T::SA];
    float4 rb[T::SB];
    bool aok = false, bok[T::SB];
    auto gload = [&](int k0) {
        const int ka = k0 + kq;
        aok = ka < kend;
        const uint32_t kc = aok ? (uint32_t)ka : (uint32_t)kbeg;
#pragma unroll
        for (int s = 0; s < T::SA; ++s) ra[s] = load_act_raw<MODE>(g, in, abase[s], kc, aok && aval[s]);
#pragma unroll
        for (int s = 0; s < T::SB; ++s) {
            const int k = k0 + bkk0 + s * BROWS;
            bok[s] = k < kend;
            rb[s] = load_row_raw<VECB>(w, bok[s] ? k : kbeg, n0 + bcg, N);
        }
    };
    // Rows >= Mtot and columns >= N only feed accumulator entries that are never stored, and the clamped addresses
    // read finite data, so with full K-chunks (every layer of the Nature CNN: K % 32 == 0) nothing needs masking.
    const bool kfull = ((kend - kbeg) & 31) == 0 && MODE != MODE_GENERIC;
    auto lstore = [&]() {
#pragma unroll
        for (int s = 0; s < T::SA; ++s) {
            const float4 v = act_finish<MODE>(g, ra[s], kfull || (aok && aval[s]));
            As[(kq + 0) * LDA + arow + 32 * s] = v.x;
            As[(kq + 1) * LDA + arow + 32 * s] = v.y;
            As[(kq + 2) * LDA + arow + 32 * s] = v.z;
            As[(kq + 3) * LDA + arow + 32 * s] = v.w;
        }
#pragma unroll
        for (int s = 0; s < T::SB; ++s)
            *reinterpret_cast<float4 *>(&Bs[(bkk0 + s * BROWS) * LDB + bcg]) =
                kfull ? rb[s] : row_finish(rb[s], bok[s], n0 + bcg, N);
    };
    gload(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        __syncthreads();  // every wave is done reading the previous chunk's LDS image
        lstore();         // first use of the prefetched registers: the loads had the whole MFMA phase to land
        __syncthreads();
        if (k0 + 32 < kend) gload(k0 + 32);
        mma_chunk<T::TM, T::TN, LDA, LDB>(As, Bs, wm * T::TM * 32, wn * T::TN * 32, lane, acc);
    }
    // epilogue: bias + ReLU, NHWC store (split-K: raw partial, finished by k_splitk_finish)
    float *dst = partial ? partial + (int64_t)blockIdx.z * Mtot * N : out;
    const bool fin = partial == nullptr;
    float *ob = dst + (m0 + wm * T::TM * 32) * N + (n0 + wn * T::TN * 32);
    const int rows_left = (int)min((int64_t)(T::TM * 32), Mtot - m0 - wm * T::TM * 32) - 4 * (lane >> 5);
    const int cols_left = N - (n0 + wn * T::TN * 32) - (lane & 31);
    const uint32_t voff = (uint32_t)(4 * (lane >> 5)) * (uint32_t)N + (uint32_t)(lane & 31);
    const bool full = m0 + BM <= Mtot && n0 + BN <= N;
    const float *b0 = (fin && bias) ? bias + n0 + wn * T::TN * 32 : nullptr;
    if (!fin) {
        if (full) store_fwd_tile<T::TM, T::TN, 0, true>(acc, ob, voff, N, rows_left, cols_left, nullptr, lane & 31, 0);
        else store_fwd_tile<T::TM, T::TN, 0, false>(acc, ob, voff, N, rows_left, cols_left, nullptr, lane & 31, 0);
    } else if (g.relu == 1) {
        if (full) store_fwd_tile<T::TM, T::TN, 1, true>(acc, ob, voff, N, rows_left, cols_left, b0, lane & 31, 1);
        else store_fwd_tile<T::TM, T::TN, 1, false>(acc, ob, voff, N, rows_left, cols_left, b0, lane & 31, 1);
    } else {
        store_fwd_tile<T::TM, T::TN, -1, false>(acc, ob, voff, N, rows_left, cols_left, b0, lane & 31, g.relu);
    }
}

// out[m][n] = act(sum_z partial[z][m][n] + bias[n]), z ascending (deterministic)
__global__ __launch_bounds__(256) void k_splitk_finish(const float *__restrict__ partial,
                                                       const float *__restrict__ bias, float *__restrict__ out,
                                                       int64_t MN, int N, int Z, int relu) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < MN; i += (int64_t)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int z = 0; z < Z; ++z) s += partial[(int64_t)z * MN + i];
        if (bias) s += bias[(int)(i % N)];
        out[i] = act_fwd(s, relu);
    }
}

// ============================================================================================== WEIGHT GRADIENT
// dW[k][n] = sum_m col(in)[m][k] * dY[m][n].  GEMM rows = k (BM), cols = n (BN), reduction = m, split over
// gridDim.z contiguous m-ranges; partial tiles go to workspace[z][K][N], k_reduce_partials sums them in fixed order.
// The bias gradient (column sums of dY) is accumulated from the staged dY tile by the blockIdx.x == 0 blocks.
template <int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__(256) void k_conv_wgrad(ConvG g, const void *__restrict__ in, int64_t in_stride,
                                                    const int32_t *__restrict__ index, int64_t offset,
                                                    const float *__restrict__ dy, float *__restrict__ partial,
                                                    float *__restrict__ partial_b, int64_t Mtot,
                                                    int64_t m_per_split) {
    constexpr int BM = 128;
    using T = Tile<BM, BN, WM, WN>;
    constexpr bool U8 = MODE == MODE_U8;
    constexpr bool VECB = MODE != MODE_GENERIC;
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

    // A' slots (free-axis-major): 4 consecutive weight rows k for one reduction index m.
    // f32 NHWC input: k-group fastest across lanes (32 lanes read 512 contiguous bytes of channels); slot s holds
    //   k-group tid%32 of reduction row tid/32 + 8*s.
    // raw u8 NCHW input: m fastest (32 consecutive output pixels = one contiguous 128-byte span); slot s holds k-group
    //   tid/32 + 8*s of reduction row tid%32, so the m -> (sample, pixel) decomposition is done once per chunk.
    int kg[T::SA], tapo[T::SA];
    bool kval[T::SA];
#pragma unroll
    for (int s = 0; s < T::SA; ++s) {
        kg[s] = (U8 ? (tid >> 5) + 8 * s : (tid & 31)) * 4;
        const int k = k0row + kg[s];
        kval[s] = k < K;
        const uint32_t kc = kval[s] ? (uint32_t)k : 0u;
        tapo[s] = MODE == MODE_GENERIC ? 0 : (U8 ? tap_offset<true>(g, kc) : tap_offset<false>(g, kc));
    }
    constexpr int BG = BN / 4, BROWS = 256 / BG;
    const int bcg = (tid % BG) * 4, bkk0 = tid / BG;

    f32x16 acc[T::TM][T::TN];
    ZERO_ACC(acc);

    ARaw<MODE> ra[T::SA];
    float4 rb[T::SB];
    bool aok[T::SA], bok[T::SB];
    auto patch_base = [&](int64_t m, bool ok) -> int64_t {
        const uint32_t mm = ok ? (uint32_t)m : (uint32_t)mbeg;
        const uint32_t smp = fdiv(mm, g.dOHOW), pix = mm - smp * (uint32_t)(g.OH * g.OW);
        const bool u8 = MODE == MODE_GENERIC ? g.in_u8 != 0 : U8;
        return sample_base(g, index, offset, in_stride, smp) +
               (u8 ? patch_origin<true>(g, pix) : patch_origin<false>(g, pix));
    };
    auto gload = [&](int64_t mc) {
        if constexpr (U8) {
            const int64_t m = mc + (tid & 31);
            const bool mok = m < mend;
            const int64_t base = patch_base(m, mok);
#pragma unroll
            for (int s = 0; s < T::SA; ++s) {
                ra[s].v = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint8_t *>(in) + base + tapo[s]);
                aok[s] = mok && kval[s];
            }
        } else {
#pragma unroll
            for (int s = 0; s < T::SA; ++s) {
                const int64_t m = mc + (tid >> 5) + 8 * s;
                const bool mok = m < mend;
                aok[s] = mok && kval[s];
                const int64_t base = patch_base(m, mok);
                if constexpr (MODE == MODE_F32)
                    ra[s].v = *reinterpret_cast<const float4 *>(reinterpret_cast<const float *>(in) + base + tapo[s]);
                else
                    ra[s] = load_act_raw<MODE_GENERIC>(g, in, base, (uint32_t)(k0row + kg[s]), aok[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < T::SB; ++s) {
            const int64_t m = mc + bkk0 + s * BROWS;
            bok[s] = m < mend;
            rb[s] = load_row_raw<VECB>(dy, bok[s] ? m : mbeg, n0 + bcg, N);
        }
    };
    const bool do_colsum = partial_b != nullptr && blockIdx.x == 0 && tid < BN;  // bias gradient: column sums of dY
    float colacc = 0.f;
    if (mbeg < mend) gload(mbeg);
    for (int64_t mc = mbeg; mc < mend; mc += 32) {
        __syncthreads();
#pragma unroll
        for (int s = 0; s < T::SA; ++s) {
            const int lrow = U8 ? (tid & 31) : (tid >> 5) + 8 * s;  // reduction index (m) inside the chunk
            // rows k >= K are never stored and reduction rows >= mend are annihilated by the zeroed dY rows below,
            // so the activation operand is stored unmasked on the vector paths (clamped loads read finite data)
            *reinterpret_cast<float4 *>(&As[lrow * LDA + kg[s]]) =
                act_finish<MODE>(g, ra[s], MODE != MODE_GENERIC || aok[s]);
        }
        const bool tail = mc + 32 > mend;
#pragma unroll
        for (int s = 0; s < T::SB; ++s)
            *reinterpret_cast<float4 *>(&Bs[(bkk0 + s * BROWS) * LDB + bcg]) =
                (VECB && !tail) ? rb[s] : row_finish(rb[s], bok[s], n0 + bcg, N);
        __syncthreads();
        if (mc + 32 < mend) gload(mc + 32);
        if (do_colsum) {
#pragma unroll 8
            for (int kk = 0; kk < 32; ++kk) colacc += Bs[kk * LDB + tid];
        }
        mma_chunk<T::TM, T::TN, LDA, LDB>(As, Bs, wm * T::TM * 32, wn * T::TN * 32, lane, acc);
    }
    if (do_colsum && n0 + tid < N) partial_b[(int64_t)blockIdx.z * N + n0 + tid] = colacc;
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

// out[i] = sum_z partial[z][i] in a FIXED order (deterministic): eight interleaved accumulators (z mod 8) keep eight
// loads in flight per lane — the single-accumulator loop was latency-bound (Z up to 1024 dependent loads per lane).
__global__ __launch_bounds__(256) void k_reduce_partials(const float *__restrict__ partial, float *__restrict__ out,
                                                         int64_t n, int Z) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int z = 0;
        for (; z + 8 <= Z; z += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += partial[(int64_t)(z + j) * n + i];
        }
        for (int j = 0; z < Z; ++z, ++j) s[j] += partial[(int64_t)z * n + i];
        out[i] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    }
}

// Many partials, few outputs (persistent weight-gradient kernels write one partial per work-group: 256 - 512 of them for
// a 27 x 64 ... 256 x 32 result): the loop above is then a handful of work-groups walking Z dependent rounds.  Here a
// work-group takes 32 outputs x 8 groups of consecutive partials; every thread sums its group as above (ascending z,
// eight interleaved accumulators), the eight group sums meet in LDS and are added in a fixed tree.  Deterministic.
__global__ __launch_bounds__(256) void k_reduce_partials_tree(const float *__restrict__ partial, float *__restrict__ out,
                                                              int64_t n, int Z) {
    __shared__ float sm[8][33];
    const int li = threadIdx.x & 31, zg = threadIdx.x >> 5;
    const int64_t i = (int64_t)blockIdx.x * 32 + li;
    const int zper = (Z + 7) / 8, z0 = zg * zper, z1 = z0 + zper < Z ? z0 + zper : Z;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < n) {
        int z = z0;
        for (; z + 8 <= z1; z += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += partial[(int64_t)(z + j) * n + i];
        }
        for (int j = 0; z < z1; ++z, ++j) s[j] += partial[(int64_t)z * n + i];
    }
    sm[zg][li] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (zg == 0 && i < n)
        out[i] = ((sm[0][li] + sm[1][li]) + (sm[2][li] + sm[3][li])) + ((sm[4][li] + sm[5][li]) + (sm[6][li] + sm[7][li]));
}
static void launch_reduce_partials(const float *partial, float *out, int64_t n, int Z, hipStream_t st) {
    static const int tree_on = getenv("SF_REDUCE_TREE") ? atoi(getenv("SF_REDUCE_TREE")) : 1;
    if (tree_on && Z >= 64 && (n + 255) / 256 < 256)  // fewer loop work-groups than CUs and a long walk each
        k_reduce_partials_tree<<<dim3((unsigned)((n + 31) / 32)), dim3(256), 0, st>>>(partial, out, n, Z);
    else
        k_reduce_partials<<<dim3((unsigned)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048)), dim3(256), 0, st>>>(partial, out, n, Z);
}

// ============================================================================================== DATA GRADIENT
// din[sample, ih, iw, c] = mask * sum_{kh,kw,n} dY[sample, (ih-kh)/S, (iw-kw)/S, n] * W[(kh,kw,c), n], only taps with
// kh = ih mod S (+ a*S), kw = iw mod S (+ b*S) contribute -> one GEMM per parity class (ph, pw) = blockIdx.z:
// rows m' = (sample, ihh, iww) with ih = ihh*S+ph, cols = c, reduction k' = ((a*KWs + b)*Cout + n).
template <int BM, int BN, int WM, int WN, bool VEC>
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

    const int kq = (tid & 7) * 4;
    const FastDiv fHW = g.dHcWc[blockIdx.z], fW = g.dWc[blockIdx.z], fKWs = g.dKWs[blockIdx.z];
    // per-slot dY row of tap (0,0) as a 32-bit index (launcher guarantees rows*Cout < 2^31), plus (ihh, iww)
    int arow[T::SA], aih[T::SA], aiw[T::SA];
    bool aval[T::SA];
    const uint32_t HcWc = (uint32_t)(Hc * Wc);
#pragma unroll
    for (int s = 0; s < T::SA; ++s) {
        const int64_t m = m0 + (tid >> 3) + 32 * s;
        aval[s] = m < Mc;
        const uint32_t mm = aval[s] ? (uint32_t)m : 0u;
        const uint32_t smp = fdiv(mm, fHW), pix = mm - smp * HcWc;
        aih[s] = (int)fdiv(pix, fW);
        aiw[s] = (int)pix - aih[s] * Wc;
        arow[s] = (int)(smp * (uint32_t)(g.OH * g.OW)) + aih[s] * g.OW + aiw[s];
    }
    const bool kfull = (Kp & 31) == 0;  // every chunk full: the weight operand needs no masking
    f32x16 acc[T::TM][T::TN];
    ZERO_ACC(acc);

    float4 ra[T::SA], rb[T::SB];
    bool aok[T::SA], bok = true;
    int ncol = 0;
    auto gload = [&](int k0) {
        const int k = k0 + kq;
        const bool kval = k < Kp;
        const int kc = kval ? k : 0;
        const int tap = (int)fdiv((uint32_t)kc, g.dCout);
        const int n = kc - tap * Cout;
        ncol = n;
        const int a = (int)fdiv((uint32_t)tap, fKWs), b = tap - a * KWs;
        const int drow = a * g.OW + b;  // tap (a,b) reads dY pixel (ihh - a, iww - b)
#pragma unroll
        for (int s = 0; s < T::SA; ++s) {
            const int oh = aih[s] - a, ow = aiw[s] - b;
            aok[s] = kval && aval[s] && oh >= 0 && oh < g.OH && ow >= 0 && ow < g.OW;
            const int row = aok[s] ? arow[s] - drow : 0;
            ra[s] = load_row_raw<VEC>(dy, row, n, Cout);
        }
        const int wrow = ((ph + a * g.S) * g.KW + (pw + b * g.S)) * Cin;
        bok = kval;
#pragma unroll
        for (int s = 0; s < T::SB; ++s) {
            const int c = n0 + (tid >> 3) + 32 * s;
            rb[s] = load_row_raw<VEC>(w, wrow + (c < Cin ? c : 0), n, Cout);  // columns >= Cin are never stored
        }
    };
    if (Kp > 0) gload(0);
    for (int k0 = 0; k0 < Kp; k0 += 32) {
        __syncthreads();
#pragma unroll
        for (int s = 0; s < T::SA; ++s) {
            const float4 v = row_finish(ra[s], aok[s], ncol, Cout);
            As[(kq + 0) * LDA + (tid >> 3) + 32 * s] = v.x;
            As[(kq + 1) * LDA + (tid >> 3) + 32 * s] = v.y;
            As[(kq + 2) * LDA + (tid >> 3) + 32 * s] = v.z;
            As[(kq + 3) * LDA + (tid >> 3) + 32 * s] = v.w;
        }
#pragma unroll
        for (int s = 0; s < T::SB; ++s) {
            const float4 v = (VEC && kfull) ? rb[s] : row_finish(rb[s], bok, ncol, Cout);
            Bs[(kq + 0) * LDB + (tid >> 3) + 32 * s] = v.x;
            Bs[(kq + 1) * LDB + (tid >> 3) + 32 * s] = v.y;
            Bs[(kq + 2) * LDB + (tid >> 3) + 32 * s] = v.z;
            Bs[(kq + 3) * LDB + (tid >> 3) + 32 * s] = v.w;
        }
        __syncthreads();
        if (k0 + 32 < Kp) gload(k0 + 32);
        mma_chunk<T::TM, T::TN, LDA, LDB>(As, Bs, wm * T::TM * 32, wn * T::TN * 32, lane, acc);
    }
    const int akind = g.relu;
#pragma unroll
    for (int tm = 0; tm < T::TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + wm * T::TM * 32 + tm * 32 + FRAG_ROW(r, lane);
            const bool mok = m < Mc;
            const uint32_t mm = mok ? (uint32_t)m : 0u;
            const uint32_t smp = fdiv(mm, fHW), pix = mm - smp * HcWc;
            const int ihh = (int)fdiv(pix, fW), iww = (int)pix - ihh * Wc;
            const int64_t obase = (((int64_t)smp * g.H + (ihh * g.S + ph)) * g.W + (iww * g.S + pw)) * Cin;
#pragma unroll
            for (int tn = 0; tn < T::TN; ++tn) {
                const int c = n0 + wn * T::TN * 32 + tn * 32 + (lane & 31);
                if (mok && c < Cin) {
                    float v = acc[tm][tn][r];
                    if (in_act) v *= act_bwd(in_act[obase + c], akind);  // kind of the activation that produced in_act
                    din[obase + c] = v;
                }
            }
        }
}

__global__ __launch_bounds__(256) void k_relu_mask(float *__restrict__ gsrc, const float *__restrict__ act, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (!(act[i] > 0.f)) gsrc[i] = 0.f;
}

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
#include "sf_nn_glds.h"
#include "sf_nn_img.h"
#include "sf_nn_u8.h"
#include "sf_nn_wimg.h"
#include "sf_nn_narrow.h"

// ============================================================================================== host launchers
static inline unsigned cdiv64(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

// resident blocks per CU of a 256-thread kernel (registers + static LDS), for grid-quantisation decisions
template <class KernelT>
static int occupancy_of(KernelT kern, int threads = 256, size_t dyn_lds = 0) {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, dyn_lds) != hipSuccess || nb < 1) nb = 2;
    (void)hipGetLastError();
    return nb;
}


static int num_cus() {
    static int v = 0;
    if (!v) {
        int dev = 0;
        hipDeviceProp_t p;
        v = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount : 256;
    }
    return v;
}

static int pick_mode(const ConvG &g) {
    if (!g.vecA || !g.vecB) return MODE_GENERIC;
    return g.in_u8 ? MODE_U8 : MODE_F32;
}

// geometry contract of k_conv_u8_img<2, 4, 5, 16, *>
static bool conv1_img_ok(const ConvG &g, int mode, int64_t n) {
    return mode == MODE_U8 && n >= 256 && g.Cin == 4 && g.H == 84 && g.W == 84 && g.KH == 8 && g.KW == 8 && g.S == 4 &&
           g.Cout <= 32;
}

// k_conv1_u8_bf16: same geometry; the A operand must be exact in bf16: pixel - mean an integer of at most 8 bits
static bool conv1_bf16_ok(const ConvG &g, int mode, int64_t n) {
    static const int on = getenv("SF_CONV1_BF16") ? atoi(getenv("SF_CONV1_BF16")) : 1;
    return on && conv1_img_ok(g, mode, n) && g.sub_mean == floorf(g.sub_mean) && g.sub_mean >= 0.f && g.sub_mean <= 255.f;
}

// forward launch plan: tile config + optional split-K when the natural grid cannot fill 256 CUs several times over
struct FwdPlan {
    int cfg;  // 0: 128x32 (4x1 waves), 1: 128x64 (2x2), 2: 64x64 (2x2)
    int splits, k_per_split;
};
static FwdPlan plan_fwd(int64_t Mtot, int N, int K, int64_t ws_floats) {
    FwdPlan p;
    p.splits = 1;
    p.k_per_split = (K + 31) / 32 * 32;
    p.cfg = N <= 32 ? 0 : 1;
    const int BM = 128, BN = p.cfg == 0 ? 32 : 64;
    int64_t blocks = ((Mtot + BM - 1) / BM) * ((N + BN - 1) / BN);
    if (blocks < 768 && Mtot <= 64) { p.cfg = 2; blocks = ((Mtot + 63) / 64) * ((N + 63) / 64); }
    if (blocks < 768 && K >= 512) {
        int s = (int)((1024 + blocks - 1) / blocks);
        const int smax = K / 256;  // keep >= 8 chunks per split
        if (s > smax) s = smax;
        if (s > 16) s = 16;
        if (s > 1 && (int64_t)s * Mtot * N <= ws_floats) {
            const int chunks = (K + 31) / 32;
            p.k_per_split = ((chunks + s - 1) / s) * 32;
            p.splits = (K + p.k_per_split - 1) / p.k_per_split;
        }
    }
    return p;
}

extern "C" int64_t sf_conv_fwd_workspace(int64_t n, const sf_conv_desc *h_desc) {
    if (!h_desc || n <= 0) return 0;
    const int K = h_desc->KH * h_desc->KW * h_desc->Cin, N = h_desc->Cout;
    const int64_t Mtot = n * h_desc->OH * h_desc->OW;
    const FwdPlan p = plan_fwd(Mtot, N, K, (int64_t)1 << 60);
    return p.splits > 1 ? (int64_t)sizeof(float) * p.splits * Mtot * N + 256 : 0;
}

#define FWD_LAUNCH(BM, BN, WM, WN, MODE)                                                                   \
    k_conv_fwd<BM, BN, WM, WN, MODE><<<dim3(cdiv64(Mtot, BM), cdiv64(g.Cout, BN), Z), dim3(256), 0, st>>>( \
        g, in, in_sample_stride, index, offset, w, bias, out, Mtot, p.k_per_split, partial)
#define FWD_BY_MODE(BM, BN, WM, WN)                                    \
    do {                                                               \
        if (mode == MODE_F32) FWD_LAUNCH(BM, BN, WM, WN, MODE_F32);    \
        else if (mode == MODE_U8) FWD_LAUNCH(BM, BN, WM, WN, MODE_U8); \
        else FWD_LAUNCH(BM, BN, WM, WN, MODE_GENERIC);                 \
    } while (0)

// ReLU sign-bit masks (sf_conv_fwd_relu_mask / sf_conv_wgrad_relu_mask): the layers whose forward AND weight-gradient
// kernels can record / consume them — conv1 on raw u8 frames on the exact-product bf16 kernels, 32 output channels, ReLU.
// SF_RELU_MASK=0 switches the path off (A/B: the data gradient below reads the activation again).
static bool relu_mask_ok(const sf_conv_desc *d, int64_t n) {
    static const int on = getenv("SF_RELU_MASK") ? atoi(getenv("SF_RELU_MASK")) : 1;
    if (!on || !d->in_u8 || d->relu != 1 || d->Cout != 32) return false;
    const ConvG g = make_geom(d);
    return conv1_bf16_ok(g, pick_mode(g), n);
}
extern "C" int sf_conv_relu_mask_supported(int64_t n, const sf_conv_desc *h_desc) {
    return h_desc && n > 0 && check_desc(h_desc, "sf_conv_relu_mask_supported") == 0 && relu_mask_ok(h_desc, n) ? 1 : 0;
}

static int conv_fwd_impl(const void *in, int64_t in_sample_stride, const int32_t *index, int64_t offset, const float *w,
                         const float *bias, float *out, uint32_t *relu_mask, int64_t n, const sf_conv_desc *h_desc,
                         void *workspace, int64_t workspace_bytes, void *stream) {
    int rc = check_desc(h_desc, "sf_conv_fwd");
    if (rc) return rc;
    SF_REQUIRE(in && w && out && n > 0, "sf_conv_fwd: bad args");
    SF_REQUIRE(((uintptr_t)workspace & 15) == 0, "sf_conv_fwd: workspace must be 16-byte aligned");
    const ConvG g = make_geom(h_desc);
    const int64_t Mtot = n * g.OH * g.OW;
    SF_REQUIRE(Mtot < (1LL << 31), "sf_conv_fwd: M=%lld exceeds 2^31 rows; split the batch", (long long)Mtot);
    int mode = pick_mode(g);
    if (mode != MODE_GENERIC) {  // vector loads also need aligned bases
        const bool al = h_desc->in_u8 ? (((uintptr_t)in & 3) == 0 && in_sample_stride % 4 == 0)
                                      : (((uintptr_t)in & 15) == 0 && in_sample_stride % 4 == 0);
        if (!al || ((uintptr_t)w & 15) != 0) mode = MODE_GENERIC;
    }
    hipStream_t st = STREAM(stream);
    // Nature-CNN conv1 on raw frames: strip-image kernel (bytes converted once into an f32 LDS image, im2col read
    // out of LDS).  Geometry contract of the <2,4,5,16> instantiation: K = 256, 2*4*OW rows = 10 fragments.
    static const int img_on = getenv("SF_CONV1_IMG") ? atoi(getenv("SF_CONV1_IMG")) : 1;
    // ... and on the bf16 matrix pipe with exact products (sf_nn_u8.h) when (pixel - mean) is an integer of <= 8 bits
    if (conv1_bf16_ok(g, mode, n) && ((uintptr_t)in & 3) == 0 && in_sample_stride % 4 == 0) {
        // [SMP][Cin][RS][WP] bf16 strip image + (whole-line stores) the staging tile [SMP][80][36] f32
        const unsigned img_bytes = 2u * 4u * 20u * (unsigned)SF_CONV1_WP * (unsigned)sizeof(uint16_t);
        // persistent grid: SF_CONV1_WGS work-groups per CU (default 2 = what the register budget of __launch_bounds__(256, 2) admits)
        static const int wgs_per_cu = getenv("SF_CONV1_WGS") ? atoi(getenv("SF_CONV1_WGS")) : 2;
        static const int occ_dbg = getenv("SF_DEBUG_OCC") ? atoi(getenv("SF_DEBUG_OCC")) : 0;
        if (occ_dbg)
            fprintf(stderr, "k_conv1_u8_bf16 occupancy: %d (dword stores, %u B LDS) / %d (whole-line stores, %u B LDS) work-groups per CU\n",
                    occupancy_of(k_conv1_u8_bf16<false>, 256, img_bytes), img_bytes,
                    occupancy_of(k_conv1_u8_bf16_w<false>, 256, img_bytes + 2u * 80u * 36u * 4u), img_bytes + 2u * 80u * 36u * 4u);
        const int64_t npairs = cdiv64(n, 2), resident = (int64_t)num_cus() * wgs_per_cu;
        const unsigned grid_q = (unsigned)(npairs < resident ? npairs : resident);
        // SF_CONV1_WIDE (default 1): whole-line output stores through an LDS staging tile (sf_nn_u8.h) — N == 32, aligned output
        static const int wide_on = getenv("SF_CONV1_WIDE") ? atoi(getenv("SF_CONV1_WIDE")) : 1;
        const bool wide = wide_on && g.Cout == 32 && ((uintptr_t)out & 15) == 0;
        const unsigned lds_bytes = img_bytes + (wide ? 2u * 80u * 36u * (unsigned)sizeof(float) : 0u);
#define CONV1_BF16_LAUNCH(KERN)                                                                                       \
    KERN<<<dim3(grid_q), dim3(256), lds_bytes, st>>>(g, reinterpret_cast<const uint8_t *>(in), in_sample_stride, index, \
                                                    offset, w, bias, out, relu_mask, (int)n)
        if (g.sub_mean != 0.f) {
            if (wide) CONV1_BF16_LAUNCH(k_conv1_u8_bf16_w<true>);
            else CONV1_BF16_LAUNCH(k_conv1_u8_bf16<true>);
        } else {
            if (wide) CONV1_BF16_LAUNCH(k_conv1_u8_bf16_w<false>);
            else CONV1_BF16_LAUNCH(k_conv1_u8_bf16<false>);
        }
#undef CONV1_BF16_LAUNCH
        return sf_launch_status("sf_conv_fwd");
    }
    // only the kernel above records ReLU sign bits: every other path would leave the mask unwritten for the
    // weight-gradient kernel to consume (misaligned operands drop to MODE_GENERIC without it)
    SF_REQUIRE(relu_mask == nullptr,
               "sf_conv_fwd_relu_mask: operands not eligible for the sign-bit kernel (sf_conv_relu_mask_supported, 4-byte "
               "aligned frames and sample stride, 16-byte aligned weights)");
    if (img_on && conv1_img_ok(g, mode, n) && ((uintptr_t)in & 3) == 0 && in_sample_stride % 4 == 0) {
        const unsigned lds_bytes = (unsigned)(2 * 4 * 20 * 84 * sizeof(float));  // [SMP][Cin][RS][W] f32
        // persistent work-groups: as many as are resident at once (two per CU: 53.8 KB of LDS each), each walks the
        // sample pairs b, b + grid, ...
        static int bpc = 0;
        if (!bpc) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&bpc, k_conv_u8_img<2, 4, 5, 16, false>, 256, lds_bytes) !=
                    hipSuccess || bpc < 1)
                bpc = 2;
            (void)hipGetLastError();
        }
        const int64_t npairs = cdiv64(n, 2);
        const unsigned grid_img = (unsigned)(npairs < (int64_t)num_cus() * bpc ? npairs : (int64_t)num_cus() * bpc);
        if (g.sub_mean != 0.f)
            k_conv_u8_img<2, 4, 5, 16, true><<<dim3(grid_img), dim3(256), lds_bytes, st>>>(
                g, reinterpret_cast<const uint8_t *>(in), in_sample_stride, index, offset, w, bias, out, (int)n);
        else
            k_conv_u8_img<2, 4, 5, 16, false><<<dim3(grid_img), dim3(256), lds_bytes, st>>>(
                g, reinterpret_cast<const uint8_t *>(in), in_sample_stride, index, offset, w, bias, out, (int)n);
        return sf_launch_status("sf_conv_fwd");
    }
    const FwdPlan p = plan_fwd(Mtot, g.Cout, g.K, workspace ? workspace_bytes / (int64_t)sizeof(float) : 0);
    float *partial = p.splits > 1 ? reinterpret_cast<float *>(workspace) : nullptr;
    const unsigned Z = (unsigned)p.splits;
    // 256-row tiles: measured +5 % for N = 32 (conv1: twice the MFMAs per barrier), -3..-15 % for N = 64 (occupancy)
    const bool big32 = p.splits == 1 && Mtot >= 256 * 2048;
    if (p.cfg == 0) { if (big32) FWD_BY_MODE(256, 32, 4, 1); else FWD_BY_MODE(128, 32, 4, 1); }
    else if (p.cfg == 1) FWD_BY_MODE(128, 64, 2, 2);
    else FWD_BY_MODE(64, 64, 2, 2);
    if (partial) {
        const int64_t MN = Mtot * g.Cout;
        k_splitk_finish<<<dim3(cdiv64(MN, 256) < 4096 ? cdiv64(MN, 256) : 4096), dim3(256), 0, st>>>(
            partial, bias, out, MN, g.Cout, p.splits, g.relu);
    }
    return sf_launch_status("sf_conv_fwd");
}

extern "C" int sf_conv_fwd(const void *in, int64_t in_sample_stride, const int32_t *index, int64_t offset,
                           const float *w, const float *bias, float *out, int64_t n, const sf_conv_desc *h_desc,
                           void *workspace, int64_t workspace_bytes, void *stream) {
    return conv_fwd_impl(in, in_sample_stride, index, offset, w, bias, out, nullptr, n, h_desc, workspace, workspace_bytes,
                         stream);
}
extern "C" int sf_conv_fwd_relu_mask(const void *in, int64_t in_sample_stride, const int32_t *index, int64_t offset,
                                     const float *w, const float *bias, float *out, uint32_t *relu_mask, int64_t n,
                                     const sf_conv_desc *h_desc, void *stream) {
    SF_REQUIRE(relu_mask && ((uintptr_t)relu_mask & 3) == 0, "sf_conv_fwd_relu_mask: relu_mask must be a 4-byte aligned device buffer");
    SF_REQUIRE(h_desc && n > 0 && relu_mask_ok(h_desc, n) && ((uintptr_t)in & 3) == 0 && in_sample_stride % 4 == 0,
               "sf_conv_fwd_relu_mask: unsupported layer / launch (see sf_conv_relu_mask_supported)");
    return conv_fwd_impl(in, in_sample_stride, index, offset, w, bias, out, relu_mask, n, h_desc, nullptr, 0, stream);
}

// ---- LDS-image forward (sf_nn_img.h): compile-time geometries (Cin, H, W, K, S, fragments per step, wave sets, output
// rows per unit).  Nature-CNN conv3 (64 x 9 x 9, 3x3 stride 1, whole images): two independent persistent work-groups
// per CU, +5 % (n = 32768) / +8 % (n = 4096) over k_fwd_glds.  conv2 (32 x 20 x 20, 4x4 stride 2) was measured in two
// forms and stays on k_fwd_glds: whole 51 KB images — X(32, 20, 20, 4, 2, 2, 2, 9) — allow only ONE
// work-group per CU whose 8 waves all stop at the same block barrier (110 vs 117 TFLOP/s); strips of 3 output rows —
// X(32, 20, 20, 4, 2, 2, 1, 3), 20 KB per strip, two work-groups per CU — reach 115: a third more image bytes (strip
// overlap), 2-way bank conflicts where a fragment wraps to the next output row, fewer MFMAs per block step.
#ifndef SF_IMG_TMF
#define SF_IMG_TMF 2  // 16-row fragments per wave and block step of the conv3 LDS-image forward (experiment switch)
#endif
#define IMG_FWD_GEOMS(X) X(64, 9, 9, 3, 1, SF_IMG_TMF, 1, 7)
static int img_fwd_index(const ConvG &g, int64_t n) {
    static const int on = getenv("SF_FWD_IMG") ? atoi(getenv("SF_FWD_IMG")) : 1;
    if (!on || g.Cout != 64 || g.KH != g.KW || n < 512) return -1;
    int idx = 0;
#define X(CIN, HH, WW, KS, ST, TMF, WS, R)                                                       \
    if (g.Cin == CIN && g.H == HH && g.W == WW && g.KH == KS && g.S == ST) return idx;           \
    ++idx;
    IMG_FWD_GEOMS(X)
#undef X
    return -1;
}
static bool launch_img_fwd(const ConvG &g, const float *in, int64_t in_stride, const float *wt, const float *bias,
                           float *out, int64_t n, hipStream_t st) {
    const int which = img_fwd_index(g, n);
    if (which < 0) return false;
    int idx = 0;
#define X(CIN, HH, WW, KS, ST, TMF, WS, R)                                                                           \
    if (which == idx) {                                                                                              \
        /* persistent: as many work-groups as fit on the chip at once */                                            \
        static const int bpc = occupancy_of(k_fwd_img<CIN, HH, WW, KS, ST, TMF, WS, R>, 256 * WS);                      \
        if (getenv("SF_DEBUG_IMG")) fprintf(stderr, "k_fwd_img R=%d: %d work-groups per CU\n", R, bpc);                 \
        k_fwd_img<CIN, HH, WW, KS, ST, TMF, WS, R><<<dim3(num_cus() * (bpc > 0 ? bpc : 1)), dim3(256 * WS), 0, st>>>(   \
            in, in_stride, wt, bias, out, (int)n, g.relu);                                                           \
    }                                                                                                                \
    ++idx;
    IMG_FWD_GEOMS(X)
#undef X
    return true;
}

// ---- glds forward (pre-transposed weights)
static bool glds_fwd_ok(const sf_conv_desc *d) {
    return !d->in_u8 && d->Cin % 32 == 0 && d->traj_T == 0;
}
// Launch plan of the LDS-DMA forward.  Grids of >= 768 128x64 tiles run unsplit (tile width by grid quantisation).
// A wide layer with a long reduction and too few rows to fill the chip (the fc layer of a rollout step: 4096 x 512,
// K = 3136 -> 128 tiles of 128x128) is split along K so that ~2 blocks land on every CU; the slices go through
// k_splitk_finish (ascending z: deterministic).  Measured at n = 4096: register-staged split-K kernel 140 us, this
// plan 4 x 128 blocks 127 + 10 us.  Where 64x64 tiles alone give two work-groups per CU (>= 512 tiles: exactly that fc
// launch) the reduction runs unsplit on k_fwd_glds<64, 64>: 118-121 us and no partial sums (step -0.4 ms, three
// alternations on one box, profiles/r04_d_fc64_ab.log; 128x64 tiles x 2 slices and a third pipeline stage: +-0).  SF_GLDS_FC64=0: the split plan.
struct GldsFwdPlan {
    bool ok, wide, sq64;  // sq64: 64x64 tiles, unsplit
    int Z, k_per_split;
};
static GldsFwdPlan plan_fwd_t(int64_t Mtot, int N, int K) {
    static const int cfg = getenv("SF_GLDS_CFG") ? atoi(getenv("SF_GLDS_CFG")) : 0;
    static const int split_on = getenv("SF_GLDS_SPLITK") ? atoi(getenv("SF_GLDS_SPLITK")) : 1;
    static const int occ64 = occupancy_of(k_fwd_glds<128, 64, 2, 2, 2>), occ128 = occupancy_of(k_fwd_glds<128, 128, 2, 2, 2>);
    GldsFwdPlan p;
    p.ok = false; p.wide = false; p.sq64 = false; p.Z = 1; p.k_per_split = (K + 31) / 32 * 32;
    static const int fc64 = getenv("SF_GLDS_FC64") ? atoi(getenv("SF_GLDS_FC64")) : 1;
    const int64_t t64 = cdiv64(Mtot, 128) * (int64_t)cdiv64(N, 64), t128 = cdiv64(Mtot, 128) * (int64_t)cdiv64(N, 128);
    // (wide outputs of moderate height — the recurrent projection of one rollout step, 2048 x 512 x 2048: 512 tiles,
    //  two per CU — also beat the register-staged kernel: 75 -> measured in profiles/r02_c5_*; the GRU's 2048 x 512 x
    //  1536 is 384 tiles)
    static const int wide_min = getenv("SF_GLDS_WIDE_MIN") ? atoi(getenv("SF_GLDS_WIDE_MIN")) : 384;
    // Launches of a few hundred tiles — the per-split inference launches of a host-env run: conv2 at n = 512 is 324 tiles
    // of 128 x 64, the fc layer 4 x 8 tiles of 64 x 64 — used to fall to the register-staged kernel.  Measured at
    // n = 512 / 1024 (profiles/r05_b_kbench_small_n.log): conv2 47.6 / 84.3 us there, 40.1 / 61.4 us on 128 x 64 LDS-DMA
    // tiles, 34.4 / 61.7 us on 64 x 64 tiles (SF_GLDS_SMALL64 = rows/64 from which they are used; 0 = off); the fc layer
    // 170 / 52 us -> 26 / 39 us on 64 x 64 tiles split along K (SF_GLDS_SPLIT64 below).  SF_GLDS_MIN_TILES: 128 x 64
    // tiles from which the unsplit 128-row plan is used (unchanged: 768).
    static const int min_tiles = getenv("SF_GLDS_MIN_TILES") ? atoi(getenv("SF_GLDS_MIN_TILES")) : 768;
    static const int small64 = getenv("SF_GLDS_SMALL64") ? atoi(getenv("SF_GLDS_SMALL64")) : 256;
    static const int force64 = getenv("SF_GLDS_FORCE64") ? atoi(getenv("SF_GLDS_FORCE64")) : 0;  // experiment switch
    if (force64 && N == 64 && t64 <= force64) { p.ok = true; p.sq64 = true; return p; }
    if (small64 && t64 < 768 && N == 64 && K >= 256 && cdiv64(Mtot, 64) >= small64) {
        p.ok = true; p.sq64 = true;  // narrow layer, few rows: 64-row tiles double the work-groups on the chip
        return p;
    }
    if (t64 >= min_tiles || (t64 >= wide_min && N >= 512 && K >= 256)) {
        p.ok = true;
        if (N >= 128 && cfg == 0) {  // efficiency = rounds / ceil(rounds) with the kernel's own occupancy
            const double u64 = (double)t64 / (256.0 * occ64), u128 = (double)t128 / (256.0 * occ128);
            const double e64 = u64 / (double)(int64_t)(u64 + 0.999999), e128 = u128 / (double)(int64_t)(u128 + 0.999999);
            p.wide = e128 * 1.03 >= e64;  // 64x64 wave tiles: fewer LDS reads and DMA instructions per MFMA
        }
        if (cfg == 2) p.wide = true;
        return p;
    }
    if (fc64 == 1 && N >= 128 && K >= 1024 && cdiv64(Mtot, 64) * (int64_t)cdiv64(N, 64) >= 512) {
        p.ok = true; p.sq64 = true;  // two 64x64 work-groups per CU, the whole reduction in one pass, no partial sums
        return p;
    }
    // a wide layer on very few rows (the fc layer of a 512-sample inference launch: 4 x 8 tiles of 64 x 64): 64 x 64
    // tiles split along K until ~2 work-groups per CU exist.  SF_GLDS_SPLIT64=<min 64x64 tiles> (0 = off)
    static const int split64 = getenv("SF_GLDS_SPLIT64") ? atoi(getenv("SF_GLDS_SPLIT64")) : 32;
    const int64_t t6464 = cdiv64(Mtot, 64) * (int64_t)cdiv64(N, 64);
    if (split64 && split_on && N >= 128 && K >= 1024 && t6464 >= split64 && t6464 < 512) {
        int z = (int)((512 + t6464 - 1) / t6464);
        const int zmax = K / 256;
        z = z > zmax ? zmax : z;
        z = z > 16 ? 16 : z;
        if (z > 1) {
            const int chunks = (K + 31) / 32;
            p.k_per_split = ((chunks + z - 1) / z) * 32;
            p.Z = (K + p.k_per_split - 1) / p.k_per_split;
            if (p.Z > 1) {
                p.ok = true; p.sq64 = true;
                return p;
            }
            p.Z = 1; p.k_per_split = (K + 31) / 32 * 32;
        }
    }
    if (split_on && N >= 128 && K >= 1024 && t128 >= 32) {
        const int slots = 256 * occ128;
        int z = (int)((slots + t128 - 1) / t128);
        const int zmax = K / 256;  // >= 8 chunks per slice
        z = z > zmax ? zmax : z;
        z = z > 16 ? 16 : z;
        if (z > 1) {
            const int chunks = (K + 31) / 32;
            p.k_per_split = ((chunks + z - 1) / z) * 32;
            p.Z = (K + p.k_per_split - 1) / p.k_per_split;
            p.ok = p.Z > 1;
            p.wide = true;
        }
    }
    return p;
}
static bool small_linear_wgrad_ok(const sf_conv_desc *d) {
    static const int on = getenv("SF_LINEAR_NARROW") ? atoi(getenv("SF_LINEAR_NARROW")) : 1;
    return on && !d->in_u8 && d->traj_T == 0 && d->KH == 1 && d->KW == 1 && d->H == 1 && d->W == 1 && d->Cout <= 64 &&
           d->Cin <= 64;
}
// narrow linear layers (the heads): one wave per 16 rows, operands straight from memory (sf_nn_narrow.h)
static bool narrow_fwd_ok(const sf_conv_desc *d, int64_t n) {
    static const int on = getenv("SF_LINEAR_NARROW") ? atoi(getenv("SF_LINEAR_NARROW")) : 1;
    return on && !d->in_u8 && d->traj_T == 0 && d->KH == 1 && d->KW == 1 && d->H == 1 && d->W == 1 && d->Cout <= 32 &&
           d->Cin % 16 == 0 && n < (1 << 30);
}
extern "C" int sf_conv_fwd_t_supported(int64_t n, const sf_conv_desc *h_desc) {
    if (h_desc && n > 0 && narrow_fwd_ok(h_desc, n)) return 1;
    if (!h_desc || n <= 0 || !glds_fwd_ok(h_desc)) return 0;
    // small grids keep the split-K register-staged kernel (a 128-row tile grid must fill 256 CUs a few times over)
    const int64_t Mtot = n * h_desc->OH * h_desc->OW;
    if (img_fwd_index(make_geom(h_desc), n) >= 0) return 1;
    return plan_fwd_t(Mtot, h_desc->Cout, h_desc->KH * h_desc->KW * h_desc->Cin).ok ? 1 : 0;
}
extern "C" int64_t sf_conv_fwd_t_workspace(int64_t n, const sf_conv_desc *h_desc) {
    if (!h_desc || n <= 0 || narrow_fwd_ok(h_desc, n) || !glds_fwd_ok(h_desc)) return 0;
    const int64_t Mtot = n * h_desc->OH * h_desc->OW;
    if (img_fwd_index(make_geom(h_desc), n) >= 0) return 0;
    const GldsFwdPlan p = plan_fwd_t(Mtot, h_desc->Cout, h_desc->KH * h_desc->KW * h_desc->Cin);
    return p.ok && p.Z > 1 ? (int64_t)sizeof(float) * p.Z * Mtot * h_desc->Cout + 256 : 0;
}
// (XCD-aware block order of the launches with more than one column tile: sf_nn_glds.h; SF_XCD_RASTER=0 switches it off)
static int xcd_raster_on() {
    static const int on = getenv("SF_XCD_RASTER") ? atoi(getenv("SF_XCD_RASTER")) : 1;
    return on;
}
// SF_XCD_ROWS=1: the same re-mapping for launches with ONE column tile (conv2's forward): XCD c then owns a contiguous run
// of row tiles, so the image rows two neighbouring tiles share (a sample straddling a tile boundary, the 2-row halo of the
// 4x4 stride-2 window) are fetched into one L2 instead of two
static int xcd_rows_on() {
    static const int on = getenv("SF_XCD_ROWS") ? atoi(getenv("SF_XCD_ROWS")) : 0;
    return on;
}
// SF_GLDS_ZL (default 2; 1 = wave tiles of at most two 32x32 blocks only, 0 = off): the LDS-DMA forward with no vector-ALU
// instruction in its k-loop (k_fwd_glds_z, sf_nn_glds.h).  Measured (profiles/r05_k_zl_ab.log, r05_k_zl128_ab.log, same box,
// alternating): conv2 forward 213 / 218 -> 192 / 195 us at n = 4096, 1491 / 1507 -> 1413 / 1406 us at n = 32768; fc forward
// of a rollout step (64 x 64 tiles) 110 / 113 -> 105 / 104 us; fc forward at n = 32768 (128 x 128 tiles) 832 / 837 -> 802 / 811 us.
static int glds_zl_on() {
    static const int on = getenv("SF_GLDS_ZL") ? atoi(getenv("SF_GLDS_ZL")) : 2;
    return on;
}
// SF_TAP_PERM=1 (default 0): conv2's forward visits its 16 filter taps in groups of the four taps that read the same input
// elements (k_fwd_glds, sf_nn_glds.h).  Measured (profiles/r05_h_*): counter traffic of the dominant kernel 805.6 -> 544.1 MB
// per launch (1.56 -> 1.05 x algorithmic), launch time 1573 / 1591 vs 1591 / 1597 us at n = 32768 — the re-reads were being
// served by the Infinity Cache, so the time does not move.  It is a different fp32 summation order, though, and the
// normalised-input replay (train_cnn84_norm: inputs up to +-5, two SGD steps) lands on other ReLU flips with it and leaves the
// tolerance the replays are held to (profiles/r05_m_norm_bisect.log); with no time to gain, the natural order stays.
static int tap_perm_on() {
    static const int on = getenv("SF_TAP_PERM") ? atoi(getenv("SF_TAP_PERM")) : 0;
    return on;
}
// SF_GLDS_TAILSPLIT (default 1): single-column unsplit 128 x 64 launches go through k_fwd_glds_zt, which runs a last round of
// work-groups that is at most half full as 64-row tiles.  Returns the number of 128-row tiles in front of that round (all of
// them when there is nothing to split), 0 = not such a launch (k_fwd_glds_z / k_fwd_glds).
static int fwd_tail_split(const ConvG &g, int64_t Mtot, int Z, bool zl_ok) {
    static const int on = getenv("SF_GLDS_TAILSPLIT") ? atoi(getenv("SF_GLDS_TAILSPLIT")) : 1;
    if (!on || !zl_ok || glds_zl_on() < 1 || Z != 1 || g.Cout > 64 || xcd_rows_on()) return 0;
    static const int occ = occupancy_of(k_fwd_glds_zt<128, 64, 2, 2>, 256, (128 + 64) * 32 * 2 * sizeof(float));
    const int64_t tiles = cdiv64(Mtot, 128), resident = (int64_t)num_cus() * occ, tail = tiles % resident;
    if (tiles <= resident || tail == 0 || 2 * tail > resident) return (int)tiles;
    return (int)(tiles - tail);
}
#define GLDS_FWD(BM, BN, WM, WN, NS)                                                                           \
    do {                                                                                                       \
        dim3 gq(cdiv64(Mtot, BM), cdiv64(g.Cout, BN), p.Z);                                                    \
        const int rx = (int)gq.x, ry = (int)gq.y,                                                               \
                  rtot = (xcd_raster_on() && (gq.y > 1 || (xcd_rows_on() && gq.x >= 64))) ? (int)(gq.x * gq.y * gq.z) : 0; \
        if (rtot > 0) gq = dim3((unsigned)(8 * ((rtot + 7) / 8)), 1, 1);                                        \
        bool launched_z = false;                                                                               \
        {                                                                                                      \
            if (zl_ok && ((BM / WM / 32) * (BN / WN / 32) <= 2 || glds_zl_on() >= 2)) {                        \
                k_fwd_glds_z<BM, BN, WM, WN><<<gq, dim3(256), 0, st>>>(                                        \
                    g, in, in_sample_stride, wt, bias, out, Mtot, p.k_per_split, partial, nullptr, 0, rx, ry,  \
                    rtot, tap_perm_on());                                                                      \
                launched_z = true;                                                                             \
            }                                                                                                  \
        }                                                                                                      \
        if (!launched_z)                                                                                       \
        k_fwd_glds<BM, BN, WM, WN, NS><<<gq, dim3(256), 0, st>>>(                                              \
            g, in, in_sample_stride, wt, bias, out, Mtot, p.k_per_split, partial, nullptr, 0, rx, ry, rtot,    \
            tap_perm_on());                                                                                    \
    } while (0)
extern "C" int sf_conv_fwd_t(const float *in, int64_t in_sample_stride, const float *wt, const float *bias, float *out,
                             int64_t n, const sf_conv_desc *h_desc, void *workspace, int64_t workspace_bytes,
                             void *stream) {
    int rc = check_desc(h_desc, "sf_conv_fwd_t");
    if (rc) return rc;
    SF_REQUIRE(in && wt && out && n > 0, "sf_conv_fwd_t: bad args");
    SF_REQUIRE(((uintptr_t)in & 15) == 0 && ((uintptr_t)wt & 15) == 0 && in_sample_stride % 4 == 0,
               "sf_conv_fwd_t: operands must be 16-byte aligned");
    if (narrow_fwd_ok(h_desc, n)) {
        const dim3 grid((unsigned)cdiv64(n, 16)), block(64);
        if (h_desc->Cout <= 16)
            k_linear_narrow<1><<<grid, block, 0, STREAM(stream)>>>(in, in_sample_stride, wt, bias, out, (int)n, h_desc->Cout,
                                                                    h_desc->Cin, h_desc->relu);
        else
            k_linear_narrow<2><<<grid, block, 0, STREAM(stream)>>>(in, in_sample_stride, wt, bias, out, (int)n, h_desc->Cout,
                                                                    h_desc->Cin, h_desc->relu);
        return sf_launch_status("sf_conv_fwd_t");
    }
    SF_REQUIRE(glds_fwd_ok(h_desc), "sf_conv_fwd_t: needs f32 NHWC input with Cin %% 32 == 0 (use sf_conv_fwd)");
    const ConvG g = make_geom(h_desc);
    const int64_t Mtot = n * g.OH * g.OW;
    SF_REQUIRE(Mtot < (1LL << 31), "sf_conv_fwd_t: M=%lld exceeds 2^31 rows; split the batch", (long long)Mtot);
    if (n * in_sample_stride < ((int64_t)1 << 40) && launch_img_fwd(g, in, in_sample_stride, wt, bias, out, n, STREAM(stream)))
        return sf_launch_status("sf_conv_fwd_t");
    GldsFwdPlan p = plan_fwd_t(Mtot, g.Cout, g.K);
    if (!p.ok) {  // not a grid sf_conv_fwd_t_supported recommends: still correct, one unsplit launch
        p.Z = 1;
        p.k_per_split = (g.K + 31) / 32 * 32;
    }
    float *partial = nullptr;
    if (p.Z > 1) {
        SF_REQUIRE(workspace && ((uintptr_t)workspace & 15) == 0 &&
                       workspace_bytes >= (int64_t)sizeof(float) * p.Z * Mtot * g.Cout,
                   "sf_conv_fwd_t: this launch is split along K and needs sf_conv_fwd_t_workspace() bytes of workspace");
        partial = reinterpret_cast<float *>(workspace);
    }
    hipStream_t st = STREAM(stream);
    // zero-VALU k-loop form (k_fwd_glds_z): every per-lane operand offset must fit 32 bits
    const bool zl_ok = glds_zl_on() && (n - 1) * in_sample_stride + (int64_t)g.H * g.W * g.Cin < (1LL << 30) &&
                       (int64_t)g.Cout * g.K < (1LL << 30);
    // SF_GLDS_TALL=<min 256-row tiles> (experiment): 256 x 64 tiles (waves 4 x 1, 64 x 64 wave tiles) for 64-column layers with
    // many rows — half the per-tile fixed cost and a sixth less DMA per flop, at two work-groups per CU instead of three
    static const int tall = getenv("SF_GLDS_TALL") ? atoi(getenv("SF_GLDS_TALL")) : 0;
    static const int persist = getenv("SF_GLDS_PERSIST") ? atoi(getenv("SF_GLDS_PERSIST")) : 0;
    if (persist && !p.sq64 && !p.wide && p.Z == 1 && g.Cout <= 64 && zl_ok) {  // experiment: persistent row-tile walk
        static const int occp = occupancy_of(k_fwd_glds_zp<128, 64, 2, 2>);
        const int64_t tiles = cdiv64(Mtot, 128), slots = (int64_t)num_cus() * occp;
        k_fwd_glds_zp<128, 64, 2, 2><<<dim3((unsigned)(tiles < slots ? tiles : slots)), dim3(256), 0, st>>>(
            g, in, in_sample_stride, wt, bias, out, Mtot, p.k_per_split);
    } else
    if (tall && !p.sq64 && !p.wide && p.Z == 1 && g.Cout <= 64 && cdiv64(Mtot, 256) >= tall && zl_ok) {
        dim3 gq(cdiv64(Mtot, 256), 1, 1);
        k_fwd_glds_z<256, 64, 4, 1><<<gq, dim3(256), 0, st>>>(g, in, in_sample_stride, wt, bias, out, Mtot, p.k_per_split,
                                                                nullptr, nullptr, 0, 0, 0, 0, tap_perm_on());
    } else
    if (p.sq64) GLDS_FWD(64, 64, 2, 2, 2);
    else if (p.wide) GLDS_FWD(128, 128, 2, 2, 2);
    else if (const int main_tiles = fwd_tail_split(g, Mtot, p.Z, zl_ok); main_tiles > 0) {
        const int tail = (int)cdiv64(Mtot, 128) - main_tiles;
        k_fwd_glds_zt<128, 64, 2, 2><<<dim3((unsigned)(main_tiles + 2 * tail)), dim3(256), (128 + 64) * 32 * 2 * sizeof(float), st>>>(
            g, in, in_sample_stride, wt, bias, out, Mtot, p.k_per_split, main_tiles, tap_perm_on());
    } else GLDS_FWD(128, 64, 2, 2, 2);
    if (partial) {
        const int64_t MN = Mtot * g.Cout;
        k_splitk_finish<<<dim3(cdiv64(MN, 256) < 4096 ? cdiv64(MN, 256) : 4096), dim3(256), 0, st>>>(
            partial, bias, out, MN, g.Cout, p.Z, g.relu);
    }
    return sf_launch_status("sf_conv_fwd_t");
}
// two linear layers into one accumulator (k_fwd_glds2): out = a1 w1t^T + a2 w2t^T + bias1 + bias2
extern "C" int sf_linear_fwd_dual_supported(int64_t n, int N, int K1, int K2) {
    static const int on = getenv("SF_LINEAR_DUAL") ? atoi(getenv("SF_LINEAR_DUAL")) : 1;
    return on && n > 0 && N >= 64 && K1 > 0 && K2 > 0 && K1 % 32 == 0 && K2 % 32 == 0 && n < (1LL << 31) &&
           cdiv64(n, 128) * (int64_t)cdiv64(N, 64) >= 256;
}
extern "C" int sf_linear_fwd_dual(const float *a1, int64_t lda1, const float *w1t, const float *bias1, int K1,
                                  const float *a2, int64_t lda2, const float *w2t, const float *bias2, int K2, float *out,
                                  int64_t n, int N, int gru_H, void *stream) {
    SF_REQUIRE(a1 && w1t && a2 && w2t && out && n > 0 && N > 0, "sf_linear_fwd_dual: bad args");
    SF_REQUIRE(K1 > 0 && K2 > 0 && K1 % 32 == 0 && K2 % 32 == 0 && lda1 % 4 == 0 && lda2 % 4 == 0 && n < (1LL << 31),
               "sf_linear_fwd_dual: K1, K2 must be multiples of 32 and the row strides multiples of 4 (K1=%d K2=%d)", K1, K2);
    SF_REQUIRE((((uintptr_t)a1 | (uintptr_t)a2 | (uintptr_t)w1t | (uintptr_t)w2t) & 15) == 0,
               "sf_linear_fwd_dual: operands must be 16-byte aligned");
    SF_REQUIRE(gru_H == 0 || (gru_H > 0 && gru_H % 64 == 0 && N == 4 * gru_H),
               "sf_linear_fwd_dual: the GRU column layout needs N == 4 * gru_H and gru_H %% 64 == 0 (N=%d gru_H=%d)", N, gru_H);
    k_fwd_glds2<128, 64, 2, 2><<<dim3(cdiv64(n, 128), cdiv64(N, 64)), dim3(256), 0, STREAM(stream)>>>(
        a1, lda1, w1t, bias1, K1, a2, lda2, w2t, bias2, K2, out, n, N, gru_H);
    return sf_launch_status("sf_linear_fwd_dual");
}
extern "C" int sf_transpose(const float *w, float *wt, int K, int N, void *stream) {
    SF_REQUIRE(w && wt && K > 0 && N > 0, "sf_transpose: bad args");
    k_transpose<<<dim3(cdiv64(K, 32), cdiv64(N, 32)), dim3(256), 0, STREAM(stream)>>>(w, wt, K, N);
    return sf_launch_status("sf_transpose");
}

// split plan shared by the workspace query and the launcher
struct SplitPlan {
    int Z;
    int64_t m_per_split;
};
// bpc = resident blocks per CU of the kernel that will run (hipOccupancy...), 0 = unknown (workspace query: return the
// largest split count any bpc can lead to).  The reduction is split so that the grid fills the chip in WHOLE rounds:
// 200 tiles x Z = 8 on 1024 slots is 1.56 rounds, i.e. 2 rounds at 78 % — Z = 5 (0.98 rounds) is 25 % faster.
static SplitPlan plan_splits(int64_t Mtot, int K, int N, int BM, int BN, int bpc = 0) {
    const int64_t tiles = (int64_t)((K + BM - 1) / BM) * ((N + BN - 1) / BN);
    const int64_t chunks = (Mtot + 31) / 32;
    int64_t zcap = (2304 + tiles - 1) / tiles;  // never more than ~9 blocks per CU
    const int64_t zmax = (chunks + 7) / 8;      // at least 8 chunks (256 reduction rows) per split
    if (zcap > zmax) zcap = zmax;
    if (zcap < 1) zcap = 1;
    if (zcap > 1024) zcap = 1024;
    int64_t Z = zcap;
    if (bpc > 0) {
        const double slots = 256.0 * bpc;
        double best = -1.0;
        for (int64_t z = 1; z <= zcap; ++z) {
            const double u = (double)(tiles * z) / slots, rounds = u <= 1.0 ? 1.0 : (double)(int64_t)(u + 0.999999);
            double eff = u / rounds;
            if (tiles * z < 256) eff *= 0.5;          // fewer blocks than CUs: strictly worse than the formula says
            if (eff > best + 0.02) { best = eff; Z = z; }  // ties / near-ties: the smaller split (less partial traffic)
        }
    }
    SplitPlan p;
    p.m_per_split = ((chunks + Z - 1) / Z) * 32;
    p.Z = (int)((Mtot + p.m_per_split - 1) / p.m_per_split);
    return p;
}
template <int BN, int WM, int WN, int MODE>
static int occ_wgrad() {
    static const int v = occupancy_of(k_conv_wgrad<BN, WM, WN, MODE>);
    return v;
}
template <int BK, int BN, int WM, int WN>
static int occ_wgrad_glds() {
    static const int v = occupancy_of(k_wgrad_glds<BK, BN, WM, WN>);
    return v;
}
static inline int wgrad_bn(int N) { return N <= 32 ? 32 : 64; }
struct WgradGlds {
    int cfg;  // 0: 256x64 (waves 4x1), 1: 128x128 (2x2), 2: 128x64 (2x2), 3: 64x128 (2x2; K = 64: the recurrent input projection)
    int BK, BN, Z;
    int64_t m_per_split;
};
static WgradGlds plan_wgrad_glds(int64_t Mtot, int K, int N, bool query_occupancy = false) {
    static const int force = getenv("SF_WGRAD_GLDS") ? atoi(getenv("SF_WGRAD_GLDS")) : 1;
    WgradGlds q;
    // 256-row weight tiles only when they do not add padded rows over 128-row tiles (K = 576: 768 vs 640 rows)
    const bool k256 = K >= 256 && (K + 255) / 256 * 256 <= (K + 127) / 128 * 128;
    q.cfg = N >= 128 ? 1 : (k256 ? 0 : 2);
    if (force >= 2) q.cfg = force - 2;
    if (K == 64 && N >= 128) q.cfg = 3;  // a 128-row weight tile would be half padding
    q.BK = q.cfg == 0 ? 256 : q.cfg == 3 ? 64 : 128;
    q.BN = (q.cfg == 1 || q.cfg == 3) ? 128 : 64;
    int bpc = 0;
    if (query_occupancy)
        bpc = q.cfg == 0 ? occ_wgrad_glds<256, 64, 4, 1>() : q.cfg == 1 ? occ_wgrad_glds<128, 128, 2, 2>()
              : q.cfg == 3 ? occ_wgrad_glds<64, 128, 2, 2>() : occ_wgrad_glds<128, 64, 2, 2>();
    const SplitPlan p = plan_splits(Mtot, K, N, q.BK, q.BN, bpc);
    q.Z = p.Z;
    q.m_per_split = p.m_per_split;
    return q;
}

// Which reductions go to the LDS-DMA weight-gradient kernel: long ones, and mid-sized ones with a large K x N (the fc
// layer at n = 32768: 1044 -> 911 us; the LSTM's 512 x 2048 recurrent matrix at 16384 chunk rows; the 8-column heads
// stay on the register-staged kernel).  SF_WGRAD_GLDS_MIN
// overrides the row threshold (A/B switch).
static bool wgrad_glds_wanted(int64_t Mtot, int K, int N) {
    static const int64_t v = getenv("SF_WGRAD_GLDS_MIN") ? atoll(getenv("SF_WGRAD_GLDS_MIN")) : -1;
    if (v >= 0) return Mtot >= v;
    static const int k64 = getenv("SF_WGRAD_GLDS_K64") ? atoi(getenv("SF_WGRAD_GLDS_K64")) : 1;
    if (k64 && Mtot >= 16384 && K == 64 && N >= 512) return true;  // W_ih of a recurrent core behind a 64-wide encoder
    return Mtot >= 65536 || (Mtot >= 16384 && N >= 64 && (K >= 1024 || (int64_t)K * N >= 512 * 1024));
}

// LDS-image weight gradient (sf_nn_wimg.h): Nature-CNN conv3 (variant 1: 4 waves, two work-groups per CU) and conv2
// (variant 2: 8 waves, one work-group per CU) geometries, launches that give every persistent work-group a few samples.
// SF_WGRAD_IMG=0: back on k_wgrad_glds; =1: conv3 only; default 3: both (A/B switch).
static int wgrad_img_variant(const sf_conv_desc *d, int64_t n) {
    static const int on = getenv("SF_WGRAD_IMG") ? atoi(getenv("SF_WGRAD_IMG")) : 3;
    if (!on || d->in_u8 || d->traj_T != 0 || d->Cout != 64 || n < 512) return 0;
    if ((on & 1) && d->Cin == 64 && d->H == 9 && d->W == 9 && d->KH == 3 && d->KW == 3 && d->stride == 1) return 1;
    if ((on & 2) && d->Cin == 32 && d->H == 20 && d->W == 20 && d->KH == 4 && d->KW == 4 && d->stride == 2) return 2;
    return 0;
}
static int wgrad_img_blocks(const sf_conv_desc *d, int64_t n) {
    const int64_t nb = (wgrad_img_variant(d, n) == 1 ? 2 : 1) * (int64_t)num_cus();
    return (int)(n < nb ? n : nb);
}

extern "C" int64_t sf_conv_wgrad_workspace(int64_t n, const sf_conv_desc *h_desc) {
    if (!h_desc || n <= 0) return 0;
    const int K = h_desc->KH * h_desc->KW * h_desc->Cin, N = h_desc->Cout;
    const int64_t Mtot = n * h_desc->OH * h_desc->OW;
    const SplitPlan p = plan_splits(Mtot, K, N, 128, wgrad_bn(N));
    int Z = p.Z;
    if (wgrad_img_variant(h_desc, n) && wgrad_img_blocks(h_desc, n) > Z) Z = wgrad_img_blocks(h_desc, n);  // one partial per work-group
    if (small_linear_wgrad_ok(h_desc)) {  // one partial per 64-row tile, at most 1024
        const int64_t zs = cdiv64(Mtot, 64) < 1024 ? cdiv64(Mtot, 64) : 1024;
        if (zs > Z) Z = (int)zs;
    }
    if (!h_desc->in_u8) {  // the LDS-DMA kernel may pick other tiles (hence another split count)
        const WgradGlds q = plan_wgrad_glds(Mtot, K, N);
        if (q.Z > Z) Z = q.Z;
    }
    return (int64_t)sizeof(float) * Z * ((int64_t)K * N + N) + 256;
}

#define WGRAD_LAUNCH(BN, WM, WN, MODE)                                                                        \
    k_conv_wgrad<BN, WM, WN, MODE><<<grid, dim3(256), 0, st>>>(g, in, in_sample_stride, index, offset, dout,  \
                                                              partial_w, db ? partial_b : nullptr, Mtot,     \
                                                              p.m_per_split)
#define WGRAD_BY_MODE(BN, WM, WN)                                    \
    do {                                                             \
        if (mode == MODE_F32) WGRAD_LAUNCH(BN, WM, WN, MODE_F32);    \
        else if (mode == MODE_U8) WGRAD_LAUNCH(BN, WM, WN, MODE_U8); \
        else WGRAD_LAUNCH(BN, WM, WN, MODE_GENERIC);                 \
    } while (0)

static int conv_wgrad_impl(const void *in, int64_t in_sample_stride, const int32_t *index, int64_t offset,
                           const float *dout, const uint32_t *dmask, float *dw, float *db, int64_t n,
                           const sf_conv_desc *h_desc, void *workspace, void *stream) {
    int rc = check_desc(h_desc, "sf_conv_wgrad");
    if (rc) return rc;
    SF_REQUIRE(in && dout && dw && workspace && n > 0, "sf_conv_wgrad: bad args");
    SF_REQUIRE(((uintptr_t)workspace & 15) == 0, "sf_conv_wgrad: workspace must be 16-byte aligned");
    const ConvG g = make_geom(h_desc);
    const int64_t Mtot = n * g.OH * g.OW;
    SF_REQUIRE(Mtot < (1LL << 31), "sf_conv_wgrad: M=%lld exceeds 2^31 rows; split the batch", (long long)Mtot);
    const int K = g.K, N = g.Cout;
    const int BN = wgrad_bn(N);
    int mode = pick_mode(g);
    if (mode != MODE_GENERIC) {
        const bool al = h_desc->in_u8 ? (((uintptr_t)in & 3) == 0 && in_sample_stride % 4 == 0)
                                      : (((uintptr_t)in & 15) == 0 && in_sample_stride % 4 == 0);
        if (!al || ((uintptr_t)dout & 15) != 0) mode = MODE_GENERIC;
    }
    int bpc;
    if (BN == 32) bpc = mode == MODE_F32 ? occ_wgrad<32, 4, 1, MODE_F32>() : mode == MODE_U8 ? occ_wgrad<32, 4, 1, MODE_U8>()
                                                                                              : occ_wgrad<32, 4, 1, MODE_GENERIC>();
    else bpc = mode == MODE_F32 ? occ_wgrad<64, 2, 2, MODE_F32>() : mode == MODE_U8 ? occ_wgrad<64, 2, 2, MODE_U8>()
                                                                                     : occ_wgrad<64, 2, 2, MODE_GENERIC>();
    const SplitPlan p = plan_splits(Mtot, K, N, 128, BN, bpc);
    const int Zws = plan_splits(Mtot, K, N, 128, BN).Z;  // what sf_conv_wgrad_workspace promised room for
    float *partial_w = reinterpret_cast<float *>(workspace);
    float *partial_b = partial_w + (int64_t)p.Z * K * N;
    hipStream_t st = STREAM(stream);
    static const int glds_on = getenv("SF_WGRAD_GLDS") ? atoi(getenv("SF_WGRAD_GLDS")) : 1;
    int Zused = p.Z;
    static const int img_on = getenv("SF_CONV1_IMG") ? atoi(getenv("SF_CONV1_IMG")) : 1;
    if (conv1_bf16_ok(g, mode, n) && N == 32 && ((uintptr_t)in & 3) == 0 && in_sample_stride % 4 == 0 &&
        ((uintptr_t)dout & 15) == 0) {
        // exact products on the bf16 matrix pipe (sf_nn_u8.h): persistent blocks over sample pairs, one partial per block
        const int npairs = (int)((n + 1) / 2);
        int nb = npairs < 2 * num_cus() ? npairs : 2 * num_cus();
        if (nb > Zws) nb = Zws;  // the workspace was sized for Zws partials
        partial_b = partial_w + (int64_t)nb * K * N;
        Zused = nb;
        const unsigned lds_bytes = (unsigned)((2 * 4 * 20 * 4 * 36 + 3 * 32 * 168) * sizeof(uint16_t));
        if (g.sub_mean != 0.f)
            k_conv1_wgrad_bf16<true><<<dim3(nb), dim3(256), lds_bytes, st>>>(
                g, reinterpret_cast<const uint8_t *>(in), in_sample_stride, index, offset, dout, dmask, partial_w,
                db ? partial_b : nullptr, (int)n, npairs);
        else
            k_conv1_wgrad_bf16<false><<<dim3(nb), dim3(256), lds_bytes, st>>>(
                g, reinterpret_cast<const uint8_t *>(in), in_sample_stride, index, offset, dout, dmask, partial_w,
                db ? partial_b : nullptr, (int)n, npairs);
    } else
    if (dmask) {
        SF_REQUIRE(false, "sf_conv_wgrad_relu_mask: this launch does not resolve to the mask-consuming kernel");
    } else
    if (img_on && conv1_img_ok(g, mode, n) && N == 32 && ((uintptr_t)in & 3) == 0 && in_sample_stride % 4 == 0 &&
        ((uintptr_t)dout & 15) == 0) {
        // Nature-CNN conv1 on raw frames: strip-image kernel, persistent blocks, one partial per block
        const int npairs = (int)((n + 1) / 2);
        int nb = npairs < 512 ? npairs : 512;
        if (nb > Zws) nb = Zws;  // the workspace was sized for Zws partials
        partial_b = partial_w + (int64_t)nb * K * N;
        Zused = nb;
        const unsigned lds_bytes = (unsigned)((160 * 32 + 2 * 4 * 20 * 84) * sizeof(float));
        if (g.sub_mean != 0.f)
            k_conv1_wgrad_img<2, 4, true><<<dim3(nb), dim3(256), lds_bytes, st>>>(
                g, reinterpret_cast<const uint8_t *>(in), in_sample_stride, index, offset, dout, partial_w,
                db ? partial_b : nullptr, (int)n, npairs);
        else
            k_conv1_wgrad_img<2, 4, false><<<dim3(nb), dim3(256), lds_bytes, st>>>(
                g, reinterpret_cast<const uint8_t *>(in), in_sample_stride, index, offset, dout, partial_w,
                db ? partial_b : nullptr, (int)n, npairs);
    } else
    if (small_linear_wgrad_ok(h_desc) && !index) {
        // 27 -> 64 -> 64 encoder layers: 64-row tiles through LDS, fmaf (sf_nn_narrow.h); one partial per work-group
        int nb = (int)(cdiv64(Mtot, 64) < 256 ? cdiv64(Mtot, 64) : 256);  // (<= what sf_conv_wgrad_workspace sized for)
        const int64_t mps = cdiv64(cdiv64(Mtot, nb), 64) * 64;
        nb = (int)cdiv64(Mtot, mps);
        partial_b = partial_w + (int64_t)nb * K * N;
        Zused = nb;
        k_linear_wgrad_small<<<dim3(nb), dim3(256), 0, st>>>(reinterpret_cast<const float *>(in), in_sample_stride, dout,
                                                             partial_w, db ? partial_b : nullptr, Mtot, mps, K, N);
    } else
    if (mode == MODE_F32 && !index && wgrad_img_variant(h_desc, n)) {
        // conv2 / conv3: persistent LDS-image kernel, every operand byte fetched once, one partial per work-group
        const int nb = wgrad_img_blocks(h_desc, n);
        partial_b = partial_w + (int64_t)nb * K * N;
        Zused = nb;
        const float *inf = reinterpret_cast<const float *>(in);
        if (wgrad_img_variant(h_desc, n) == 1)
            k_wgrad_img<64, 9, 9, 3, 1, 1><<<dim3(nb), dim3(256), 0, st>>>(inf, in_sample_stride, dout, partial_w,
                                                                            db ? partial_b : nullptr, (int)n);
        else
            k_wgrad_img<32, 20, 20, 4, 2, 2><<<dim3(nb), dim3(512), 0, st>>>(inf, in_sample_stride, dout, partial_w,
                                                                              db ? partial_b : nullptr, (int)n);
    } else
    if (glds_on && mode == MODE_F32 && !index && g.traj_T == 0 && wgrad_glds_wanted(Mtot, K, N) &&
        n * max(in_sample_stride, (int64_t)g.H * g.W * g.Cin) < ((int64_t)1 << 30)) {  // 32-bit byte offsets in the kernel
        // gfx950 LDS-DMA kernel (dense f32 NHWC input): different tiles, so its own split plan and partial layout
        const WgradGlds q = plan_wgrad_glds(Mtot, K, N, true);
        partial_b = partial_w + (int64_t)q.Z * K * N;
        Zused = q.Z;
        dim3 gq(cdiv64(K, q.BK), cdiv64(N, q.BN), (unsigned)q.Z);
        // XCD-aware block order (sf_nn_glds.h): a 1-D launch whose ids are re-mapped so that every XCD owns a contiguous
        // run of (row tile, column tile, slice) — only worth it when tiles share strips (more than one tile per slice)
        const int rx = (int)gq.x, ry = (int)gq.y, rtot = (xcd_raster_on() && gq.x * gq.y > 1) ? (int)(gq.x * gq.y * gq.z) : 0;
        if (rtot > 0) gq = dim3((unsigned)(8 * ((rtot + 7) / 8)), 1, 1);
        const float *inf = reinterpret_cast<const float *>(in);
        // linear layers: the zero-VALU reduction loop (k_wgrad_glds_z); SF_WGRAD_ZL=0 switches it off
        static const int wzl = getenv("SF_WGRAD_ZL") ? atoi(getenv("SF_WGRAD_ZL")) : 1;
        const bool wgrad_zl = wzl && g.KH == 1 && g.KW == 1 && g.H == 1 && g.W == 1 && g.OH == 1 && g.OW == 1 &&
                              n * in_sample_stride < (1LL << 30) && Mtot * N < (1LL << 30);
#define WGRAD_GLDS(BK_, BN_, WM_, WN_)                                                                      \
    do {                                                                                                    \
        if (wgrad_zl)                                                                                       \
            k_wgrad_glds_z<BK_, BN_, WM_, WN_><<<gq, dim3(256), 0, st>>>(g, inf, in_sample_stride, dout, partial_w, \
                                                                        db ? partial_b : nullptr, Mtot, q.m_per_split, rx, ry, rtot); \
        else                                                                                                \
    k_wgrad_glds<BK_, BN_, WM_, WN_><<<gq, dim3(256), 0, st>>>(g, inf, in_sample_stride, dout, partial_w,   \
                                                              db ? partial_b : nullptr, Mtot, q.m_per_split, rx, ry, rtot); \
    } while (0)
        if (q.cfg == 3) WGRAD_GLDS(64, 128, 2, 2);
        else if (q.cfg == 0) WGRAD_GLDS(256, 64, 4, 1);
        else if (q.cfg == 1) WGRAD_GLDS(128, 128, 2, 2);
        else WGRAD_GLDS(128, 64, 2, 2);
    } else {
        dim3 grid(cdiv64(K, 128), cdiv64(N, BN), (unsigned)p.Z);
        if (BN == 32) WGRAD_BY_MODE(32, 4, 1);
        else WGRAD_BY_MODE(64, 2, 2);
    }
    const int64_t KN = (int64_t)K * N;
    launch_reduce_partials(partial_w, dw, KN, Zused, st);
    if (db) launch_reduce_partials(partial_b, db, N, Zused, st);
    return sf_launch_status("sf_conv_wgrad");
}

extern "C" int sf_conv_wgrad(const void *in, int64_t in_sample_stride, const int32_t *index, int64_t offset,
                             const float *dout, float *dw, float *db, int64_t n, const sf_conv_desc *h_desc,
                             void *workspace, void *stream) {
    return conv_wgrad_impl(in, in_sample_stride, index, offset, dout, nullptr, dw, db, n, h_desc, workspace, stream);
}
extern "C" int sf_conv_wgrad_relu_mask(const void *in, int64_t in_sample_stride, const int32_t *index, int64_t offset,
                                       const float *dout, const uint32_t *relu_mask, float *dw, float *db, int64_t n,
                                       const sf_conv_desc *h_desc, void *workspace, void *stream) {
    SF_REQUIRE(relu_mask && ((uintptr_t)relu_mask & 7) == 0, "sf_conv_wgrad_relu_mask: relu_mask must be an 8-byte aligned device buffer");
    SF_REQUIRE(h_desc && n > 0 && relu_mask_ok(h_desc, n) && ((uintptr_t)in & 3) == 0 && in_sample_stride % 4 == 0 &&
                   ((uintptr_t)dout & 15) == 0,
               "sf_conv_wgrad_relu_mask: unsupported layer / launch (see sf_conv_relu_mask_supported)");
    return conv_wgrad_impl(in, in_sample_stride, index, offset, dout, relu_mask, dw, db, n, h_desc, workspace, stream);
}

// ---- conv1 on raw u8 frames WITH the observation normaliser's running statistics applied in the loader (cfg.normalize_input
// on image observations: utils/normalize.py:51-70, running_mean_std.py:79-110, cfg/cfg.py:337-341 default True): no
// normalised f32 copy of the frames exists anywhere.  mu / rstd: the normaliser's f32 tables [Cin*H*W] in the frame's NCHW
// order (sf_obsnorm_update writes them).  Launches sf_conv_norm_supported() accepts: the Nature-CNN conv1 geometry the
// strip-image kernels are built for, n >= 256, 4-byte aligned frames; everything else goes through sf_obsnorm_apply.
static bool conv_norm_ok(const sf_conv_desc *d, int64_t n) {
    static const int on = getenv("SF_CONV1_NORM") ? atoi(getenv("SF_CONV1_NORM")) : 1;
    if (!on || !d || n <= 0 || !d->in_u8 || d->Cout != 32) return false;
    const ConvG g = make_geom(d);
    // the strip kernels' compile-time geometry; ANY n (their n >= 256 dispatch threshold is a speed heuristic of the plain
    // entry points, the kernels themselves are correct for every n >= 1 and there is no other kernel to fall back to)
    return pick_mode(g) == MODE_U8 && g.Cin == 4 && g.H == 84 && g.W == 84 && g.KH == 8 && g.KW == 8 && g.S == 4;
}
extern "C" int sf_conv_norm_supported(int64_t n, const sf_conv_desc *h_desc) {
    return h_desc && check_desc(h_desc, "sf_conv_norm_supported") == 0 && conv_norm_ok(h_desc, n) ? 1 : 0;
}
extern "C" int sf_conv_fwd_norm(const void *in, int64_t in_sample_stride, const int32_t *index, int64_t offset,
                                const float *mu, const float *rstd, const float *w, const float *bias, float *out,
                                int64_t n, const sf_conv_desc *h_desc, void *stream) {
    int rc = check_desc(h_desc, "sf_conv_fwd_norm");
    if (rc) return rc;
    SF_REQUIRE(in && mu && rstd && w && out && n > 0, "sf_conv_fwd_norm: bad args");
    SF_REQUIRE(conv_norm_ok(h_desc, n) && ((uintptr_t)in & 3) == 0 && in_sample_stride % 4 == 0 &&
                   ((uintptr_t)mu & 15) == 0 && ((uintptr_t)rstd & 15) == 0,
               "sf_conv_fwd_norm: unsupported layer / launch (see sf_conv_norm_supported; frames 4-byte, tables 16-byte aligned)");
    ConvG g = make_geom(h_desc);
    g.nmu = mu; g.nrstd = rstd;
    SF_REQUIRE(n * g.OH * g.OW < (1LL << 31), "sf_conv_fwd_norm: M exceeds 2^31 rows; split the batch");
    const unsigned lds_bytes = (unsigned)(2 * 4 * 20 * 84 * sizeof(float));
    static const int bpc = occupancy_of(k_conv_u8_img_norm<2, 4, 5, 16>, 256, lds_bytes);
    const int64_t npairs = cdiv64(n, 2), resident = (int64_t)num_cus() * (bpc > 0 ? bpc : 1);
    k_conv_u8_img_norm<2, 4, 5, 16><<<dim3((unsigned)(npairs < resident ? npairs : resident)), dim3(256), lds_bytes, STREAM(stream)>>>(
        g, reinterpret_cast<const uint8_t *>(in), in_sample_stride, index, offset, w, bias, out, (int)n);
    return sf_launch_status("sf_conv_fwd_norm");
}
extern "C" int sf_conv_wgrad_norm(const void *in, int64_t in_sample_stride, const int32_t *index, int64_t offset,
                                  const float *mu, const float *rstd, const float *dout, float *dw, float *db, int64_t n,
                                  const sf_conv_desc *h_desc, void *workspace, void *stream) {
    int rc = check_desc(h_desc, "sf_conv_wgrad_norm");
    if (rc) return rc;
    SF_REQUIRE(in && mu && rstd && dout && dw && workspace && n > 0, "sf_conv_wgrad_norm: bad args");
    SF_REQUIRE(conv_norm_ok(h_desc, n) && ((uintptr_t)in & 3) == 0 && in_sample_stride % 4 == 0 &&
                   ((uintptr_t)mu & 15) == 0 && ((uintptr_t)rstd & 15) == 0 && ((uintptr_t)dout & 15) == 0 &&
                   ((uintptr_t)workspace & 15) == 0,
               "sf_conv_wgrad_norm: unsupported layer / launch (see sf_conv_norm_supported)");
    ConvG g = make_geom(h_desc);
    g.nmu = mu; g.nrstd = rstd;
    const int K = g.K, N = g.Cout;
    const int Zws = plan_splits(n * g.OH * g.OW, K, N, 128, wgrad_bn(N)).Z;  // what sf_conv_wgrad_workspace promised room for
    const int npairs = (int)((n + 1) / 2);
    int nb = npairs < 512 ? npairs : 512;
    if (nb > Zws) nb = Zws;
    float *partial_w = reinterpret_cast<float *>(workspace), *partial_b = partial_w + (int64_t)nb * K * N;
    const unsigned lds_bytes = (unsigned)((160 * 32 + 2 * 4 * 20 * 84) * sizeof(float));
    hipStream_t st = STREAM(stream);
    k_conv1_wgrad_img_norm<2, 4><<<dim3(nb), dim3(256), lds_bytes, st>>>(
        g, reinterpret_cast<const uint8_t *>(in), in_sample_stride, index, offset, dout, partial_w, db ? partial_b : nullptr,
        (int)n, npairs);
    launch_reduce_partials(partial_w, dw, (int64_t)K * N, nb, st);
    if (db) launch_reduce_partials(partial_b, db, N, nb, st);
    return sf_launch_status("sf_conv_wgrad_norm");
}

#define DGRAD_LAUNCH(BM, BN, WM, WN)                                                                          \
    do {                                                                                                      \
        dim3 grid(cdiv64(Mc, BM), cdiv64(g.Cin, BN), classes);                                                \
        if (vec) k_conv_dgrad<BM, BN, WM, WN, true><<<grid, dim3(256), 0, st>>>(g, dout, w, in_act, din, n);  \
        else k_conv_dgrad<BM, BN, WM, WN, false><<<grid, dim3(256), 0, st>>>(g, dout, w, in_act, din, n);     \
    } while (0)

static sf_conv_desc linear_desc(int K, int N, int relu);
static bool linear_dgrad_glds_ok(const ConvG &g, int64_t n) {
    return g.H == 1 && g.W == 1 && g.KH == 1 && g.KW == 1 && g.Cout % 32 == 0 && g.Cin >= 128 &&
           cdiv64(n, 128) * (int64_t)cdiv64(g.Cin, 128) >= 512;
}
// ... and narrow ones (Cin == 64 behind a deep reduction: the data gradient of a recurrent core's input projection,
// 16384 x 2048 -> 64): 128-row tiles are 128 work-groups, half the chip; 64 x 64 tiles fill it
static bool linear_dgrad_glds64_ok(const ConvG &g, int64_t n) {
    static const int on = getenv("SF_DGRAD_LINEAR64") ? atoi(getenv("SF_DGRAD_LINEAR64")) : 1;
    return on && g.H == 1 && g.W == 1 && g.KH == 1 && g.KW == 1 && g.Cout % 32 == 0 && g.Cout >= 512 && g.Cin == 64 &&
           n >= 8192;
}
static int dgrad_lpt() {  // longest rows first (k_dgrad_pix block order, sf_nn_glds.h); SF_DGRAD_LPT=0: row-major block ids
    static const int on = getenv("SF_DGRAD_LPT") ? atoi(getenv("SF_DGRAD_LPT")) : 1;
    return on;
}
// k_dgrad_pix_z (SF_DGRAD_ZL bit 1): per-lane dY / W offsets of the SADDR-form DMA must fit 32 bits, channels in whole 64s
static bool dgrad_pix_zl(const ConvG &g, int64_t n) {
    static const int dzl = getenv("SF_DGRAD_ZL") ? atoi(getenv("SF_DGRAD_ZL")) : 3;
    return (dzl & 2) && g.Cout % 64 == 0 && n * (int64_t)g.OH * g.OW * g.Cout < (1LL << 30) && (int64_t)g.K * g.Cout < (1LL << 30);
}
// k_dgrad_quadrow_z addresses dY / W lanes as 32-bit element offsets and the input-gradient / activation elements as
// 32-bit BYTE offsets from a uniform base: both tensors must stay below 2^30 elements
static bool dgrad_quadrow_zl(const ConvG &g, int64_t n) {
    static const int dzl = getenv("SF_DGRAD_ZL") ? atoi(getenv("SF_DGRAD_ZL")) : 3;
    return (dzl & 1) && g.Cout % 64 == 0 && n * (int64_t)g.OH * g.OW * g.Cout < (1LL << 30) &&
           (int64_t)g.K * g.Cout < (1LL << 30) && n * (int64_t)g.H * g.W * g.Cin < (1LL << 30);
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
    SF_REQUIRE(n * g.H * g.W < (1LL << 31) && n * g.OH * g.OW * (int64_t)g.Cout < (1LL << 31) &&
                   (int64_t)g.K * g.Cout < (1LL << 31),
               "sf_conv_dgrad: operand too large for 32-bit element offsets; split the batch");
    hipStream_t st = STREAM(stream);
    const unsigned classes = (unsigned)(g.S * g.S);
    const bool vec = g.vecB && ((uintptr_t)dout & 15) == 0 && ((uintptr_t)w & 15) == 0;
    // Linear layer (1x1 on a 1x1 image): din[n, Cin] = dY[n, Cout] * W^T is the forward GEMM of the LDS-DMA kernel with
    // the canonical [Cin, Cout] weight array AS its Cout-major operand (no transpose needed) and a mask epilogue:
    // 128x128 tiles, 64x64 per wave (fc layer at n = 32768: 105 -> 120 TFLOP/s against the pixel-major kernel).
    static const int lin_on = getenv("SF_DGRAD_LINEAR") ? atoi(getenv("SF_DGRAD_LINEAR")) : 1;
    if (lin_on && linear_dgrad_glds_ok(g, n) && ((uintptr_t)dout & 15) == 0 && ((uintptr_t)w & 15) == 0) {
        sf_conv_desc d2 = linear_desc(g.Cout, g.Cin, g.relu);  // reduction = Cout, columns = Cin, relu = kind of in_act
        const ConvG g2 = make_geom(&d2);
        dim3 gq(cdiv64(n, 128), cdiv64(g.Cin, 128), 1);
        // (measured slower for this launch — 950 vs 930 us at n = 32768: both orders re-read one operand from the Infinity
        // Cache, and the row-strip order sweeps the 6.4 MB weight matrix per strip — so only SF_XCD_RASTER=2 enables it here)
        const int rx = (int)gq.x, ry = (int)gq.y, rtot = (xcd_raster_on() >= 2 && gq.y > 1) ? (int)(gq.x * gq.y) : 0;
        if (rtot > 0) gq = dim3((unsigned)(8 * ((rtot + 7) / 8)), 1, 1);
        const bool zl = glds_zl_on() >= 2 && n * (int64_t)g.Cout < (1LL << 30) && (int64_t)g.Cin * g.Cout < (1LL << 30);
        if (zl)
            k_fwd_glds_z<128, 128, 2, 2><<<gq, dim3(256), 0, st>>>(
                g2, dout, g.Cout, w, nullptr, din, n, (g.Cout + 31) / 32 * 32, nullptr, in_act, 1, rx, ry, rtot);
        else
        k_fwd_glds<128, 128, 2, 2, 2><<<gq, dim3(256), 0, st>>>(
            g2, dout, g.Cout, w, nullptr, din, n, (g.Cout + 31) / 32 * 32, nullptr, in_act, 1, rx, ry, rtot);
        return sf_launch_status("sf_conv_dgrad");
    }
    if (lin_on && linear_dgrad_glds64_ok(g, n) && ((uintptr_t)dout & 15) == 0 && ((uintptr_t)w & 15) == 0) {
        sf_conv_desc d2 = linear_desc(g.Cout, g.Cin, g.relu);
        const ConvG g2 = make_geom(&d2);
        const bool zl = glds_zl_on() && n * (int64_t)g.Cout < (1LL << 30) && (int64_t)g.Cin * g.Cout < (1LL << 30);
        if (zl)
            k_fwd_glds_z<64, 64, 2, 2><<<dim3(cdiv64(n, 64), 1, 1), dim3(256), 0, st>>>(
                g2, dout, g.Cout, w, nullptr, din, n, (g.Cout + 31) / 32 * 32, nullptr, in_act, 1);
        else
        k_fwd_glds<64, 64, 2, 2, 2><<<dim3(cdiv64(n, 64), 1, 1), dim3(256), 0, st>>>(
            g2, dout, g.Cout, w, nullptr, din, n, (g.Cout + 31) / 32 * 32, nullptr, in_act, 1);
        return sf_launch_status("sf_conv_dgrad");
    }
    // pixel-major LDS-DMA kernel: needs enough samples to fill BM-sample row tiles and Cout % 32 == 0
    static const int pix_cfg = getenv("SF_DGRAD_PIX") ? atoi(getenv("SF_DGRAD_PIX")) : 1;
    if (pix_cfg && vec && g.Cout % 32 == 0 && n >= 1024) {
#define DGRAD_PIX(BM, BN, WM, WN)                                                                          \
    do {                                                                                                   \
        const int ntiles = (int)((n + BM - 1) / BM), tiles8 = (ntiles + 7) / 8, ctiles = (g.Cin + BN - 1) / BN; \
        if (dgrad_zl)                                                                                      \
            k_dgrad_pix_z<BM, BN, WM, WN><<<dim3((unsigned)(tiles8 * 8 * g.H * ctiles)), dim3(256), 0, st>>>( \
                g, dout, w, in_act, din, (int)n, ntiles, tiles8, dgrad_lpt());                              \
        else                                                                                               \
        k_dgrad_pix<BM, BN, WM, WN><<<dim3((unsigned)(tiles8 * 8 * g.H * ctiles)), dim3(256), 0, st>>>(    \
            g, dout, w, in_act, din, (int)n, ntiles, tiles8, dgrad_lpt());                                 \
    } while (0)
        // zero-VALU reduction loop (k_dgrad_pix_z / k_dgrad_quadrow_z): per-lane operand offsets must fit 32 bits
        // SF_DGRAD_ZL (default 3): bit 0 = k_dgrad_quadrow_z (conv2: 1918 / 1921 -> 1877 / 1892 us at n = 32768), bit 1 =
        // k_dgrad_pix_z with SADDR-form DMA only (conv3: 1308 / 1316 -> 1277 / 1297 us; the full form — pointer fragment reads,
        // two chunks per trip — costs hipcc 256 + 168 registers against 173 + 32 and the second wave per SIMD with them:
        // 1214 -> 1316 us, compile-time switch SF_DGRAD_PIX_ZL_LITE=0) — profiles/r05_k_dgrad_zl_ab.log, r05_n_dgrad_pix_lite_ab.log
        const bool dgrad_zl = dgrad_pix_zl(g, n);
        if (g.S > 1 && g.KH % g.S == 0 && g.KW % g.S == 0 && g.W % g.S == 0 && pix_cfg != 3) {
            // strided conv, row-walking tiles of (sample, group-column) rows: contiguous activation / gradient rows
            const int Wg = g.W / g.S;
            const int64_t Mrows = n * Wg;
            if (dgrad_quadrow_zl(g, n))
                k_dgrad_quadrow_z<128, 128, 2, 2><<<dim3(cdiv64(Mrows, 128), cdiv64(g.S * g.S * g.Cin, 128)), dim3(256), 0,
                                                    st>>>(g, dout, w, in_act, din, Mrows, make_fastdiv((uint32_t)Wg));
            else
            k_dgrad_quadrow<128, 128, 2, 2><<<dim3(cdiv64(Mrows, 128), cdiv64(g.S * g.S * g.Cin, 128)), dim3(256), 0,
                                              st>>>(g, dout, w, in_act, din, Mrows, make_fastdiv((uint32_t)Wg));
            return sf_launch_status("sf_conv_dgrad");
        }
        if (g.Cin <= 32) { if (pix_cfg == 2) DGRAD_PIX(128, 32, 4, 1); else DGRAD_PIX(256, 32, 4, 1); }
        else DGRAD_PIX(128, 64, 2, 2);
        return sf_launch_status("sf_conv_dgrad");
    }
    if (g.Cin <= 32) DGRAD_LAUNCH(128, 32, 4, 1);
    else if (Mc * ((g.Cin + 63) / 64) < 128LL * 1024) DGRAD_LAUNCH(64, 64, 2, 2);
    else DGRAD_LAUNCH(128, 64, 2, 2);
    return sf_launch_status("sf_conv_dgrad");
}

// Name of the kernel instantiation a launch with these arguments resolves to (aligned operands assumed), spelled the
// way rocprofv3 prints it, so that bench.py can group its HIP-event timings exactly like the rocprof kernel stats.
#if SF_CONV1_TRACE
// experiment builds only (tools/conv1_trace.py): read and clear the per-phase cycle sums of k_conv1_u8_bf16
extern "C" int sf_debug_conv1_trace(unsigned long long *host_out12) {
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(host_out12, HIP_SYMBOL(sf_conv1_trace_acc), 12 * sizeof(unsigned long long)) != hipSuccess) return 1;
    unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return hipMemcpyToSymbol(HIP_SYMBOL(sf_conv1_trace_acc), z, sizeof(z)) == hipSuccess ? 0 : 1;
}
#endif
extern "C" int sf_conv_kernel_name(int op, int64_t n, const sf_conv_desc *h_desc, int split_k_allowed, char *out,
                                   int cap) {
    int rc = check_desc(h_desc, "sf_conv_kernel_name");
    if (rc) return rc;
    SF_REQUIRE(out && cap >= 48 && n > 0 && op >= 0 && op <= 3, "sf_conv_kernel_name: bad args");
    const ConvG g = make_geom(h_desc);
    const int64_t Mtot = n * g.OH * g.OW;
    const int mode = pick_mode(g);
    if (op == 0 && conv1_img_ok(g, mode, n)) {
        static const int wide_on = getenv("SF_CONV1_WIDE") ? atoi(getenv("SF_CONV1_WIDE")) : 1;
        if (conv1_bf16_ok(g, MODE_U8, n))
            snprintf(out, cap, "k_conv1_u8_bf16%s<%s>", wide_on && g.Cout == 32 ? "_w" : "", g.sub_mean != 0.f ? "true" : "false");
        else snprintf(out, cap, g.sub_mean != 0.f ? "k_conv_u8_img<2, 4, 5, 16, true>" : "k_conv_u8_img<2, 4, 5, 16, false>");
    } else if (op == 0) {
        const FwdPlan p = plan_fwd(Mtot, g.Cout, g.K, split_k_allowed ? (int64_t)1 << 60 : 0);
        const bool big32 = p.splits == 1 && Mtot >= 256 * 2048;
        if (p.cfg == 0) snprintf(out, cap, "k_conv_fwd<%d, 32, 4, 1, %d>", big32 ? 256 : 128, mode);
        else snprintf(out, cap, "k_conv_fwd<%d, 64, 2, 2, %d>", p.cfg == 1 ? 128 : 64, mode);
    } else if (op == 3) {
        if (narrow_fwd_ok(h_desc, n)) snprintf(out, cap, "k_linear_narrow<%d>", g.Cout <= 16 ? 1 : 2);
        else if (img_fwd_index(g, n) >= 0) snprintf(out, cap, "k_fwd_img<%d, %d, %d, %d, %d, 2, 1, %d>", g.Cin, g.H, g.W, g.KH, g.S, g.OH);
        else {
            const GldsFwdPlan q = plan_fwd_t(Mtot, g.Cout, g.K);
            const bool zl = glds_zl_on() && (!q.wide || glds_zl_on() >= 2) && (n - 1) * (int64_t)(g.H * g.W * g.Cin) + (int64_t)g.H * g.W * g.Cin < (1LL << 30) &&
                            (int64_t)g.Cout * g.K < (1LL << 30);  // (dense samples: the stride the model launches with)
            const int qz = q.ok ? q.Z : 1;
            if (zl && !q.sq64 && !q.wide && fwd_tail_split(g, Mtot, qz, true) > 0) snprintf(out, cap, "k_fwd_glds_zt<128, 64, 2, 2>");
            else if (zl) snprintf(out, cap, q.sq64 ? "k_fwd_glds_z<64, 64, 2, 2>" : q.wide ? "k_fwd_glds_z<128, 128, 2, 2>" : "k_fwd_glds_z<128, 64, 2, 2>");
            else
            snprintf(out, cap, q.sq64 ? "k_fwd_glds<64, 64, 2, 2, 2>" : q.wide ? "k_fwd_glds<128, 128, 2, 2, 2>" : "k_fwd_glds<128, 64, 2, 2, 2>");
        }
    } else if (op == 1 && small_linear_wgrad_ok(h_desc)) {
        snprintf(out, cap, "k_linear_wgrad_small");
    } else if (op == 1 && conv1_img_ok(g, mode, n) && g.Cout == 32) {
        if (conv1_bf16_ok(g, MODE_U8, n)) snprintf(out, cap, g.sub_mean != 0.f ? "k_conv1_wgrad_bf16<true>" : "k_conv1_wgrad_bf16<false>");
        else snprintf(out, cap, g.sub_mean != 0.f ? "k_conv1_wgrad_img<2, 4, true>" : "k_conv1_wgrad_img<2, 4, false>");
    } else if (op == 1 && mode == MODE_F32 && wgrad_img_variant(h_desc, n)) {
        snprintf(out, cap, wgrad_img_variant(h_desc, n) == 1 ? "k_wgrad_img<64, 9, 9, 3, 1, 1>" : "k_wgrad_img<32, 20, 20, 4, 2, 2>");
    } else if (op == 1 && mode == MODE_F32 && wgrad_glds_wanted(Mtot, g.K, g.Cout) && (int64_t)n * g.H * g.W * g.Cin < ((int64_t)1 << 30)) {
        const WgradGlds q = plan_wgrad_glds(Mtot, g.K, g.Cout);
        static const int wzl = getenv("SF_WGRAD_ZL") ? atoi(getenv("SF_WGRAD_ZL")) : 1;
        const bool z = wzl && g.KH == 1 && g.KW == 1 && g.H == 1 && g.W == 1 && n * (int64_t)g.Cin < (1LL << 30) && Mtot * g.Cout < (1LL << 30);
        snprintf(out, cap, "k_wgrad_glds%s<%s>", z ? "_z" : "", q.cfg == 3 ? "64, 128, 2, 2" : q.cfg == 0 ? "256, 64, 4, 1" : q.cfg == 1 ? "128, 128, 2, 2" : "128, 64, 2, 2");
    } else if (op == 1) {
        if (wgrad_bn(g.Cout) == 32) snprintf(out, cap, "k_conv_wgrad<32, 4, 1, %d>", mode);
        else snprintf(out, cap, "k_conv_wgrad<64, 2, 2, %d>", mode);
    } else {
        const int Hc = (g.H + g.S - 1) / g.S, Wc = (g.W + g.S - 1) / g.S;
        const int64_t Mc = n * Hc * Wc;
        const char *v = g.vecB ? "true" : "false";
        const bool zlf = n * (int64_t)g.Cout < (1LL << 30) && (int64_t)g.Cin * g.Cout < (1LL << 30);
        if (g.vecB && linear_dgrad_glds_ok(g, n)) snprintf(out, cap, glds_zl_on() >= 2 && zlf ? "k_fwd_glds_z<128, 128, 2, 2>" : "k_fwd_glds<128, 128, 2, 2, 2>");
        else if (g.vecB && linear_dgrad_glds64_ok(g, n)) snprintf(out, cap, glds_zl_on() && zlf ? "k_fwd_glds_z<64, 64, 2, 2>" : "k_fwd_glds<64, 64, 2, 2, 2>");
        else if (g.vecB && g.Cout % 32 == 0 && n >= 1024 && g.S > 1 && g.KH % g.S == 0 && g.KW % g.S == 0 && g.W % g.S == 0)
            snprintf(out, cap, dgrad_quadrow_zl(g, n) ? "k_dgrad_quadrow_z<128, 128, 2, 2>" : "k_dgrad_quadrow<128, 128, 2, 2>");
        else if (g.vecB && g.Cout % 32 == 0 && n >= 1024)
            snprintf(out, cap, "k_dgrad_pix%s<%s>", dgrad_pix_zl(g, n) ? "_z" : "", g.Cin <= 32 ? "256, 32, 4, 1" : "128, 64, 2, 2");
        else if (g.Cin <= 32) snprintf(out, cap, "k_conv_dgrad<128, 32, 4, 1, %s>", v);
        else if (Mc * ((g.Cin + 63) / 64) < 128LL * 1024) snprintf(out, cap, "k_conv_dgrad<64, 64, 2, 2, %s>", v);
        else snprintf(out, cap, "k_conv_dgrad<128, 64, 2, 2, %s>", v);
    }
    return SF_OK;
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
    return sf_conv_fwd(in, K, nullptr, 0, w, bias, out, M, &d, nullptr, 0, stream);
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
    const sf_conv_desc d = linear_desc(K, N, in_act ? 1 : 0);  // this wrapper's contract: ReLU mask of in_act
    return sf_conv_dgrad(dout, w, in_act, din, M, &d, stream);
}

extern "C" int sf_relu_mask(float *g, const float *act, int64_t n, void *stream) {
    SF_REQUIRE(g && act && n >= 0, "sf_relu_mask: bad args");
    if (n == 0) return SF_OK;
    const unsigned blocks = cdiv64(n, 256) < 4096 ? cdiv64(n, 256) : 4096;
    k_relu_mask<<<dim3(blocks), dim3(256), 0, STREAM(stream)>>>(g, act, n);
    return sf_launch_status("sf_relu_mask");
}

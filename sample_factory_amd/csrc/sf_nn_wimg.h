// LDS-image WEIGHT GRADIENT for the convolutions behind conv1 with Cout = 64 and Cin in {32, 64} (Nature-CNN conv2:
// 32 x 20 x 20, 4x4 stride 2; conv3: 64 x 9 x 9, 3x3 stride 1):
//   dW[(kh, kw, c)][n] = sum_{sample, oh, ow} X[sample][S*oh + kh][S*ow + kw][c] * dY[sample][oh][ow][n]
//
// The im2col DMA kernel (k_wgrad_glds) moves every input element KS*KS/S^2 times through the vector-memory path and the
// dY rows once per weight-row tile (conv3: 5 tiles for K = 576): 4.55 GB per launch at n = 32768 against 1.09 GB of
// operands (profiles/r02_b_traffic.json) at 0.63 of the f32-MFMA peak.  Here
//   * work-groups are PERSISTENT over a contiguous run of samples; a sample's X image and dY image enter LDS exactly once
//     (LDS-DMA, two stages, one sample ahead): vector-memory traffic == the operands;
//   * per sample the gradient is ONE small GEMM [K rows] x [64 cols] with the OH*OW output pixels as the reduction index
//     (padded to a multiple of 4: 4 pixels per v_mfma_f32_16x16x4_f32).  The whole K x 64 accumulator lives in registers
//     for the whole launch: wave (nt, ks) owns output columns 16nt .. 16nt+15 and the row tiles of tap group ks
//     (conv3: 4 waves x 144 accumulator registers, two work-groups per CU; conv2: 8 waves x 64, one work-group per CU),
//     one partial per work-group, summed in fixed order by k_reduce_partials;
//   * im2col happens in the LDS ADDRESS: lane (i = row-in-tile, g = pixel-in-quad) reads the CT = Cin/16 channels
//     CT*i .. CT*i+CT-1 of its pixel at tap (kh, kw) with one ds_read_b128 / b64 whose tap part is an immediate — one LDS
//     read per CT MFMAs (the CT channel tiles: row i of tile j is channel CT*i + j), no VALU, no vector memory in the
//     pixel loop; the dY operand of a pixel quad is one ds_read_b32 reused by all row tiles; its running sum is the bias
//     gradient.
// Padded pixels: the dY slot is zero there (the DMA lanes past the image fetch a zero page), the X reads land in the
// zero-filled tail of the X slot or in the dY slot behind it (finite): 0 * finite = 0.
#pragma once

template <int CIN, int HH, int WW, int KS, int ST, int KSPLIT>
struct WimgGeom {
    static constexpr int OH = (HH - KS) / ST + 1, OW = (WW - KS) / ST + 1, P = OH * OW, STEPS = (P + 3) / 4, TAPS = KS * KS;
    static constexpr int CT = CIN / 16, TAPS_W = TAPS / KSPLIT, NW = 4 * KSPLIT;
    static constexpr int XB = HH * WW * CIN * 4, YB = P * 64 * 4;                 // image bytes
    static constexpr int XI = (XB + 1023) / 1024, YI = (4 * STEPS * 256 + 1023) / 1024;  // 1-KiB DMA instructions
    static constexpr int XSLOT = XI * 1024, YSLOT = YI * 1024, STAGE = XSLOT + YSLOT;
    static constexpr int NI = (XI + YI + NW - 1) / NW;                             // DMA instructions per wave and sample
    // furthest X byte a padded pixel can touch: pixel 4*STEPS-1 at the last tap, + one channel row
    static constexpr int PMAX = 4 * STEPS - 1;
    static constexpr int XREACH = (((ST * (PMAX / OW) + KS - 1) * WW + ST * (PMAX % OW) + KS - 1) * CIN + CIN) * 4;
    static_assert(CT == 2 || CT == 4, "Cin = 32 or 64: one float2 / float4 per lane");
    static_assert(TAPS % KSPLIT == 0 && TAPS_W % KS == 0, "a wave's tap group is whole filter rows");
    static_assert(XREACH <= STAGE, "padded-pixel reads must stay inside the stage");
    static_assert(2 * STAGE <= 160 * 1024 / (KSPLIT == 1 ? 2 : 1), "LDS: two work-groups per CU (4 waves) or one (8 waves)");
};

template <int CIN, int HH, int WW, int KS, int ST, int KSPLIT>
__global__ __launch_bounds__(256 * KSPLIT, 2) void k_wgrad_img(const float *__restrict__ in, int64_t in_stride,
                                                               const float *__restrict__ dy, float *__restrict__ partial,
                                                               float *__restrict__ partial_b, int nsamples) {
    typedef WimgGeom<CIN, HH, WW, KS, ST, KSPLIT> G;
    constexpr int OW = G::OW, STEPS = G::STEPS, TAPS_W = G::TAPS_W, CT = G::CT, NW = G::NW, N = 64, K = G::TAPS * CIN;
    constexpr int XB = G::XB, YB = G::YB, XI = G::XI, NI = G::NI, XSLOT = G::XSLOT, STAGE = G::STAGE;
    typedef float fvec __attribute__((ext_vector_type(CT)));
    __shared__ __attribute__((aligned(1024))) char lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kg = lane >> 4;
    const int nt = wave & 3, ks = wave >> 2;  // column tile, tap group
    // ---- this work-group's samples: a contiguous run
    const int per = (nsamples + (int)gridDim.x - 1) / (int)gridDim.x;
    const int s_beg = (int)blockIdx.x * per, s_end = min(nsamples, s_beg + per);
    f32x4 acc[TAPS_W][CT];
#pragma unroll
    for (int t = 0; t < TAPS_W; ++t)
#pragma unroll
        for (int j = 0; j < CT; ++j) acc[t][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    if (s_beg < s_end) {
        // ---- loader: instruction q = wave + NW*j of a sample; q < XI: 1 KiB of the X image, else of the dY image.  Lanes
        // past the image fetch zeros (the slot tails must be zero: padded pixels).
        int doff[NI];       // byte offset inside the sample's image, or -1: zero page
        bool is_y[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int q = wave + NW * j;
            is_y[j] = q >= XI;
            const int b = (is_y[j] ? q - XI : q) * 1024 + lane * 16;
            doff[j] = (b < (is_y[j] ? YB : XB)) ? b : -1;
        }
        const char *inb = reinterpret_cast<const char *>(in), *dyb = reinterpret_cast<const char *>(dy);
        const char *zero = reinterpret_cast<const char *>(sf_zero_page) + (lane & 7) * 16;
        auto issue = [&](int s, int stage) {
            const char *xs = inb + (int64_t)s * in_stride * 4, *ys = dyb + (int64_t)s * YB;
            char *st = lds + stage * STAGE;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int q = wave + NW * j;
                if (q < XI + G::YI) {
                    const char *src = doff[j] < 0 ? zero : (is_y[j] ? ys : xs) + doff[j];
                    GLDS16(src, st + (is_y[j] ? XSLOT + (q - XI) * 1024 : q * 1024));
                }
            }
        };
        // ---- fragment addresses: X of pixel p = 4*step + kg at this wave's first tap (channels CT*i16 ..), dY of
        // (pixel, column)
        const int tapbase = ks * (TAPS_W / KS) * WW * CIN * 4;
        int xo[STEPS];
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int p = 4 * st + kg, oh = p / OW, ow = p - oh * OW;
            xo[st] = ((ST * oh * WW + ST * ow) * CIN + CT * i16) * 4 + tapbase;
        }
        const int yo = XSLOT + kg * 256 + (16 * nt + i16) * 4;

        auto sample = [&](auto stage_c) {
            constexpr int SOFF = decltype(stage_c)::value * STAGE;
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                const float b = *reinterpret_cast<const float *>(lds + SOFF + yo + st * 1024);
                bsum += b;
#pragma unroll
                for (int t = 0; t < TAPS_W; ++t) {
                    const int imm = SOFF + ((t / KS) * WW + (t % KS)) * CIN * 4;
                    const fvec a = *reinterpret_cast<const fvec *>(__builtin_assume_aligned(lds + xo[st] + imm, 4 * CT));
#pragma unroll
                    for (int j = 0; j < CT; ++j)
                        acc[t][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b, acc[t][j], 0, 0, 0);
                }
            }
        };
        issue(s_beg, 0);
        for (int s = s_beg; s < s_end; s += 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // sample s has landed in stage 0; everybody is done reading stage 1 (sample s-1)
            if (s + 1 < s_end) issue(s + 1, 1);
            sample(std::integral_constant<int, 0>{});
            if (s + 1 < s_end) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (s + 2 < s_end) issue(s + 2, 0);
                sample(std::integral_constant<int, 1>{});
            }
        }
    }
    // ---- this work-group's partial: row k = tap*CIN + CT*(4*kg + r) + j, column 16*nt + i16 (blocks without samples
    // write zeros: the reduction reads every partial)
    float *dst = partial + (int64_t)blockIdx.x * K * N + 16 * nt + i16;
#pragma unroll
    for (int t = 0; t < TAPS_W; ++t)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[((ks * TAPS_W + t) * CIN + CT * (4 * kg + r) + j) * N] = acc[t][j][r];
    if (partial_b && ks == 0) {
        bsum += __shfl_xor(bsum, 16, 64);
        bsum += __shfl_xor(bsum, 32, 64);
        if (kg == 0) partial_b[(int64_t)blockIdx.x * N + 16 * nt + i16] = bsum;
    }
}

// LDS-image DATA GRADIENT for stride-1 convolutions with Cin = Cout = 64 (Nature-CNN conv3: 64 x 9 x 9, 3x3):
//   din[s][ih][iw][c] = act'(x[s][ih][iw][c]) * sum_{kh,kw,n} dY[s][ih - kh][iw - kw][n] * W[(kh, kw, c)][n]
//
// The pixel-major kernel (k_dgrad_pix) tiles 128 SAMPLES at one input pixel: exact (border taps skipped), but its
// operands come through the vector-memory path 2.3x (profiles/r02_b_traffic.json) and its epilogue touches one scattered
// 256-byte piece per sample and row (activation read + gradient write): 0.60 of the f32-MFMA peak.  Here the product is
// evaluated in SCATTER form, which has no structural zeros either:
//   for every filter tap (kh, kw):   T[p][c] = sum_n dY[p][n] * W[(kh, kw, c)][n]      p = the sample's 49 dY pixels
//                                    din_image[(oh + kh, ow + kw)][c] += T[p][c]
//   * work-groups are PERSISTENT (two per CU) over a contiguous run of sample PAIRS; a pair's dY images (2 x 12.5 KB)
//     enter LDS once by LDS-DMA, the pair's output images (2 x 20.7 KB, f32) are ACCUMULATED IN LDS (read-add-write);
//   * wave w owns input channels 16w .. 16w+15 with its weights for ALL taps in registers (9 taps x 64 n = 144 VGPRs,
//     loaded once per kernel); rows of an MFMA tile are 16 of the pair's 98 dY pixels (7 tiles: 12.5 % padding), the
//     reduction runs over n; one ds_read_b128 of dY feeds 4 MFMAs and is reused by all 9 taps; taps are processed three
//     at a time so that no accumulator is reused inside the 40-cycle MFMA latency;
//   * every contribution to an output element comes from ONE wave in program order: the LDS accumulation is
//     deterministic;
//   * after the last tap: the finished images are read back, multiplied by the activation derivative (the activation is
//     read with coalesced 16-byte loads, a whole contiguous image per sample), stored with coalesced 16-byte stores and
//     the LDS images are zeroed; the next pair's dY DMA is in flight meanwhile.
#pragma once

template <int HH, int WW, int KS>
struct DimgGeom {
    static constexpr int C = 64, N = 64, OH = HH - KS + 1, OW = WW - KS + 1, P = OH * OW, TAPS = KS * KS;
    static constexpr int BS = 2, ROWS = BS * P, TILES = (ROWS + 15) / 16;
    static constexpr int YB = P * N * 4;                                   // dY bytes per sample
    static constexpr int YI = (BS * YB + 1023) / 1024;                      // DMA instructions per pair
    static constexpr int YSLOT = (TILES * 16 * N * 4 + 1023) / 1024 * 1024; // covers the padded rows' reads
    static constexpr int OB = HH * WW * C * 4;                              // output image bytes per sample
    static constexpr int OSLOT = BS * OB;
    static constexpr int NI = (YI + 3) / 4;
    // where the padded rows' contributions go: 256 lanes behind the images + the largest tap offset
    static constexpr int DUMP = (((KS - 1) * WW + KS - 1) * C + 256) * 4;
    static_assert(YSLOT + OSLOT + DUMP <= 80 * 1024, "two work-groups per CU");
    static_assert(KS == 3, "taps are processed one filter column (3 taps) at a time");
};

template <int HH, int WW, int KS, bool MASK>
__global__ __launch_bounds__(256, 2) void k_dgrad_img(const float *__restrict__ dy, const float *__restrict__ w,
                                                      const float *__restrict__ in_act, float *__restrict__ din,
                                                      int nsamples) {
    typedef DimgGeom<HH, WW, KS> G;
    constexpr int C = G::C, N = G::N, OW = G::OW, P = G::P, TAPS = G::TAPS, BS = G::BS, ROWS = G::ROWS, TILES = G::TILES;
    constexpr int YB = G::YB, YI = G::YI, YSLOT = G::YSLOT, OB = G::OB, OSLOT = G::OSLOT, NI = G::NI;
    __shared__ __attribute__((aligned(1024))) char lds[YSLOT + OSLOT + G::DUMP];
    char *ylds = lds;
    float *olds = reinterpret_cast<float *>(lds + YSLOT);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kg = lane >> 4;
    const int npairs = (nsamples + BS - 1) / BS;
    const int per = (npairs + (int)gridDim.x - 1) / (int)gridDim.x;
    const int p_beg = (int)blockIdx.x * per, p_end = min(npairs, p_beg + per);
    if (p_beg >= p_end) return;
    // ---- weights of this wave's 16 input channels, all taps: lane (c = i16, kg) holds W[(tap, 16w + c)][16s + 4kg + j]
    f32x4 breg[TAPS][4];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s)
            breg[t][s] = *reinterpret_cast<const f32x4 *>(
                __builtin_assume_aligned(w + (int64_t)(t * C + 16 * wave + i16) * N + 16 * s + 4 * kg, 16));
    // ---- zero the accumulation images once (afterwards the epilogue leaves them zeroed)
    for (int i = tid; i < OSLOT / 16; i += 256) reinterpret_cast<f32x4 *>(olds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // ---- dY loader: instruction q = wave + 4j covers bytes [1024q, 1024q + 1024) of the pair's two images
    const char *dyb = reinterpret_cast<const char *>(dy);
    const char *zero = reinterpret_cast<const char *>(sf_zero_page) + (lane & 7) * 16;
    auto issue = [&](int pair) {
        const int64_t base = (int64_t)pair * BS * YB, limit = (int64_t)nsamples * YB;  // an odd last sample: zeros
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int q = wave + 4 * j;
            if (q < YI) {
                const int b = q * 1024 + lane * 16;  // (recomputed per pair: registers are the scarce resource here)
                const bool ok = b < BS * YB && base + b < limit;
                GLDS16(ok ? dyb + base + b : zero, ylds + q * 1024);
            }
        }
    };
    // ---- per (tile, r): LDS float index of the output element of row 16*tile + 4*kg + r at tap (0, 0), channel 16w + i16
    int obase[TILES][4];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + 4 * kg + r;
            const int sl = row / P, p = row - sl * P, oh = p / OW, ow = p - oh * OW;
            // padded rows (past the pair's 98 pixels) accumulate garbage into a dump area behind the images: no predicate,
            // no branch between the MFMAs and the LDS adds
            obase[t][r] = row < ROWS ? ((sl * HH + oh) * WW + ow) * C + 16 * wave + i16 : OSLOT / 4 + tid;
        }
    const int arow = i16 * (N * 4) + 16 * kg;  // byte offset of this lane's dY row / n-quad inside a tile

    issue(p_beg);
    for (int pair = p_beg; pair < p_end; ++pair) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // this pair's dY has landed; the output images are zero
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            f32x4 a[4];
#pragma unroll
            for (int s = 0; s < 4; ++s)
                a[s] = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(ylds + t * (16 * N * 4) + arow + s * 64, 16));
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {  // three taps at a time: (kh = 0..2, kw) — one COLUMN of the filter
                f32x4 acc[KS];
#pragma unroll
                for (int u = 0; u < KS; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int u = 0; u < KS; ++u)
                            acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s][j], breg[u * KS + kw][s][j], acc[u], 0, 0, 0);
                // accumulate into the LDS images with plain read-add-write: every element is only ever touched by THIS wave
                // (it owns the channel) and a wave's LDS operations execute in program order, so a later group's read sees
                // an earlier group's write.  The 12 elements of one group (4 consecutive dY pixels x 3 filter ROWS) are 12
                // different addresses — pixel indices of the rows differ by 1..5, those of the filter rows by 9 and 18 —
                // so all reads go out together: one LDS round trip per 48 MFMAs.  (ds_add_f32 is correct too but
                // serialises the lanes: 5.5 ms per launch at n = 32768 against 1.27 ms for k_dgrad_pix.)
                float old[KS][4];
#pragma unroll
                for (int u = 0; u < KS; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) old[u][r] = olds[obase[t][r] + (u * WW + kw) * C];
#pragma unroll
                for (int u = 0; u < KS; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) olds[obase[t][r] + (u * WW + kw) * C] = old[u][r] + acc[u][r];
            }
        }
        __syncthreads();  // every contribution is in; dY is free again
        if (pair + 1 < p_end) issue(pair + 1);
        // ---- epilogue: activation derivative, coalesced stores, re-zero the images
        const int s0 = pair * BS;
#pragma unroll 2
        for (int i = tid; i < OSLOT / 16; i += 256) {
            const int sl = i / (OB / 16), e = i - sl * (OB / 16);
            f32x4 v = reinterpret_cast<const f32x4 *>(olds)[i];
            reinterpret_cast<f32x4 *>(olds)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (s0 + sl < nsamples) {
                const int64_t g = ((int64_t)(s0 + sl) * OB) / 16 + e;
                if (MASK) {
                    const f32x4 x = reinterpret_cast<const f32x4 *>(in_act)[g];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = x[j] > 0.f ? v[j] : 0.f;
                }
                reinterpret_cast<f32x4 *>(din)[g] = v;
            }
        }
    }
}

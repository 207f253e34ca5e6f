// LDS-image forward for the stride-S convolutions behind conv1 (Nature-CNN conv2 / conv3 and relatives):
//   out[m][n] = act( sum_k A[m][k] * Wt[n][k] + bias[n] ),  m = (sample, oh, ow),  k = (kh*KS + kw)*CIN + c,  Cout = 64.
//
// The im2col DMA kernel (k_fwd_glds) moves every input element KS*KS/S^2 times through the vector-memory path (conv3:
// 9x, conv2: 4x) plus the whole weight matrix once per 128-row tile: 3.45 KB per output row for conv3, and the DMA
// issue is what holds it at 0.71 MFMA-busy.  Here
//   * a work-group is PERSISTENT (one per CU) and owns a contiguous range of output rows = a contiguous run of samples;
//     each sample's NHWC image is copied into LDS exactly once (ring of RING image slots, LDS-DMA, one sample ahead);
//   * im2col happens in the LDS ADDRESS: lane (row = output pixel, kg) reads the 4 consecutive channels of its tap with
//     one ds_read_b128 whose tap / channel-block part is an IMMEDIATE offset (compile-time geometry), so the k-loop has
//     no VALU and no vector-memory instruction at all: 1 ds_read_b128 per 4 MFMAs;
//   * the 4 waves split the 64 output channels; a wave's 16 weight columns for the WHOLE reduction live in registers
//     (K/16 x f32x4 = 144 VGPRs for conv3), loaded once per kernel: the weight matrix is read 256 times per launch
//     instead of once per tile (conv3 at n = 32768: 12544 times);
//   * v_mfma_f32_16x16x4_f32: 16-row fragments, so the only padding is the last fragment of the launch.
// LDS image layout (16-byte chunks): [channel chunk c][w-parity pw][ih][iw / S] — consecutive output pixels of a row
// read consecutive chunks (conflict-free for any stride), and (kh, kw, c) only move the immediate.
// Vector-memory traffic: one image per sample + 147 KB of weights per CU: 0.6 KB per output row (conv3).
#pragma once

// R = output rows per UNIT: a unit is a horizontal strip of one sample (R = OH: the whole image).  Units are numbered
// sample-major, so output row m = unit * (R*OW) + pixel-in-strip — the same flat row space as without strips; a strip
// needs HU = (R-1)*S + KS input rows (consecutive strips overlap by KS - S rows, loaded twice).
// SF_IMG_FLIP (round 6): the image rows are stored BOTTOM-UP in LDS and every channel-chunk plane is padded to a multiple of
// 16 chunks.  ds_read_b128 is served in groups of 16 lanes, one LDS cycle per group when their 16-byte chunks fall into 16
// different residues mod 16 (64 banks x 4 B); a group holds every fragment row exactly once, half of them from plane kg and
// half from plane kg + 1.  Top-down rows at pitch WQ = 9 put output pixel j = oh*7 + ow at chunk 9*oh + ow = j + 2*(j/7): the
// +2 at every row wrap makes rows of one fragment collide (and PLANE = 81 shifts the odd plane by one more) — 2-3 LDS cycles
// per group, SQ_LDS_BANK_CONFLICT 6 x the active cycles (profiles/r04_f_stall_counters.txt).  Bottom-up, the same pixel sits
// at 9*(6 - oh) + ow = 54 - 9*oh + ow, and because 9 = -7 (mod 16) that is 54 + 7*oh + ow = 54 + j (mod 16): LINEAR in the
// output pixel index, so any 16 consecutive pixels of a sample hit 16 different residues; with PLANE = 0 (mod 16) both planes
// of a group see the same residues.  Only fragments that straddle two samples keep one 2-way pair.  The filter tap stays an
// immediate ((KS-1-kh)*WQ + kw), products and summation order are unchanged: bit-identical results.
#ifndef SF_IMG_FLIP
#define SF_IMG_FLIP 1
#endif
#ifndef SF_IMG_PRIO
#define SF_IMG_PRIO 0
#endif
#ifndef SF_IMG_HALFSTEP
#define SF_IMG_HALFSTEP 1  // a last step with one live fragment runs the one-fragment k-loop (see the kernel)
#endif
template <int CIN, int H, int W, int KS, int ST, int TMF, int WSETS, int R>
struct ImgFwdGeom {
    static constexpr int OH = (H - KS) / ST + 1, OW = (W - KS) / ST + 1, OHW = R * OW;  // rows per unit
    static constexpr int U = OH / R, HU = (R - 1) * ST + KS;  // units per sample, input rows per unit
    static constexpr int K = KS * KS * CIN, KG = K / 16;  // 16-deep reduction groups (one f32x4 of weights per lane)
    static constexpr int C4 = CIN / 4, WQ = W / ST;       // 16-byte chunks per pixel, columns per w-parity class
    static constexpr int PLANE_USED = ST * HU * WQ;       // image chunks per channel chunk
    static constexpr int PLANE = SF_IMG_FLIP ? (PLANE_USED + 15) / 16 * 16 : PLANE_USED;  // plane pitch in chunks
    static constexpr int IMG_CH = C4 * PLANE;             // chunks per image
    static constexpr int IMG_B = (IMG_CH * 16 + 1023) / 1024 * 1024;  // slot size: whole 1-KiB DMA instructions
    // samples touched by two consecutive row blocks (the one being multiplied + the one being fetched)
    static constexpr int BROWS = WSETS * TMF * 16;  // output rows per block step (WSETS wave sets x TMF fragments)
    static constexpr int RING = (2 * BROWS - 2) / OHW + 2;
    static_assert(CIN % 16 == 0 && W % ST == 0 && K % 16 == 0 && OH % R == 0, "geometry");
    static_assert(RING * IMG_B <= 160 * 1024, "LDS");
};

template <int CIN, int H, int W, int KS, int ST, int TMF, int WSETS, int R>
__global__ __launch_bounds__(256 * WSETS, 1) void k_fwd_img(const float *__restrict__ in, int64_t in_stride,
                                                    const float *__restrict__ wt, const float *__restrict__ bias,
                                                    float *__restrict__ out, int nsamples, int act) {
    typedef ImgFwdGeom<CIN, H, W, KS, ST, TMF, WSETS, R> G;
    constexpr int NW = 4 * WSETS, TB = TMF * WSETS;  // waves per block, fragments per block step
    constexpr int OW = G::OW, OHW = G::OHW, K = G::K, KG = G::KG, WQ = G::WQ, PLANE = G::PLANE, RING = G::RING;
    constexpr int IMG_B = G::IMG_B, N = 64, U = G::U, HU = G::HU;
    __shared__ __attribute__((aligned(1024))) char ring[RING * IMG_B];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, kg = lane >> 4, n = (wave & 3) * 16 + col, wset = wave >> 2;
#if SF_IMG_PRIO
    // experiment: the two co-resident work-groups of a CU (hardware wave slots of different parity) get different static
    // priorities, so that they do not run their per-step overhead (barrier, address set-up, DMA, stores) in phase
    if (__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1u) __builtin_amdgcn_s_setprio(SF_IMG_PRIO);
#endif
    const int nunits = nsamples * U;
    const int64_t Mtot = (int64_t)nunits * OHW;
    // ---- this work-group's rows: a contiguous run of 16-row fragments
    const int total_frags = (int)((Mtot + 15) >> 4);
    const int per = (total_frags + (int)gridDim.x - 1) / (int)gridDim.x;
    const int f_beg = (int)blockIdx.x * per, f_end = min(total_frags, f_beg + per);
    if (f_beg >= f_end) return;
    // ---- weights of this wave's 16 columns, whole reduction: lane (col, kg) holds Wt[n][16g + 4kg .. +3]
    f32x4 breg[KG];
#pragma unroll
    for (int g = 0; g < KG; ++g)
        breg[g] = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(wt + (int64_t)n * K + 16 * g + 4 * kg, 16));
    const float bv = bias ? bias[n] : 0.f;

    // ---- image loader: sample s -> slot s % RING; the 4 waves share the 1-KiB DMA instructions of an image.  The
    // per-lane source offsets of this wave's instructions are the same for every image: decoded once.
    constexpr int NDMA = IMG_B / 1024, NI = (NDMA + NW - 1) / NW;
    int srcoff[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        int q = (wave + NW * j) * 64 + lane;  // chunk index inside the slot: (c, pw, ih, iwq)
        q = q < G::IMG_CH ? q : 0;           // slot padding: any valid address
        const int c = q / PLANE;
        int r0 = q - c * PLANE;
        r0 = r0 < G::PLANE_USED ? r0 : 0;    // plane padding (SF_IMG_FLIP): any valid address
        const int pw = r0 / (HU * WQ), r1 = r0 - pw * (HU * WQ);
        const int ihl = r1 / WQ, iwq = r1 - ihl * WQ;
        const int ih = SF_IMG_FLIP ? HU - 1 - ihl : ihl;  // LDS row ihl holds image row ih
        srcoff[j] = (ih * W + iwq * ST + pw) * CIN + c * 4;
    }
    auto load_image = [&](int s) {  // s: unit index
        const int u = s < nunits ? s : nunits - 1, smp = u / U, strip = u - smp * U;
        const float *img = in + (int64_t)smp * in_stride + strip * (R * ST * W * CIN);
        char *dst = ring + (s % RING) * IMG_B;
#pragma unroll
        for (int j = 0; j < NI; ++j)
            if (wave + NW * j < NDMA) GLDS16(img + srcoff[j], dst + (wave + NW * j) * 1024);
    };
    const uint32_t M32 = (uint32_t)Mtot;  // the launcher guarantees Mtot < 2^31
    auto last_sample = [&](int fb, int nf) {  // last sample a block of nf fragments starting at fragment fb touches
        uint32_t m = (uint32_t)(fb + nf) * 16u - 1u;
        m = m < M32 ? m : M32 - 1u;
        return (int)(m / (uint32_t)OHW);
    };
    int loaded = (int)(((uint32_t)f_beg * 16u) / (uint32_t)OHW);  // next sample to fetch
    {
        const int hi = last_sample(f_beg, min(TB, f_end - f_beg));
        for (; loaded <= hi; ++loaded) load_image(loaded);
    }
    // the finished block is stored during the NEXT block's MFMA phase: a store issued right before the s_waitcnt
    // vmcnt(0) of the block barrier would put its whole drain latency in front of every block (one wave per SIMD:
    // nothing else to run meanwhile)
    f32x4 pend[TMF];
    int pend_fb = -1, pend_nf = 0;
    auto store_pend = [&]() {
        float *ob = out + ((int64_t)pend_fb * 16 + 4 * kg) * N + n;
        const int rows_left = (int)(M32 - ((uint32_t)pend_fb * 16u + 4u * (uint32_t)kg));
#pragma unroll
        for (int f = 0; f < TMF; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (f < pend_nf && f * 16 + r < rows_left) ob[(f * 16 + r) * N] = pend[f][r];
    };
    for (int fb0 = f_beg; fb0 < f_end; fb0 += TB) {
        const int fb = fb0 + wset * TMF;                     // this wave set's fragments: fb .. fb + nf - 1
        const int nf = max(0, min(TMF, f_end - fb));         // (0: nothing left for this set in the last step)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // this block's images have landed; the previous block's reads are over
        if (fb0 + TB < f_end) {  // fetch what the NEXT step needs beyond what is there (never a slot still in use)
            const int hi = last_sample(fb0 + TB, min(TB, f_end - fb0 - TB));
            for (; loaded <= hi; ++loaded) load_image(loaded);
        }
        if (pend_fb >= 0) store_pend();
        // ---- per fragment: this lane's row -> LDS address of its patch origin (+ its channel chunk kg)
        int base[TMF];
#pragma unroll
        for (int f = 0; f < TMF; ++f) {
            uint32_t m = (uint32_t)(nf > 0 ? fb + (f < nf ? f : 0) : fb0) * 16u + (uint32_t)col;
            m = m < M32 ? m : M32 - 1u;
            const uint32_t s = m / (uint32_t)OHW, p = m - s * (uint32_t)OHW, oh = p / (uint32_t)OW, ow = p - oh * (uint32_t)OW;
            // row of the patch origin in LDS: top-down oh*ST, or bottom-up (R-1-oh)*ST with the tap's (KS-1-kh) added by the
            // immediate: (R-1-oh)*ST + (KS-1-kh) = HU-1 - (oh*ST + kh)
            const uint32_t prow = SF_IMG_FLIP ? ((uint32_t)(R - 1) - oh) * ST : oh * ST;
            base[f] = (int)((s % (uint32_t)RING) * (uint32_t)IMG_B + ((uint32_t)kg * PLANE + prow * WQ + ow) * 16u);
        }
        // The k-loop for NF live fragments (compile time).  SF_IMG_HALFSTEP: a work-group's LAST step may hold fewer fragments
        // than TMF (conv3 at a rollout step: 12544 fragments on 512 work-groups = 24.5 each, 13 steps of 2) — it then runs the
        // NF = 1 instantiation instead of multiplying a duplicate of fragment 0: the rollout-size launch is 12.5 step times
        // long instead of 13.  Same MFMAs in the same order for the live fragments: bit-identical.
        auto compute = [&](auto nfc) {
        constexpr int NF = decltype(nfc)::value;
        f32x4 acc[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
        // k-groups are processed in PAIRS: the fragment reads of the next pair sit between the two groups of the
        // current pair.  hipcc waits with lgkmcnt(0), i.e. for the youngest read, so what matters is the distance
        // from the LAST read to the wait: one whole group of 4*TMF MFMAs here (first version: reads right in front of
        // the wait, 18 exposed LDS latencies per block step = 25 % of the step with one wave per SIMD).
        static_assert(KG % 2 == 0, "k-groups come in pairs");
        f32x4 a[2][2][NF];
        auto fetch = [&](int g, int slot) {  // g is a compile-time constant after unrolling: the offset is an immediate
            const int tap = (16 * g) / CIN, cb = ((16 * g) % CIN) / 4, kh = tap / KS, kw = tap % KS;
            const int imm = (cb * PLANE + ((kw % ST) * HU + (SF_IMG_FLIP ? KS - 1 - kh : kh)) * WQ + kw / ST) * 16;
#pragma unroll
            for (int f = 0; f < NF; ++f)
                a[slot][g & 1][f] = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(ring + base[f] + imm, 16));
        };
        auto mfmas = [&](int g, int slot) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int f = 0; f < NF; ++f)
                    acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[slot][g & 1][f][j], breg[g][j], acc[f], 0, 0, 0);
        };
        fetch(0, 0);
        fetch(1, 0);
#pragma unroll
        for (int gp = 0; gp < KG / 2; ++gp) {
            mfmas(2 * gp, gp & 1);
            __builtin_amdgcn_sched_barrier(0);
            if (gp + 1 < KG / 2) {
                fetch(2 * gp + 2, (gp + 1) & 1);
                fetch(2 * gp + 3, (gp + 1) & 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfmas(2 * gp + 1, gp & 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue into the parking registers: pend[f][r] = out[(fb + f)*16 + 4*kg + r][n]
        auto park = [&](auto kc) {
            constexpr int KIND = decltype(kc)::value;
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int r = 0; r < 4; ++r) pend[f][r] = act_fwd_c<KIND>(acc[f][r] + bv, act);
        };
        if (act == 1) park(std::integral_constant<int, 1>{});
        else park(std::integral_constant<int, -1>{});
        };
        if (SF_IMG_HALFSTEP && TMF == 2 && nf == 1) compute(std::integral_constant<int, 1>{});
        else compute(std::integral_constant<int, TMF>{});
        pend_fb = fb;
        pend_nf = nf;
    }
    if (pend_fb >= 0) store_pend();
}

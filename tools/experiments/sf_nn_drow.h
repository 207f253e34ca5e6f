// EXPERIMENT (round 4, not wired): stride-1 data gradient that reuses every staged dY chunk across the KW filter columns.
// Built to cut the 9x re-fetch of dY in k_dgrad_pix (3.3 GB of fetches per conv3 launch).  It does cut the DMA work 3x —
// and is SLOWER (MI355X, n = 32768, conv3, tools/kbench.py; profiles/r04_dgrad_row_ablation.log):
//     k_dgrad_pix<128,64>                      1294 us  (1246 us with the longest-rows-first block order kept in the product)
//     k_dgrad_row<128,64,4,2,3> (8 waves)      1285 .. 1460 us depending on the fragment-read schedule
//     k_dgrad_row<64,64,2,2,3>  (2 blocks/CU)  1341 .. 1390 us
// Ablation of the 64-row form (same launch, pieces compiled out): no epilogue loads/stores 1239, dY DMA from one hot region
// 1302, no DMA at all 1191, neither 1073, + no stage barrier 1083, + no fragment reads 1033, all of them 1007 us = 117 TFLOP/s:
// the skeleton alone (MFMAs + per-column park + per-block prologue / tail + uneven rows) is 30 % above the pure MFMA time, and
// the re-fetches this kernel removes were worth 3 % (they hit L2 / Infinity Cache).  What it taught: (1) row blocks of 1..3
// filter rows dispatched in row order idle the chip by 9-12 % -> longest-first order, kept in k_dgrad_pix (-3.7 %);
// (2) traffic is not what holds the data gradient at 0.6 of the matrix pipe.
// To build it again: #include this file from sf_nn_glds.h behind k_dgrad_pix and launch it as
//   k_dgrad_row<128, 64, 4, 2, 3><<<tiles8 * 8 * H * ctiles, 512>>>(g, dout, w, in_act, din, n, ntiles, tiles8, lpt).
#pragma once

// ============================================================================================== DATA GRADIENT, stride 1, tap reuse
// k_dgrad_pix fetches every dY chunk once per (input pixel, tap) that uses it: KH*KW times (conv3: 9x, 3.3 GB of fetches
// per launch against 0.41 GB of dY), and every 32-deep chunk is one barrier + 6 DMA instructions per 32 MFMAs of a wave.
// For stride 1 the KW input pixels iw = ow .. ow + KW - 1 of a row all read the SAME dY pixel (oh, ow) — only the filter
// column kw = iw - ow differs.  So the block walks the OUTPUT columns ow of its input row ih, and a staged dY chunk
//   A[s][co] = dY[s, ih - kh, ow, co-chunk]
// is multiplied with the KW weight chunks B_kw[c][co] = W[(kh*KW + kw)*Cin + c][co-chunk] into KW accumulator sets, one per
// input pixel iw = ow + kw.  Pixel iw always lives in accumulator set iw % KW (no register moves); it is complete once
// column ow = iw has been processed, is parked / stored like in k_dgrad_pix, and its set restarts at zero for iw + KW.
// Per stage and wave: 4 + KW*2 DMA instructions and TM + KW*TN fragment reads for KW*32 MFMAs (KW = 3: 10 DMA per 96
// MFMAs instead of 18; one barrier per 96 instead of per 32); dY is fetched KH times instead of KH*KW times.
// One staged chunk: fragments are double-buffered in registers (the reads of 8-deep group c+1 are issued in the middle of
// group c's MFMAs, so the LDS latency never sits in front of an MFMA group); head() runs right behind the first group's
// reads (their latency covers it), mid(c) in the middle of group c (the DMA instructions of the NEXT stage, spread out).
template <int TM, int TN, int KWT, int ROT, int BN, typename FH, typename FM>
__device__ __forceinline__ void mma_chunk_rows_kw(const float *__restrict__ As, const float *__restrict__ Bs, int arow0,
                                                  int brow0, int lane, f32x16 (&acc)[KWT][TM][TN], FH &&head, FM &&mid) {
    const int r = lane & 31, h = lane >> 5, sw = (r >> 1) & 7;
    const float *ap = As + (arow0 + r) * 32, *bp = Bs + (brow0 + r) * 32;
    float4 a[2][TM], b[2][KWT][TN];
#ifndef SF_ROW_ABL
#define SF_ROW_ABL 0  // ablation bits (tools/build_variant.sh): 1 no epilogue traffic, 2 dY DMA from one hot region, 4 no DMA,
#endif                //   8 no stage barrier / vmcnt wait, 16 no fragment reads (registers keep whatever they hold)
    auto fetch = [&](int c) {
        if (SF_ROW_ABL & 16) {  // operands that cost nothing to produce
#pragma unroll
            for (int i = 0; i < TM; ++i) a[c & 1][i] = make_float4((float)lane, 1.f, 2.f, 3.f);
#pragma unroll
            for (int kw = 0; kw < KWT; ++kw)
#pragma unroll
                for (int i = 0; i < TN; ++i) b[c & 1][kw][i] = make_float4(1.f, (float)lane, 3.f, 2.f);
            return;
        }
        const int pos = (((2 * c + h) ^ sw) << 2);
#pragma unroll
        for (int i = 0; i < TM; ++i) a[c & 1][i] = *reinterpret_cast<const float4 *>(ap + i * 32 * 32 + pos);
#pragma unroll
        for (int kw = 0; kw < KWT; ++kw)
#pragma unroll
            for (int i = 0; i < TN; ++i)
                b[c & 1][kw][i] = *reinterpret_cast<const float4 *>(bp + (kw * BN + i * 32) * 32 + pos);
    };
    auto mfmas = [&](int c, int j) {
#pragma unroll
        for (int kw = 0; kw < KWT; ++kw)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[(ROT + kw) % KWT][tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                        j == 0 ? a[c & 1][tm].x : j == 1 ? a[c & 1][tm].y : j == 2 ? a[c & 1][tm].z : a[c & 1][tm].w,
                        j == 0 ? b[c & 1][kw][tn].x : j == 1 ? b[c & 1][kw][tn].y : j == 2 ? b[c & 1][kw][tn].z : b[c & 1][kw][tn].w,
                        acc[(ROT + kw) % KWT][tm][tn], 0, 0, 0);
    };
#ifndef SF_ROW_PIPE
#define SF_ROW_PIPE 0
#endif
#if SF_ROW_PIPE
    fetch(0);
    head();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        mfmas(c, 0);
        mfmas(c, 1);
        __builtin_amdgcn_sched_barrier(0);
        mid(c);
        if (c + 1 < 4) fetch(c + 1);
        __builtin_amdgcn_sched_barrier(0);
        mfmas(c, 2);
        mfmas(c, 3);
    }
#else  // everything the stage has to issue first, then the MFMA groups in the compiler's order
    head();
#pragma unroll
    for (int c = 0; c < 4; ++c) mid(c);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        fetch(c);
#pragma unroll
        for (int j = 0; j < 4; ++j) mfmas(c, j);
    }
#endif
}

template <int BM, int BN, int WM, int WN, int KWT>
__global__ __launch_bounds__(64 * WM * WN, 2) void k_dgrad_row(ConvG g, const float *__restrict__ dy, const float *__restrict__ w,
                                                      const float *__restrict__ in_act, float *__restrict__ din,
                                                      int nsamples, int ntiles, int tiles8, int lpt) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int NW = WM * WN;                            // 4 waves (one per SIMD) or 8 (two per SIMD, one block per CU)
    constexpr int AI = BM / (8 * NW), BI = BN / (8 * NW);  // DMA instructions per wave: 8 rows x 128 B each
    constexpr int STAGE = (BM + KWT * BN) * 32;
    static_assert((NW == 4 || NW == 8) && TM >= 1 && TN >= 1 && AI >= 1 && BI >= 1 && KWT >= 2 && KWT <= 3, "tile shape");
    __shared__ __attribute__((aligned(1024))) float lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int Cin = g.Cin, Cout = g.Cout, OH = g.OH, OW = g.OW, Wd = g.W;
    // (sample tile, input row, Cin tile) from the XCD-swizzled linear id: all rows of a sample tile on one XCD
    // Rows differ in work by the number of filter rows that reach them (conv3: 1, 2, 3, 3, 3, 3, 3, 2, 1 — a dispatch in
    // row order leaves the chip 12 % idle behind the last long blocks): rows are dealt LONGEST FIRST, i.e. centre-out
    // (rank 0 = the middle row), every sample tile of a rank before the next rank (SF_DGRAD_LPT=0: row-major ids).
    const uint32_t xcd = blockIdx.x & 7u, local = blockIdx.x >> 3;
    uint32_t t;
    int ih;
    if (lpt) {
        const uint32_t per_rank = gridDim.x / (8u * (uint32_t)g.H), rank = local / per_rank;
        t = local - rank * per_rank;
        const int c = (g.H - 1) >> 1, d = (int)((rank + 1u) >> 1);
        ih = (rank & 1u) ? c + d : c - d;
    } else {
        ih = (int)(local % (uint32_t)g.H);
        t = local / (uint32_t)g.H;
    }
    const int st = (int)((t % (uint32_t)tiles8) * 8u + xcd), ct = (int)(t / (uint32_t)tiles8);
    if (st >= ntiles) return;
    const int s0 = st * BM, n0 = ct * BN;
    const int CC = Cout >> 5;

    const int lrow = lane >> 3, lpos = lane & 7;
    const float *asrc[AI], *bsrc[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int row = (i * NW + wave) * 8 + lrow;
        int s = s0 + row;
        s = s < nsamples ? s : nsamples - 1;
        asrc[i] = dy + (int64_t)s * (OH * OW) * Cout + ((lpos ^ ((row >> 1) & 7)) << 2);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int row = (i * NW + wave) * 8 + lrow;
        int c = n0 + row;
        c = c < Cin ? c : Cin - 1;
        bsrc[i] = w + (int64_t)c * Cout + ((lpos ^ ((row >> 1) & 7)) << 2);
    }
    // filter rows of input row ih: oh = ih - kh in [0, OH)
    const int kh_lo = ih - OH + 1 > 0 ? ih - OH + 1 : 0, kh_hi = ih < g.KH - 1 ? ih : g.KH - 1;
    const int total = (kh_hi - kh_lo + 1) * CC;  // stages per output column (>= CC: every input row has a tap)

    f32x16 acc[KWT][TM][TN];
#pragma unroll
    for (int k_ = 0; k_ < KWT; ++k_)
#pragma unroll
        for (int a_ = 0; a_ < TM; ++a_)
#pragma unroll
            for (int b_ = 0; b_ < TN; ++b_)
#pragma unroll
                for (int r_ = 0; r_ < 16; ++r_) acc[k_][a_][b_][r_] = 0.f;
    const uint32_t sstride = (uint32_t)(g.H * Wd * Cin);
    const int srow = s0 + wm * TM * 32 + 4 * (lane >> 5), ccol = n0 + wn * TN * 32 + (lane & 31);
    const uint32_t obase = (uint32_t)srow * sstride + (uint32_t)(ih * Wd) * (uint32_t)Cin + (uint32_t)ccol;
    const int slim = nsamples - srow;
    float pend[TM][TN][16], actv[TM][TN][16];
    const bool full = s0 + BM <= nsamples && n0 + BN <= Cin;
    auto prefetch_act = [&](int iw) {
        if (!in_act) return;
        if ((SF_ROW_ABL & 1) && iw != 0) return;
        const uint32_t pix = (uint32_t)iw * (uint32_t)Cin;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rc = tm * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    if (full) {
                        actv[tm][tn][r] = (in_act + (size_t)(pix + (uint32_t)rc * sstride + (uint32_t)(tn * 32)))[obase];
                    } else {
                        const bool ok = rc < slim && ccol + tn * 32 < Cin;
                        const uint32_t o = obase + pix + (uint32_t)rc * sstride + (uint32_t)(tn * 32);
                        actv[tm][tn][r] = in_act[ok ? o : 0u];
                    }
                }
            }
    };
    auto park_set = [&](f32x16 (&a)[TM][TN]) {  // pend = masked result of a finished pixel; its set restarts at zero
        const int akind = g.relu;
        auto park = [&](auto kc) {
            constexpr int KIND = decltype(kc)::value;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        pend[tm][tn][r] = act_bwd_mul<KIND>(a[tm][tn][r], actv[tm][tn][r], akind);
                        a[tm][tn][r] = 0.f;
                    }
        };
        if (!in_act) park(std::integral_constant<int, 0>{});
        else if (akind == 1) park(std::integral_constant<int, 1>{});
        else park(std::integral_constant<int, -1>{});
    };
    auto store_pixel = [&](int iw) {
        if ((SF_ROW_ABL & 1) && iw != Wd - 1) return;
        const uint32_t pix = (uint32_t)iw * (uint32_t)Cin;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rc = tm * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    if (full) {
                        (din + (size_t)(pix + (uint32_t)rc * sstride + (uint32_t)(tn * 32)))[obase] = pend[tm][tn][r];
                    } else if (rc < slim && ccol + tn * 32 < Cin) {
                        din[obase + pix + (uint32_t)rc * sstride + (uint32_t)(tn * 32)] = pend[tm][tn][r];
                    }
                }
            }
    };
    // DMA instructions of stage (ow, q), piece by piece: pieces 0 .. AI-1 = dY rows, then KWT x BI weight rows
    constexpr int NPIECE = AI + KWT * BI;
    auto issue_piece = [&](int ow, int q, int stage, int piece) {
        const int khi = q / CC, cc = q - khi * CC;  // CC: small wave-uniform divisor (scalar unit)
        const int kh = kh_lo + khi, oh = ih - kh;
        float *sa = lds + stage * STAGE, *sb = sa + BM * 32;
        if (SF_ROW_ABL & 4) return;
        if (piece < AI) {
            const int64_t aoff = (SF_ROW_ABL & 2) ? (int64_t)cc * 32 - (int64_t)s0 * (OH * OW) * Cout
                                                  : (int64_t)(oh * OW + ow) * Cout + cc * 32;
            GLDS16(asrc[piece] + aoff, sa + (piece * NW + wave) * 256);
        } else {
            const int kw = (piece - AI) / BI, i = (piece - AI) - kw * BI;
            const int64_t boff = (int64_t)((kh * KWT + kw) * Cin) * Cout + cc * 32;
            GLDS16(bsrc[i] + boff, sb + kw * BN * 32 + (i * NW + wave) * 256);
        }
    };
#pragma unroll
    for (int pc = 0; pc < NPIECE; ++pc) issue_piece(0, 0, 0, pc);
    int stage = 0, parked = -1, rot = 0;  // rot = ow % KWT: the set of pixel iw = ow
    for (int ow = 0; ow < OW; ++ow) {
        for (int q = 0; q < total; ++q, stage ^= 1) {
            if (!(SF_ROW_ABL & 8)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            const bool more = q + 1 < total || ow + 1 < OW;      // the DMA pipeline runs across the column boundary
            const int now_ = q + 1 < total ? ow : ow + 1, nq = q + 1 < total ? q + 1 : 0;
            const float *sa = lds + stage * STAGE;
            auto head = [&]() {
                if (q == 0 && parked >= 0) store_pixel(parked);
            };
            auto mid = [&](int c) {  // c is a compile-time constant after unrolling: pieces c, c + 4, ... of the next stage
                if (more) {
#pragma unroll
                    for (int pc = 0; pc < NPIECE; ++pc)
                        if (pc % 4 == c) issue_piece(now_, nq, stage ^ 1, pc);
                }
                // activation values of the pixel this column completes: fetched during the column's LAST stage, so that
                // they and the parked result of the previous pixel (stored during the FIRST stage) are never live together
                if (c == 3 && q == total - 1) prefetch_act(ow);
            };
            if (rot == 0) mma_chunk_rows_kw<TM, TN, KWT, 0, BN>(sa, sa + BM * 32, wm * TM * 32, wn * TN * 32, lane, acc, head, mid);
            else if (rot == 1) mma_chunk_rows_kw<TM, TN, KWT, 1, BN>(sa, sa + BM * 32, wm * TM * 32, wn * TN * 32, lane, acc, head, mid);
            else mma_chunk_rows_kw<TM, TN, KWT, (KWT > 2 ? 2 : 0), BN>(sa, sa + BM * 32, wm * TM * 32, wn * TN * 32, lane, acc, head, mid);
        }
        // pixel iw = ow is complete (its last tap column kw = 0 was this one)
        if (rot == 0) park_set(acc[0]);
        else if (rot == 1) park_set(acc[1]);
        else park_set(acc[KWT > 2 ? 2 : 0]);
        parked = ow;
        rot = rot + 1 == KWT ? 0 : rot + 1;
    }
    if (parked >= 0) store_pixel(parked);
    // the last KWT - 1 pixels of the row (iw = OW .. W - 1) are complete as well: their sets hold every tap they have
    for (int iw = OW; iw < Wd; ++iw) {
        prefetch_act(iw);
        if (rot == 0) park_set(acc[0]);
        else if (rot == 1) park_set(acc[1]);
        else park_set(acc[KWT > 2 ? 2 : 0]);
        store_pixel(iw);
        rot = rot + 1 == KWT ? 0 : rot + 1;
    }
}


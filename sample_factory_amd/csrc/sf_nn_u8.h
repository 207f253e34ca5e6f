// Nature-CNN conv1 on raw u8 frames, on the bf16 matrix pipe with EXACT products (included by sf_nn.hip).
//
// A u8 pixel (and pixel - integer mean) has at most 8 significant bits: it IS a bf16 number.  An f32 weight is the
// exact sum of three bf16 numbers (truncation split 8 + 8 + 8 significand bits: w = hi + mid + lo, every residual
// exactly representable).  So   sum_k x_k * w_k  =  sum_k x_k*hi_k + x_k*mid_k + x_k*lo_k   with every product exact
// in f32 (8 x 8 significand bits) and f32 accumulation inside v_mfma_f32_16x16x32_bf16 — no operand is rounded at all
// (the f32 kernel rounds (x - mean) * 1/scale before multiplying), and the bf16 pipe retires 1024 FLOP/clk/SIMD
// against 64 for v_mfma_f32_16x16x4_f32: three passes cost 3/16 of the f32 MFMA time.  The 1/scale factor is applied
// to the accumulated sum in the epilogue (one rounding).
//
// Structure: as k_conv_u8_img (persistent work-groups, strips of R = 4 output rows, bytes prefetched into registers
// during the previous strip's MFMAs), with
//   * SMP = 4 samples per work-group: a strip is 4 x 80 = 320 output rows = 20 fragments of 16 rows, wave w owns the
//     80 rows of sample w (5 fragments) and ALL 32 output channels (2 column tiles), so an A fragment is read from
//     LDS once per strip and k-block;
//   * the strip image in LDS as bf16 [SMP][Cin][20][88] (rows padded to 176 B); k = (c*8 + kh)*8 + kw, a lane's 8
//     k-values of one MFMA are the 8 kw of one (c, kh): 16 contiguous bytes = two ds_read_b64;
//   * the weight fragments (3 terms x 8 k-blocks x 2 column tiles x 4 VGPRs = 192 registers) live in registers for
//     the whole kernel: one wave per SIMD, 512-register budget.
// MFMA layout (v_mfma_f32_16x16x32_bf16): A lane l -> row l & 15, k = 8 * (l >> 4) + j; B lane l -> col l & 15, same
// k; C/D col = l & 15, row = 4 * (l >> 4) + reg.
#ifndef SF_CONV1_WP
#define SF_CONV1_WP 84  // LDS row pitch of the forward strip image, bf16 elements: 168 B = the packed row — a fragment of 16
                       // pixels that wraps into the next output row (4 image rows = 672 B further) continues the bank sequence
                       // (88: rollout-size launch 110 -> 105 us, profiles/r05_s_conv1_trace_quadrow.log)
#endif
#ifndef SF_CONV1_TRACE
#define SF_CONV1_TRACE 0  // experiment builds only: per-phase shader-cycle sums of k_conv1_u8_bf16 (tools/conv1_trace.py)
#endif
#if SF_CONV1_TRACE
__device__ unsigned long long sf_conv1_trace_acc[12];
#define SF_C1T(i)                                 \
    do {                                          \
        const unsigned long long tn_ = clock64(); \
        tacc_[i] += tn_ - tprev_;                 \
        tprev_ = tn_;                             \
    } while (0)
#else
#define SF_C1T(i)
#endif
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// exact 3-way split of an f32 into bf16 bit patterns (truncation: each residual is exact and same-signed)
__device__ __forceinline__ void split3_bf16(float w, uint32_t &hi, uint32_t &mid, uint32_t &lo) {
    const uint32_t b = __float_as_uint(w);
    hi = b >> 16;
    const float r1 = w - __uint_as_float(b & 0xFFFF0000u);
    const uint32_t b1 = __float_as_uint(r1);
    mid = b1 >> 16;
    const float r2 = r1 - __uint_as_float(b1 & 0xFFFF0000u);
    lo = __float_as_uint(r2) >> 16;
}

// NCT = column tiles (of 16 channels) per wave.  NCT = 2: SMP = 4 samples per work-group, wave w = sample w, all 32
// channels, 192 weight registers -> one wave per SIMD: 869 us at n = 32768 (every wave's load / convert / MFMA / store
// phases are serial and nothing else is resident to fill them).  NCT = 1 (the one that is launched): SMP = 2 samples,
// waves = 2 samples x 2 column tiles (an A fragment is read by two waves), 96 weight registers -> two work-groups per
// CU whose phases overlap: 718 us.
// WIDE (N == 32, 16-byte aligned output): whole-line output stores through an LDS staging tile.  The MFMA hands lane
// (c, rg) the 4 pixels 4*rg + r of ONE channel and this wave owns 16 of the 32 channels, so its natural stores are 64-byte
// HALVES of the 128-byte pixel lines, the sibling wave writing the other halves a little earlier or later.  Measured
// (tools/ubench/stream_rw.hip, profiles/r05_u_stream_rw.log): conv1's traffic mix (28 KB read + 51 KB written per sample,
// nothing else) streams at 5.2-5.9 TB/s with whole-line stores and at 3.9 TB/s — 665 us at n = 32768, exactly this
// kernel's time — when two waves write the halves.  (16-byte stores of the halves after an in-register DPP
// transposition cut the store instructions 4x and the epilogue's issue time from 30 % to 18 % of the strip loop, and the
// waves waited that much longer for their bytes instead: -1 %, profiles/r05_t_conv1_wide_stores.log.)
// So: the activated tile goes to LDS as stg[SMP][80 pixels][36 words] (20 ds_write_b32 per lane and strip, immediate
// offsets, conflict-free with the 144-byte pixel pitch), and AFTER the strip's trailing barrier every thread reads 5 x
// 16 bytes back in output order and stores them with global_store_dwordx4: a wave instruction is 1 KB of contiguous
// output.  No extra barrier: the next strip's staging writes come after its leading barrier.  The ReLU sign bits are
// taken there too (4 channels per lane, OR over the 8 lanes of a pixel by DPP): one u32 store per pixel.
template <bool SUB, int NCT, bool WIDE>
__device__ __forceinline__ void conv1_u8_bf16_body(ConvG g, const uint8_t *__restrict__ in, int64_t in_stride,
                                                   const int32_t *__restrict__ index, int64_t offset,
                                                   const float *__restrict__ w, const float *__restrict__ bias,
                                                   float *__restrict__ out, uint32_t *__restrict__ mask_out,
                                                   int nsamples) {
    constexpr int SMP = 2 * NCT, R = 4, TMF = 5, KB = 8;
    constexpr int H = 84, W = 84, WP = SF_CONV1_WP, Cin = 4, KH = 8, S = 4, OH = 20, OW = 20, OHOW = OH * OW;
    constexpr int RS = (R - 1) * S + KH;  // 20 input rows per strip
    constexpr int W4 = W >> 2;            // 4-byte words per input row
    constexpr int WORDS = Cin * RS * W4;  // u32 words per strip and sample
    constexpr int NLD = (WORDS + 255) / 256;
    constexpr int nstrips = OH / R;
    static_assert(R * OW == TMF * 16 && KB * 32 == Cin * KH * 8, "Nature-CNN conv1 geometry");
    extern __shared__ __attribute__((aligned(16))) uint16_t img[];  // [SMP][Cin][RS][WP] bf16, WIDE: + stg[SMP][R*OW][SP] f32
    constexpr int SP = 36;  // staging pitch of a pixel, words: 32 channels + 4 (lanes kg and kg + 1 of a ds_write_b32 group land 16 banks apart)
    float *const stg = reinterpret_cast<float *>(img + SMP * Cin * RS * WP);
    static_assert((SMP * Cin * RS * WP * 2) % 16 == 0, "staging tile must start 16-byte aligned");
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = g.Cout;
    const int wsmp = NCT == 2 ? wave : (wave >> 1), ct0 = NCT == 2 ? 0 : (wave & 1);  // this wave's sample / first column tile
    const int nquads = (nsamples + SMP - 1) / SMP;
    const int my_quads = (int)blockIdx.x < nquads ? (nquads - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int total_units = my_quads * nstrips;  // unit = (local quad, strip)
    if (total_units == 0) return;
    int gofs[NLD], lofs[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        int q = tid + 256 * i;
        const bool ok = q < WORDS;
        q = ok ? q : 0;
        const int x4 = q % W4, t1 = q / W4, row = t1 % RS, c = t1 / RS;
        gofs[i] = (c * H + row) * W + x4 * 4;
        lofs[i] = ok ? (c * RS + row) * WP + x4 * 4 : -1;
    }
    uint32_t pre[SMP][NLD];
    auto load_strip = [&](int unit) {
        const int lq = unit / nstrips, st = unit - lq * nstrips;
        const int s0u = ((int)blockIdx.x + lq * (int)gridDim.x) * SMP;
        const int rowoff = st * R * S * W;
#pragma unroll
        for (int z = 0; z < SMP; ++z) {
            int sg = s0u + z;
            sg = sg < nsamples ? sg : nsamples - 1;
            const uint8_t *sb = in + sample_base(g, index, offset, in_stride, (uint32_t)sg);  // wave-uniform
#pragma unroll
            for (int i = 0; i < NLD; ++i) pre[z][i] = *reinterpret_cast<const uint32_t *>(sb + rowoff + gofs[i]);
        }
    };
    const float sub = g.sub_mean;
    auto store_strip = [&]() {
#pragma unroll
        for (int z = 0; z < SMP; ++z)
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                uint32_t fb[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float b8 = (float)((pre[z][i] >> (8 * j)) & 0xFFu);
                    fb[j] = __float_as_uint(SUB ? b8 - sub : b8);  // an integer of <= 8 bits: its bf16 form is exact
                }
                uint2 v;
                v.x = __builtin_amdgcn_perm(fb[1], fb[0], 0x07060302u);  // {hi16(fb[0]), hi16(fb[1])}
                v.y = __builtin_amdgcn_perm(fb[3], fb[2], 0x07060302u);
                if (lofs[i] >= 0)
                    *reinterpret_cast<uint2 *>(__builtin_assume_aligned(img + z * Cin * RS * WP + lofs[i], 8)) = v;
            }
    };
    load_strip(0);
    // ---- weight fragments: lane (col, kg) holds k = 32*kb + 8*kg + j of column ct*16 + col, split into 3 bf16 terms
    const int col = lane & 15, kg = lane >> 4;
    s16x8 breg[3][KB][NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int cc = (ct0 + ct) * 16 + col, colc = cc < N ? cc : N - 1;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float wv = cc < N ? w[(int64_t)(kb * 32 + kg * 8 + j) * N + colc] : 0.f;
                uint32_t h, m, l;
                split3_bf16(wv, h, m, l);
                breg[0][kb][ct][j] = (short)h;
                breg[1][kb][ct][j] = (short)m;
                breg[2][kb][ct][j] = (short)l;
            }
    }
    // (Computing the product transposed — weights as the MFMA's A operand — gives every lane 4 consecutive channels of
    // one pixel, i.e. 16-byte stores, but 64 separate 16-byte requests per instruction instead of 4 runs of 64 bytes:
    // measured 841 us against 718 us at n = 32768.  Not used.)
    float bv[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) bv[ct] = (bias && (ct0 + ct) * 16 + col < N) ? bias[(ct0 + ct) * 16 + col] : 0.f;
    // fragment rows of this wave: sample wsmp, rows p = 16*t + (lane & 15) of the strip's 80
    int origin[TMF];
#pragma unroll
    for (int t = 0; t < TMF; ++t) {
        const int p = t * 16 + (lane & 15), ohl = p / OW, ow = p - ohl * OW;
        origin[t] = (wsmp * Cin * RS + ohl * S + kg) * WP + ow * S;  // kh = 4*(kb & 1) + kg
    }
    const float scl = g.inv_scale;
#if SF_CONV1_TRACE
    unsigned long long tacc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev_ = clock64();
    const unsigned long long tc0_ = tprev_, tw0_ = wall_clock64();
#endif
    for (int unit = 0; unit < total_units; ++unit) {
        const int lq = unit / nstrips, st = unit - lq * nstrips;
        const int s0 = ((int)blockIdx.x + lq * (int)gridDim.x) * SMP;
#if SF_CONV1_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SF_C1T(0);  // waiting for the prefetched bytes (and this unit's output stores)
#endif
        store_strip();
#if SF_CONV1_TRACE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        SF_C1T(1);  // convert + LDS writes
#endif
        __syncthreads();
        SF_C1T(2);  // barrier 1
        if (unit + 1 < total_units) load_strip(unit + 1);  // lands during the MFMA phase
        f32x4 acc[TMF][NCT];
#pragma unroll
        for (int t = 0; t < TMF; ++t)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) acc[t][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
        s16x8 a[2][TMF];
        auto fetch = [&](int kb) {
            const int tap = ((kb >> 1) * RS + (kb & 1) * 4) * WP;  // (c, kh0) of this k-block
#pragma unroll
            for (int t = 0; t < TMF; ++t) {
                const uint16_t *p = img + origin[t] + tap;
                const s16x4 lo = *reinterpret_cast<const s16x4 *>(__builtin_assume_aligned(p, 8));
                const s16x4 hi = *reinterpret_cast<const s16x4 *>(__builtin_assume_aligned(p + 4, 8));
                a[kb & 1][t] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
        };
        fetch(0);
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            if (kb + 1 < KB) fetch(kb + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int term = 2; term >= 0; --term)  // small terms first
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int t = 0; t < TMF; ++t)
                        acc[t][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, a[kb & 1][t]), __builtin_bit_cast(bf16x8, breg[term][kb][ct]),
                            acc[t][ct], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#if SF_CONV1_TRACE
        {
            float sink = 0.f;
#pragma unroll
            for (int t = 0; t < TMF; ++t) sink += acc[t][0][0];
            asm volatile("" ::"v"(sink));  // results landed
        }
        SF_C1T(3);  // load issue + MFMA phase
#endif
        auto epilogue = [&](auto kc) {
            constexpr int KIND = decltype(kc)::value;
            const bool sok = s0 + wsmp < nsamples;
            float *ob = out + ((int64_t)(s0 + wsmp) * OHOW + st * (R * OW)) * N;
            // ReLU sign bits for the backward pass (mask_out != NULL, N == 32): one u32 per output pixel, bit c = channel c
            // is positive.  This wave owns 16 channels = one u16 half of each word: a wave ballot of (v > 0) is 4 pixels
            // (kg) x 16 channels (col); lane (kg, col 0) stores its pixel's half.
            uint16_t *mb = mask_out ? reinterpret_cast<uint16_t *>(mask_out) +
                                          ((int64_t)(s0 + wsmp) * OHOW + st * (R * OW)) * 2 + ct0 : nullptr;
            if constexpr (WIDE) {  // stage: lane (col, kg), register r -> pixel t*16 + 4*kg + r, channel ct*16 + col of this wave's sample
                float *sl = stg + (wsmp * (R * OW) + 4 * kg) * SP + ct0 * 16 + col;
#pragma unroll
                for (int t = 0; t < TMF; ++t)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            sl[(t * 16 + r) * SP + ct * 16] = act_fwd_c<KIND>(acc[t][ct][r] * scl + bv[ct], g.relu);
                return;
            }
#pragma unroll
            for (int t = 0; t < TMF; ++t)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const int cc = (ct0 + ct) * 16 + col;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = act_fwd_c<KIND>(acc[t][ct][r] * scl + bv[ct], g.relu);
                        if (sok && cc < N) ob[(int64_t)(t * 16 + 4 * kg + r) * N + cc] = v;
                        if (KIND == 1 && NCT == 1 && mb) {  // (wave-uniform)
                            const unsigned long long bits = __ballot(v > 0.f);
                            if (col == 0 && sok) mb[(t * 16 + 4 * kg + r) * 2] = (uint16_t)(bits >> (16 * kg));
                        }
                    }
                }
        };
        if (g.relu == 1) epilogue(std::integral_constant<int, 1>{});
        else epilogue(std::integral_constant<int, -1>{});
        SF_C1T(4);  // epilogue issue
        __syncthreads();  // everybody is done reading this strip before it is overwritten
        SF_C1T(5);  // barrier 2
        if constexpr (WIDE) {  // the staged tile, in output order: thread -> 16-byte word q of [SMP][80 pixels][8 words]
            constexpr int QS = R * OW * 8;  // 16-byte words per sample and strip
#pragma unroll
            for (int k = 0; k < (SMP * QS + 255) / 256; ++k) {
                const int q = tid + 256 * k, z = q >= QS ? 1 : 0, qq = q - z * QS, px = qq >> 3, ch = qq & 7;
                if (SMP * QS % 256 != 0 && q >= SMP * QS) break;
                const float4 v = *reinterpret_cast<const float4 *>(stg + (z * (R * OW) + px) * SP + ch * 4);
                const bool sok = s0 + z < nsamples;
                const int64_t pix0 = (int64_t)(s0 + z) * OHOW + st * (R * OW);
                if (sok) *reinterpret_cast<float4 *>(out + pix0 * 32 + qq * 4) = v;
                if (g.relu == 1 && mask_out) {  // (uniform) sign bits of the pixel: 4 per lane, OR over its 8 lanes
                    int nib = (v.x > 0.f ? 1 : 0) | (v.y > 0.f ? 2 : 0) | (v.z > 0.f ? 4 : 0) | (v.w > 0.f ? 8 : 0);
                    nib <<= 4 * ch;
                    nib |= __builtin_amdgcn_update_dpp(0, nib, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
                    nib |= __builtin_amdgcn_update_dpp(0, nib, 0x4E /* quad_perm [2,3,0,1] */, 0xF, 0xF, true);
                    nib |= __builtin_amdgcn_update_dpp(0, nib, 0x141 /* row_half_mirror */, 0xF, 0xF, true);
                    if (ch == 0 && sok) mask_out[pix0 + px] = (uint32_t)nib;
                }
            }
            SF_C1T(6);  // staged tile -> global
        }
    }
#if SF_CONV1_TRACE
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 7; ++i) atomicAdd(&sf_conv1_trace_acc[i], tacc_[i]);
        atomicAdd(&sf_conv1_trace_acc[7], (unsigned long long)total_units);
        atomicAdd(&sf_conv1_trace_acc[8], clock64() - tc0_);      // the strip loop in s_memtime ticks ...
        atomicAdd(&sf_conv1_trace_acc[9], wall_clock64() - tw0_);  // ... and in 100 MHz ticks
        atomicAdd(&sf_conv1_trace_acc[10], 1ull);
    }
#endif
}

template <bool SUB>
__global__ __launch_bounds__(256, 2)
void k_conv1_u8_bf16(ConvG g, const uint8_t *__restrict__ in, int64_t in_stride, const int32_t *__restrict__ index,
                     int64_t offset, const float *__restrict__ w, const float *__restrict__ bias,
                     float *__restrict__ out, uint32_t *__restrict__ mask_out, int nsamples) {
    conv1_u8_bf16_body<SUB, 1, false>(g, in, in_stride, index, offset, w, bias, out, mask_out, nsamples);
}
template <bool SUB>
__global__ __launch_bounds__(256, 2)
void k_conv1_u8_bf16_w(ConvG g, const uint8_t *__restrict__ in, int64_t in_stride, const int32_t *__restrict__ index,
                       int64_t offset, const float *__restrict__ w, const float *__restrict__ bias,
                       float *__restrict__ out, uint32_t *__restrict__ mask_out, int nsamples) {
    conv1_u8_bf16_body<SUB, 1, true>(g, in, in_stride, index, offset, w, bias, out, mask_out, nsamples);
}

// ============================================================================================== WEIGHT GRADIENT, raw u8 frames
// dW[(c, kh, kw)][n] = sum over output pixels m of x[m, (c, kh, kw)] * dY[m, n] with the same exact-product arithmetic:
// x (pixel - integer mean) is a bf16 number, dY is split exactly into three bf16 terms, products are exact, accumulation
// is f32 inside v_mfma_f32_16x16x32_bf16; 1/scale multiplies the finished sums.  The reduction index of the MFMA is the
// output pixel: a lane's 8 reduction elements are two QUADS of 4 consecutive output pixels (ow0 = 0, 4, .., 16: five per
// output row, nothing ragged), 8 quads = 32 pixels per instruction.
//   * x along ow at fixed (c, kh, kw) is a stride-4 walk through the input row, so the strip image is stored
//     DE-INTERLEAVED in LDS: img[smp][c][row][phase = x & 3][q = x >> 2] (bf16, line pitch 36): the 4 pixels of a quad for
//     kw = phase are 4 contiguous elements at q = ow0 (one ds_read_b64), and for kw = 4 + phase the same line one
//     element further (q = ow0 + 1): the 5th element is fetched with a ds_read_b32 and the fragment is formed with two
//     v_alignbit — tiles are arranged so that the shift is uniform per tile (tile = (kh half, kw half), row i -> kh = 4*khalf
//     + i/4, kw = 4*kwhalf + i%4).
//   * dY is stored TRANSPOSED and split: dyT[term][n][pixel] (pitch 168), written as pixel PAIRS (one ds_write_b32); a
//     lane's two quads are 8 consecutive pixel indices there: one ds_read_b128 per fragment.
//   * wave w owns input channel c = w: 4 k-tiles x 2 n-tiles = 8 accumulator tiles, and walks ALL pixels of the strip
//     (5 MFMA k-steps per strip of 2 samples x 80 pixels): no cross-wave reduction, one partial [256][32] per work-group
//     (summed by k_reduce_partials in fixed order).  The bias gradient (column sums of dY) is taken from the f32 values
//     on their way into LDS.
template <bool SUB>
__global__ __launch_bounds__(256, 2)
void k_conv1_wgrad_bf16(ConvG g, const uint8_t *__restrict__ in, int64_t in_stride, const int32_t *__restrict__ index,
                        int64_t offset, const float *__restrict__ dy, const uint32_t *__restrict__ dmask,
                        float *__restrict__ partial, float *__restrict__ partial_b, int nsamples, int npairs) {
    // dmask != NULL: dy is the gradient wrt this layer's ReLU OUTPUT, not yet masked; dmask[sample*400 + pixel] holds the
    // sign bits the forward kernel recorded (bit c: channel c was positive) and the mask is applied on the way into LDS —
    // the producing data-gradient kernel then never reads the 1.68 GB activation (n = 32768) to mask its output.
    constexpr int SMP = 2, R = 4, H = 84, W = 84, Cin = 4, S = 4, OH = 20, OW = 20, OHOW = OH * OW, N = 32, K = 256;
    constexpr int RS = (R - 1) * S + 8, W4 = W >> 2;
    constexpr int WP2 = (W4 + 1) / 2, PAIRS = Cin * RS * WP2, NLD = (PAIRS + 255) / 256;  // word pairs (8 pixels) per thread
    constexpr int QP = 36;                 // elements per (c, row, phase) line: 21 used; 18 dwords -> conflict-free b64 reads
    constexpr int PIX = SMP * R * OW;      // 160 output pixels per strip
    constexpr int DP = 168;                // dyT pitch (elements): 336 B, 16-byte aligned lines
    constexpr int IMG_E = SMP * Cin * RS * 4 * QP;
    constexpr int NDY = (PIX / 2 * (N / 4) + 255) / 256;  // (pixel pair, channel quad) items per thread: 3
    constexpr int nstrips = OH / R;
    extern __shared__ __attribute__((aligned(16))) uint16_t smem16[];
    uint16_t *img = smem16;            // [SMP][Cin][RS][4][QP]
    uint16_t *dyT = smem16 + IMG_E;    // [3][N][DP]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kg = lane >> 4;
    // ---- image word PAIRS of this thread: pair u -> (c, row, xp): words 2xp, 2xp+1 = pixels of phases 0..3 at q = 2xp,
    // 2xp+1 (the 11th pair of a row has no second word: the first is used twice, q = 21 is never read)
    int gofs[NLD], gofs2[NLD], lofs[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        int u = tid + 256 * i;
        const bool ok = u < PAIRS;
        u = ok ? u : 0;
        const int xp = u % WP2, t1 = u / WP2, row = t1 % RS, c = t1 / RS;
        gofs[i] = (c * H + row) * W + xp * 8;
        gofs2[i] = gofs[i] + (2 * xp + 1 < W4 ? 4 : 0);
        lofs[i] = ok ? ((c * RS + row) * 4) * QP + 2 * xp : -1;
    }
    // ---- dY items of this thread: item e -> (pixel pair pp, channel quad n4); n4 = tid & 7 for every item
    const int n4 = tid & 7;
    // ---- fragment addressing, fixed per lane.  k-step s, half h: quad qd = 8*s + 2*kg + h -> pixel p0 = 4*qd
    // A: lines of (khalf): kh = 4*khalf + (i16 >> 2), phase = i16 & 3
    const int a_lane = ((wave * RS + (i16 >> 2)) * 4 + (i16 & 3)) * QP;  // + smp*Cin*RS*4*QP + (ohl*4 + 4*khalf)*4*QP + ow0
    f32x4 acc[2][2][2];
#pragma unroll
    for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
            for (int c_ = 0; c_ < 2; ++c_) acc[a_][b_][c_] = f32x4{0.f, 0.f, 0.f, 0.f};
    float colacc[4] = {0.f, 0.f, 0.f, 0.f};
    const float sub = g.sub_mean;
    // Operands are prefetched TWO strips ahead into two register sets (a strip's MFMA phase, ~1 us, is shorter than a
    // loaded HBM round trip): unit u = (local pair, strip) uses set u & 1.
    struct Regs { uint32_t pre[SMP][NLD][2]; f32x4 dpre[NDY][2]; uint2 mpre[NDY]; };
    const int my_pairs = (int)blockIdx.x < npairs ? (npairs - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int total_units = my_pairs * nstrips;
    auto load_strip = [&](int unit, Regs &rg) {
        const int lp = unit / nstrips, st = unit - lp * nstrips;
        const int pair = (int)blockIdx.x + lp * (int)gridDim.x;
        const int rowoff = st * R * S * W;
        bool sok[SMP];
        int sidx[SMP];
#pragma unroll
        for (int z = 0; z < SMP; ++z) {
            int sg = pair * SMP + z;
            sok[z] = sg < nsamples;
            sg = sok[z] ? sg : nsamples - 1;
            sidx[z] = sg;
            const uint8_t *sb = in + sample_base(g, index, offset, in_stride, (uint32_t)sg);  // wave-uniform
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                rg.pre[z][i][0] = *reinterpret_cast<const uint32_t *>(sb + rowoff + gofs[i]);
                rg.pre[z][i][1] = *reinterpret_cast<const uint32_t *>(sb + rowoff + gofs2[i]);
            }
        }
#pragma unroll
        for (int it = 0; it < NDY; ++it) {
            const int e = it * 256 + tid, pp = e >> 3;           // pixel pair 0..79 (PIX/2), 40 per sample
            const bool ok = pp < PIX / 2;
            const int z = (ok && pp >= R * OW / 2) ? 1 : 0, pl = 2 * (pp - z * (R * OW / 2));
            const bool live = ok && (z == 0 ? sok[0] : sok[1]);
            const float *src = dy + ((int64_t)(z == 0 ? sidx[0] : sidx[1]) * OHOW + st * (R * OW) + pl) * N + n4 * 4;
            rg.dpre[it][0] = live ? *reinterpret_cast<const f32x4 *>(src) : f32x4{0.f, 0.f, 0.f, 0.f};
            rg.dpre[it][1] = live ? *reinterpret_cast<const f32x4 *>(src + N) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (dmask)  // the two pixels' sign words (pl is even: one aligned 8-byte load)
                rg.mpre[it] = live ? *reinterpret_cast<const uint2 *>(dmask + (int64_t)(z == 0 ? sidx[0] : sidx[1]) * OHOW +
                                                                      st * (R * OW) + pl)
                                   : uint2{0u, 0u};
        }
    };
    auto store_strip = [&](const Regs &rg) {
#pragma unroll
        for (int z = 0; z < SMP; ++z)
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                if (lofs[i] < 0) continue;
                uint32_t *dst = reinterpret_cast<uint32_t *>(img + z * Cin * RS * 4 * QP + lofs[i]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a8 = (float)((rg.pre[z][i][0] >> (8 * j)) & 0xFFu), b8 = (float)((rg.pre[z][i][1] >> (8 * j)) & 0xFFu);
                    // integers of <= 8 significant bits: the upper halves of the f32 patterns ARE their bf16 forms
                    dst[j * QP / 2] = __builtin_amdgcn_perm(__float_as_uint(SUB ? b8 - sub : b8),
                                                            __float_as_uint(SUB ? a8 - sub : a8), 0x07060302u);
                }
            }
#pragma unroll
        for (int it = 0; it < NDY; ++it) {
            const int e = it * 256 + tid, pp = e >> 3;
            if (pp >= PIX / 2) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v0 = rg.dpre[it][0][j], v1 = rg.dpre[it][1][j];
                if (dmask) {  // (uniform) ReLU derivative from the recorded sign bit of channel 4*n4 + j
                    v0 = ((rg.mpre[it].x >> (4 * n4 + j)) & 1u) ? v0 : 0.f;
                    v1 = ((rg.mpre[it].y >> (4 * n4 + j)) & 1u) ? v1 : 0.f;
                }
                colacc[j] += v0 + v1;
                // exact 3-way split of both values (see split3_bf16), the pair packed with one v_perm per term
                const uint32_t b0 = __float_as_uint(v0), b1 = __float_as_uint(v1);
                const float r0 = v0 - __uint_as_float(b0 & 0xFFFF0000u), r1 = v1 - __uint_as_float(b1 & 0xFFFF0000u);
                const uint32_t c0 = __float_as_uint(r0), c1 = __float_as_uint(r1);
                const float s0 = r0 - __uint_as_float(c0 & 0xFFFF0000u), s1 = r1 - __uint_as_float(c1 & 0xFFFF0000u);
                uint32_t *dst = reinterpret_cast<uint32_t *>(dyT + (n4 * 4 + j) * DP + 2 * pp);
                dst[0] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
                dst[N * DP / 2] = __builtin_amdgcn_perm(c1, c0, 0x07060302u);
                dst[2 * N * DP / 2] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
            }
        }
    };
    auto mfma_phase = [&]() {
#pragma unroll
    for (int s = 0; s < PIX / 32; ++s) {
        // this lane's two quads of the k-step
        s16x8 afr[2][2], bfr[3][2];
        uint32_t aw[2][2][3];  // [khalf][h][d0, d1, d2]
        // B: the lane's two quads are 8 CONSECUTIVE pixel indices of dyT (it is indexed by the linear strip pixel)
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                bfr[t][nt] = *reinterpret_cast<const s16x8 *>(
                    __builtin_assume_aligned(dyT + (t * N + nt * 16 + i16) * DP + 8 * (4 * s + kg), 16));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int p0 = 4 * (8 * s + 2 * kg + h);            // 0..156
            const int smp = p0 / (R * OW), rem = p0 - smp * (R * OW), ohl = rem / OW, ow0 = rem - ohl * OW;
            const uint16_t *ab = img + smp * Cin * RS * 4 * QP + a_lane + (ohl * S) * 4 * QP + ow0;
#pragma unroll
            for (int kh2 = 0; kh2 < 2; ++kh2) {
                const uint16_t *ap = ab + kh2 * 4 * 4 * QP;
                const uint2 d01 = *reinterpret_cast<const uint2 *>(__builtin_assume_aligned(ap, 8));
                aw[kh2][h][0] = d01.x;
                aw[kh2][h][1] = d01.y;
                aw[kh2][h][2] = *reinterpret_cast<const uint32_t *>(__builtin_assume_aligned(ap + 4, 4));
            }
        }
#pragma unroll
        for (int kh2 = 0; kh2 < 2; ++kh2) {
            uint32_t f0[4], f1[4];  // kw half 0: elements 0..3 of each quad; kw half 1: elements 1..4
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f0[2 * h] = aw[kh2][h][0];
                f0[2 * h + 1] = aw[kh2][h][1];
                f1[2 * h] = __builtin_amdgcn_alignbit(aw[kh2][h][1], aw[kh2][h][0], 16);
                f1[2 * h + 1] = __builtin_amdgcn_alignbit(aw[kh2][h][2], aw[kh2][h][1], 16);
            }
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            afr[kh2][0] = __builtin_bit_cast(s16x8, (u32x4){f0[0], f0[1], f0[2], f0[3]});
            afr[kh2][1] = __builtin_bit_cast(s16x8, (u32x4){f1[0], f1[1], f1[2], f1[3]});
        }
#pragma unroll
        for (int t = 2; t >= 0; --t)
#pragma unroll
            for (int kh2 = 0; kh2 < 2; ++kh2)
#pragma unroll
                for (int kw2 = 0; kw2 < 2; ++kw2)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
                        acc[kh2][kw2][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                            __builtin_bit_cast(bf16x8, afr[kh2][kw2]), __builtin_bit_cast(bf16x8, bfr[t][nt]),
                            acc[kh2][kw2][nt], 0, 0, 0);
    }
    };
    Regs ra, rb;
    if (total_units > 0) load_strip(0, ra);
    if (total_units > 1) load_strip(1, rb);
    auto step = [&](int unit, Regs &rg) {
        store_strip(rg);
        __syncthreads();
        if (unit + 2 < total_units) load_strip(unit + 2, rg);
        mfma_phase();
        __syncthreads();  // img and dyT are free again
    };
    for (int unit = 0; unit < total_units; unit += 2) {
        step(unit, ra);
        if (unit + 1 < total_units) step(unit + 1, rb);
    }
    // ---- one partial per work-group: wave c writes its 64 weight rows.  C layout: col = lane & 15 -> n, row = 4*kg + r -> i
    const float scl = g.inv_scale;
    float *dst = partial + (int64_t)blockIdx.x * K * N;
#pragma unroll
    for (int kh2 = 0; kh2 < 2; ++kh2)
#pragma unroll
        for (int kw2 = 0; kw2 < 2; ++kw2)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = wave * 64 + (4 * kh2 + kg) * 8 + 4 * kw2 + r;  // i = 4*kg + r: kh = 4*khalf + i/4, kw = 4*kwhalf + i%4
                    dst[k * N + nt * 16 + i16] = acc[kh2][kw2][nt][r] * scl;
                }
    if (partial_b) {
        __syncthreads();
        float *red = reinterpret_cast<float *>(smem16);  // [256][4]
#pragma unroll
        for (int j = 0; j < 4; ++j) red[tid * 4 + j] = colacc[j];
        __syncthreads();
        if (tid < N) {
            float sum = 0.f;
#pragma unroll 4
            for (int part = 0; part < 32; ++part) sum += red[(part * 8 + (tid >> 2)) * 4 + (tid & 3)];
            partial_b[(int64_t)blockIdx.x * N + tid] = sum;
        }
    }
}

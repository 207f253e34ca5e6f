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
template <bool SUB, int NCT>
__device__ __forceinline__ void conv1_u8_bf16_body(ConvG g, const uint8_t *__restrict__ in, int64_t in_stride,
                                                   const int32_t *__restrict__ index, int64_t offset,
                                                   const float *__restrict__ w, const float *__restrict__ bias,
                                                   float *__restrict__ out, int nsamples) {
    constexpr int SMP = 2 * NCT, R = 4, TMF = 5, KB = 8;
    constexpr int H = 84, W = 84, WP = 88, Cin = 4, KH = 8, S = 4, OH = 20, OW = 20, OHOW = OH * OW;
    constexpr int RS = (R - 1) * S + KH;  // 20 input rows per strip
    constexpr int W4 = W >> 2;            // 4-byte words per input row
    constexpr int WORDS = Cin * RS * W4;  // u32 words per strip and sample
    constexpr int NLD = (WORDS + 255) / 256;
    constexpr int nstrips = OH / R;
    static_assert(R * OW == TMF * 16 && KB * 32 == Cin * KH * 8, "Nature-CNN conv1 geometry");
    extern __shared__ __attribute__((aligned(16))) uint16_t img[];  // [SMP][Cin][RS][WP] bf16
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
    for (int unit = 0; unit < total_units; ++unit) {
        const int lq = unit / nstrips, st = unit - lq * nstrips;
        const int s0 = ((int)blockIdx.x + lq * (int)gridDim.x) * SMP;
        store_strip();
        __syncthreads();
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
        auto epilogue = [&](auto kc) {
            constexpr int KIND = decltype(kc)::value;
            const bool sok = s0 + wsmp < nsamples;
            float *ob = out + ((int64_t)(s0 + wsmp) * OHOW + st * (R * OW)) * N;
#pragma unroll
            for (int t = 0; t < TMF; ++t)
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) {
                    const int cc = (ct0 + ct) * 16 + col;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = act_fwd_c<KIND>(acc[t][ct][r] * scl + bv[ct], g.relu);
                        if (sok && cc < N) ob[(int64_t)(t * 16 + 4 * kg + r) * N + cc] = v;
                    }
                }
        };
        if (g.relu == 1) epilogue(std::integral_constant<int, 1>{});
        else epilogue(std::integral_constant<int, -1>{});
        __syncthreads();  // everybody is done reading this strip before it is overwritten
    }
}

template <bool SUB>
__global__ __launch_bounds__(256, 2)
void k_conv1_u8_bf16(ConvG g, const uint8_t *__restrict__ in, int64_t in_stride, const int32_t *__restrict__ index,
                     int64_t offset, const float *__restrict__ w, const float *__restrict__ bias,
                     float *__restrict__ out, int nsamples) {
    conv1_u8_bf16_body<SUB, 1>(g, in, in_stride, index, offset, w, bias, out, nsamples);
}

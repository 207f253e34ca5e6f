// sf_nn_glds.h — LDS-DMA ("glds") variants of the implicit-GEMM family for the dense hot layers (included by sf_nn.hip).
//
// The register-staged kernels in sf_nn.hip spend their non-MFMA time on the vector-memory path: global_load -> VGPR
// -> (convert/select) -> ds_write_b32 x4 -> ds_read_b32 per fragment word.  For operands that are already f32 and whose
// reduction axis is contiguous in memory gfx950 can do better:
//   * global_load_lds_dwordx4: the 16 bytes of every lane go straight into LDS (no staging VGPRs, no ds_write pass);
//     the destination is lane-linear (wave-uniform base + lane*16), the SOURCE address is per lane.
//   * both operands are laid out "row = free index, 32 reduction words contiguous" (128-byte LDS rows), so one
//     ds_read_b128 fetches the operand words of FOUR MFMAs.  The MFMA only needs A and B to agree on which reduction
//     index sits in which half-wave slot, so the reduction order inside a 32-chunk is permuted: for group c = 0..3 the
//     lanes of half h = lane>>5 hold words k = 8c + 4h + j, j = 0..3, and MFMA j consumes word j of both quads.
//   * bank conflicts: 16 lanes x 16 B are served per cycle, rows are 128 B, so a plain row-major image would put all
//     rows of equal parity on the same 16-byte slots (8-way).  Chunk position p of row r stores reduction chunk
//     p ^ ((r >> 1) & 7); since the DMA destination cannot be permuted the permutation is applied to the per-lane
//     SOURCE address and (the same involution) to the ds_read address (cdna_hip_programming.md rule 21).
//   * LDS is double buffered: the DMA of chunk t+1 is issued right after the barrier that publishes chunk t and has
//     the whole MFMA phase (32+ MFMAs = 2k+ cycles per wave) to land; one barrier per chunk.
//
// Preconditions (checked by the launchers, everything else stays on the register-staged kernels): f32 NHWC input,
// Cin % 32 == 0 (a 32-chunk never straddles a filter tap), dense samples (no index gather), 16-byte aligned bases.
#pragma once

// SF_GLDS_ABLATE (compile-time, tools/build_variant.sh; TIMING EXPERIMENTS ONLY — results are wrong with any bit set):
// bit 0: k_fwd_glds issues no DMA inside its k-loop, bit 1: no wait / barrier per chunk, bit 2: no epilogue stores,
// bit 3: one k-chunk per tile (prologue + epilogue only); k_dgrad_quadrow_z: bit 4 no DMA, bit 5 no output stores,
// bit 6 no wait / barrier per chunk, bit 7 no MFMAs
#ifndef SF_GLDS_ABLATE
#define SF_GLDS_ABLATE 0
#endif
#define GLDS16(gsrc, ldst)                                                                       \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc),    \
                                     (__attribute__((address_space(3))) void *)(ldst), 16, 0, 0)

// Work-group barrier WITHOUT the memory fence of __syncthreads(): the fence makes hipcc drain every outstanding DMA
// (s_waitcnt vmcnt(0)) in front of the barrier, which is exactly what a multi-stage pipeline must not do.  Ordering is
// provided by the explicit counted s_waitcnt in front of it (DMA data) and by the fact that every ds_read result has
// been consumed by an MFMA before the wave gets here.
#define BARRIER_NOFENCE()                  \
    do {                                   \
        asm volatile("" ::: "memory");     \
        __builtin_amdgcn_s_barrier();      \
        asm volatile("" ::: "memory");     \
    } while (0)

// One 32-deep chunk of MFMAs out of the swizzled row-major LDS images.  As: BM rows x 32 words, Bs: BN rows x 32 words.
template <int TM, int TN>
__device__ __forceinline__ void mma_chunk_rows(const float *__restrict__ As, const float *__restrict__ Bs, int arow0,
                                               int brow0, int lane, f32x16 (&acc)[TM][TN]) {
    const int r = lane & 31, h = lane >> 5, sw = (r >> 1) & 7;
    const float *ap = As + (arow0 + r) * 32, *bp = Bs + (brow0 + r) * 32;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int pos = (((2 * c + h) ^ sw) << 2);
        float4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4 *>(ap + i * 32 * 32 + pos);
#pragma unroll
        for (int i = 0; i < TN; ++i) b[i] = *reinterpret_cast<const float4 *>(bp + i * 32 * 32 + pos);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                        j == 0 ? a[tm].x : j == 1 ? a[tm].y : j == 2 ? a[tm].z : a[tm].w,
                        j == 0 ? b[tn].x : j == 1 ? b[tn].y : j == 2 ? b[tn].z : b[tn].w, acc[tm][tn], 0, 0, 0);
    }
}

// Same chunk, with the caller's DMA instructions for the NEXT chunk spread between the MFMA groups: mid(c) is called in
// the middle of group c's 4*TM*TN MFMAs (after 2 of its 4 k-steps), the fragment reads of group c+1 follow it.  A DMA
// instruction keeps its wave's issue port for ~64 cycles; issued in a burst in front of the MFMA phase (6 per wave)
// that is ~400 cycles with none of this wave's MFMAs in flight, issued one or two at a time between MFMAs it sits in
// the shadow of the previous MFMA's 64-cycle pass through the matrix pipe.
template <int TM, int TN, typename F>
__device__ __forceinline__ void mma_chunk_rows_mid(const float *__restrict__ As, const float *__restrict__ Bs, int arow0,
                                                   int brow0, int lane, f32x16 (&acc)[TM][TN], F &&mid) {
    const int r = lane & 31, h = lane >> 5, sw = (r >> 1) & 7;
    const float *ap = As + (arow0 + r) * 32, *bp = Bs + (brow0 + r) * 32;
    float4 a[2][TM], b[2][TN];
    auto fetch = [&](int c) {
        const int pos = (((2 * c + h) ^ sw) << 2);
#pragma unroll
        for (int i = 0; i < TM; ++i) a[c & 1][i] = *reinterpret_cast<const float4 *>(ap + i * 32 * 32 + pos);
#pragma unroll
        for (int i = 0; i < TN; ++i) b[c & 1][i] = *reinterpret_cast<const float4 *>(bp + i * 32 * 32 + pos);
    };
    auto mfmas = [&](int c, int j) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                    j == 0 ? a[c & 1][tm].x : j == 1 ? a[c & 1][tm].y : j == 2 ? a[c & 1][tm].z : a[c & 1][tm].w,
                    j == 0 ? b[c & 1][tn].x : j == 1 ? b[c & 1][tn].y : j == 2 ? b[c & 1][tn].z : b[c & 1][tn].w,
                    acc[tm][tn], 0, 0, 0);
    };
    fetch(0);
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
}

// LDS-DMA with the address split the way the hardware wants it: 64-bit UNIFORM base in SGPRs + 32-bit per-lane byte
// offset in ONE VGPR (hipcc's builtin only emits the "off" form: a 64-bit per-lane address, i.e. one v_lshl_add_u64 per
// instruction and chunk to add the chunk's uniform offset).  M0 (LDS destination base) is set inside the same statement.
__device__ __forceinline__ void glds16_s(const void *ubase, uint32_t voff, uint32_t lds_byte_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(lds_byte_addr), "v"(voff), "s"(ubase) : "memory", "m0");
}

// ... with the LDS destination as (SGPR base + compile-time immediate): hipcc otherwise keeps one loop-invariant destination per
// DMA instruction and, short of SGPRs, parks them in VGPR lanes (k_dgrad_pix_z: 6 v_readlane + 6 s_add per chunk between the
// barrier and the DMA burst)
template <int I, int N, typename F>
__device__ __forceinline__ void sf_static_for(F &&f) {  // f(integral_constant<int, I>) for I = I .. N-1, unrolled at compile time
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sf_static_for<I + 1, N>(f);
    }
}
template <int IMM>
__device__ __forceinline__ void glds16_si(const void *ubase, uint32_t voff, uint32_t lds_base) {
    asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(lds_base), "v"(voff), "s"(ubase), "n"(IMM) : "memory", "m0", "scc");
}

// mma_chunk_rows_mid with the fragment addresses as per-lane LDS POINTERS computed once per kernel (ap[c], bp[c]: group c of
// stage 0) and the stage as a compile-time float offset: every ds_read_b128 is "VGPR + immediate", no address VALU at all.
template <int TM, int TN, int OFF, typename F>
__device__ __forceinline__ void mma_chunk_ptrs_mid(const float *const (&ap)[4], const float *const (&bp)[4],
                                                   f32x16 (&acc)[TM][TN], F &&mid) {
    float4 a[2][TM], b[2][TN];
    auto fetch = [&](int c) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[c & 1][i] = *reinterpret_cast<const float4 *>(ap[c] + OFF + i * 32 * 32);
#pragma unroll
        for (int i = 0; i < TN; ++i) b[c & 1][i] = *reinterpret_cast<const float4 *>(bp[c] + OFF + i * 32 * 32);
    };
    auto mfmas = [&](int c, int j) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                    j == 0 ? a[c & 1][tm].x : j == 1 ? a[c & 1][tm].y : j == 2 ? a[c & 1][tm].z : a[c & 1][tm].w,
                    j == 0 ? b[c & 1][tn].x : j == 1 ? b[c & 1][tn].y : j == 2 ? b[c & 1][tn].z : b[c & 1][tn].w,
                    acc[tm][tn], 0, 0, 0);
    };
    fetch(0);
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
}

// mma_chunk_rows (fragments of a group fetched together, no scheduling fences: the form the 64x64 wave tiles want) with
// pointer + compile-time-offset addressing
// ZEROC: the chunk STARTS an accumulation — the first MFMA of every accumulator takes the constant 0 as its C operand
// (an inline constant of the instruction) instead of a register tile that 16 v_mov per tile would have to clear first
template <int TM, int TN, int OFF, bool ZEROC = false>
__device__ __forceinline__ void mma_chunk_ptrs(const float *const (&ap)[4], const float *const (&bp)[4],
                                               f32x16 (&acc)[TM][TN]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float4 a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const float4 *>(ap[c] + OFF + i * 32 * 32);
#pragma unroll
        for (int i = 0; i < TN; ++i) b[i] = *reinterpret_cast<const float4 *>(bp[c] + OFF + i * 32 * 32);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                        j == 0 ? a[tm].x : j == 1 ? a[tm].y : j == 2 ? a[tm].z : a[tm].w,
                        j == 0 ? b[tn].x : j == 1 ? b[tn].y : j == 2 ? b[tn].z : b[tn].w,
                        (ZEROC && c == 0 && j == 0) ? zero : acc[tm][tn], 0, 0, 0);
                }
    }
}

// SF_FWD_FRAG_DB / SF_QUADROW_FRAG_DB (64 x 64 wave tiles of the zero-VALU kernels: fc forward / data gradient, conv2 data gradient): hipcc reads a
// group's four fragments right in front of the group's 16 MFMAs and waits for them (lgkmcnt(0)) with only the previous
// group's last MFMA still in the pipe: one bare LDS round trip per 16 MFMAs and wave.  1: the fragments of group c + 1
// are read behind the first 4 MFMAs of group c into a second register set (12 MFMAs = 768 cycles to land); 2: also the
// first group's fragments are read right behind the chunk's barrier, IN FRONT of the DMA instructions and the address
// work for the next chunk (`pre`), which then run in the shadow of that LDS round trip.  Same MFMAs in the same order.
#ifndef SF_FWD_FRAG_DB
#define SF_FWD_FRAG_DB 0      // fc forward / data gradient (k_fwd_glds_z<128, 128>): 2 is +-0 on the forward launch and +1..2 % on the
#endif                        // data gradient (its ReLU-mask prefetch already fills the register file): off
#ifndef SF_QUADROW_FRAG_DB
#define SF_QUADROW_FRAG_DB 2  // conv2 data gradient: -2.1 .. -2.8 % (profiles/r05_af_fragdb_rot.log, r05_ag_fragdb_spread.log)
#endif
template <int P> using HookPt = std::integral_constant<int, P>;
// hook(HookPt<0>): right behind the first group's fragment reads; <1>, <2>: behind the first / second 4 MFMAs of group 0
// (... = 3: half of the chunk's DMA instructions each -- a DMA instruction holds its wave's issue port while the
// address unit takes it, in a burst in front of the MFMAs that is dead time of this wave, between MFMAs it is covered by
// the 4 x 64 matrix-pipe cycles just issued); <3>: behind group 0, unfenced (address work for the chunk after next: hipcc
// may spread it over group 1's MFMAs).  SINK: how many MFMAs at the END of the chunk hipcc may move behind the NEXT chunk's
// barrier (they cover the fragment round trip there); the rest is fenced in front of it, so that the next chunk's DMA is
// not issued late.
#ifndef SF_FRAG_DB_SINK
#define SF_FRAG_DB_SINK 16
#endif
template <int TM, int TN, int OFF, bool ZEROC = false, bool SPREAD = false, typename F>
__device__ __forceinline__ void mma_chunk_ptrs_db(const float *const (&ap)[4], const float *const (&bp)[4],
                                                  f32x16 (&acc)[TM][TN], F &&hook) {
    float4 a[2][TM], b[2][TN];
    auto fetch = [&](int c) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a[c & 1][i] = *reinterpret_cast<const float4 *>(ap[c] + OFF + i * 32 * 32);
#pragma unroll
        for (int i = 0; i < TN; ++i) b[c & 1][i] = *reinterpret_cast<const float4 *>(bp[c] + OFF + i * 32 * 32);
    };
    auto mfmas = [&](int c, int j) {
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(
                    j == 0 ? a[c & 1][tm].x : j == 1 ? a[c & 1][tm].y : j == 2 ? a[c & 1][tm].z : a[c & 1][tm].w,
                    j == 0 ? b[c & 1][tn].x : j == 1 ? b[c & 1][tn].y : j == 2 ? b[c & 1][tn].z : b[c & 1][tn].w,
                    (ZEROC && c == 0 && j == 0) ? zero : acc[tm][tn], 0, 0, 0);
            }
    };
    constexpr int PER = TM * TN;  // MFMAs per k-step
    fetch(0);
    hook(HookPt<0>{});
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        mfmas(c, 0);
        if (c + 1 < 4 || c == 0) {
            __builtin_amdgcn_sched_barrier(0);
            if (c == 0) hook(HookPt<1>{});
            if (c + 1 < 4) fetch(c + 1);
            __builtin_amdgcn_sched_barrier(0);
        } else if (SF_FRAG_DB_SINK == 3 * PER) __builtin_amdgcn_sched_barrier(0);
        mfmas(c, 1);
        if (c == 0 && SPREAD) {
            __builtin_amdgcn_sched_barrier(0);
            hook(HookPt<2>{});
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c == 3 && SF_FRAG_DB_SINK == 2 * PER) __builtin_amdgcn_sched_barrier(0);
        mfmas(c, 2);
        if (c == 3 && SF_FRAG_DB_SINK == PER) __builtin_amdgcn_sched_barrier(0);
        mfmas(c, 3);
        if (c == 0) hook(HookPt<3>{});
        if (c == 3 && SF_FRAG_DB_SINK == 0) __builtin_amdgcn_sched_barrier(0);
    }
}

#ifndef SF_PIX_INCR
#define SF_PIX_INCR 1  // k_dgrad_pix(_z): the next chunk's (filter row, filter column, channel chunk) stepped, not decoded from q
#endif
#ifndef SF_PIX_LDSIMM
#define SF_PIX_LDSIMM 1  // k_dgrad_pix_z: LDS destinations of the DMA instructions as SGPR base + immediate (see glds16_si)
#endif
#ifndef SF_QUADROW_ROT
#define SF_QUADROW_ROT 0  // k_dgrad_quadrow_z: co-resident work-groups walk the group rows in different rotations (see the kernel)
#endif
#ifndef SF_QUADROW_PREP
#define SF_QUADROW_PREP 1  // k_dgrad_quadrow_z: DMA addresses prepared one chunk ahead (0: computed between barrier and DMA)
#endif
#ifndef SF_DGRAD_PIX_ZL_LITE
#define SF_DGRAD_PIX_ZL_LITE 1  // k_dgrad_pix_z: 1 = SADDR-form DMA only (fragment reads stay "runtime stage + VALU add")
#endif
// the chunk out of stage `stage` (run-time, wave-uniform) of a two-stage pipeline: a scalar branch picks one of two copies of
// the chunk whose stage offset is an immediate — the k-loops whose stage parity is not static (pipelines running across
// pixel / step boundaries) get VALU-free fragment addressing this way
template <int TM, int TN, int STAGE_F>
__device__ __forceinline__ void mma_chunk_stage(const float *const (&ap)[4], const float *const (&bp)[4], int stage,
                                                f32x16 (&acc)[TM][TN]) {
    if (stage == 0) mma_chunk_ptrs<TM, TN, 0>(ap, bp, acc);
    else mma_chunk_ptrs<TM, TN, STAGE_F>(ap, bp, acc);
}

// ============================================================================================== FORWARD (glds)
// out[m][n] = act( sum_k A[m][k] * Wt[n][k] + bias[n] ),  A = im2col view of the NHWC input, Wt = weights [Cout, K].
// ZL ("zero-VALU loop", TM*TN <= 2 tiles): the k-loop issues its DMA through glds16_s (uniform base + 32-bit lane offset)
// and reads its fragments through mma_chunk_ptrs_mid — two chunks per loop trip, so the pipeline stage is a compile-time
// constant.  The loop body is then MFMA + ds_read + DMA + scalar instructions only (the plain form spends 10 v_add_u32 and
// 6 v_lshl_add_u64 per 32 MFMAs on addresses; vector ALU instructions do not overlap with MFMAs on a SIMD).  Needs
// every per-lane operand offset < 4 GiB (the launcher checks).
// DYNLDS: the pipeline stages live in the launch's dynamic LDS (k_fwd_glds_zt runs two instantiations in one kernel: two
// static arrays would both be allocated); bx_shift: added to the row-tile index (the 64-row tail tiles of that kernel)
template <int BM, int BN, int WM, int WN, int NS, bool ZL, bool PERSIST = false, bool DYNLDS = false>
__device__ __forceinline__ void fwd_glds_body(ConvG g, const float *__restrict__ in, int64_t in_stride,
                                              const float *__restrict__ wt, const float *__restrict__ bias,
                                              float *__restrict__ out, int64_t Mtot, int k_per_split,
                                              float *__restrict__ partial, const float *__restrict__ dmask,
                                              int dmask_on, int rx, int ry, int rtot, int tap_perm, int bx_shift = 0) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int AI = BM / 32, BI = BN / 32;  // DMA instructions per wave and chunk (8 rows x 128 B each)
    constexpr int STAGE = (BM + BN) * 32;      // floats per pipeline stage
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1 && (NS == 2 || NS == 3), "4 waves per block");
    __shared__ __attribute__((aligned(1024))) float lds_static[DYNLDS ? 1 : NS * STAGE];
    extern __shared__ __attribute__((aligned(1024))) float lds_dynamic[];
    float *const lds = DYNLDS ? lds_dynamic : lds_static;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: LDS-DMA bases (M0) stay on the scalar unit
    const int wm = wave / WN, wn = wave % WN;
    // XCD-aware block order (rtot > 0: 1-D launch of 8 * ceil(rtot / 8) ids, see k_wgrad_glds): XCD c owns a contiguous
    // run of the logical order COLUMN TILE FASTEST, then row tile, then slice — the ry column tiles of one activation row
    // strip run next to each other on one XCD (fc layer, n = 32768: the 411 MB activation matrix was fetched once per
    // column tile, 1.65 GB per launch); the weight strips they differ in are small and shared by every row strip anyway.
    // PERSIST (k_fwd_glds_zp, single-column unsplit launches): the work-group walks the row tiles b, b + grid, ...
    const int ptiles = PERSIST ? (int)((Mtot + BM - 1) / BM) : 1;
    for (int ptile = PERSIST ? (int)blockIdx.x : 0; ptile < ptiles; ptile += PERSIST ? (int)gridDim.x : 1) {
    int bx = PERSIST ? ptile : (int)blockIdx.x + bx_shift, by = blockIdx.y, bz = blockIdx.z;
    if (rtot > 0) {
        const int per = (rtot + 7) >> 3, L = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
        if (L >= rtot) return;
        by = L % ry;
        const int t = L / ry;
        bx = t % rx;
        bz = t / rx;
    }
    const int64_t m0 = (int64_t)bx * BM;
    const int n0 = by * BN;
    const int N = g.Cout, K = g.K;
    // split-K (grids too small to fill the chip): slice z reduces k in [kbeg, kend) into partial[z], k_splitk_finish
    // adds the slices in ascending z, the bias and the activation
    const int kbeg = bz * k_per_split, kend = min(K, kbeg + k_per_split);

    // per-lane DMA sources: lane = (row-in-group lrow, chunk position lpos); position p holds chunk p ^ swz(row)
    const int lrow = lane >> 3, lpos = lane & 7;
    const float *asrc[AI], *bsrc[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int row = (i * 4 + wave) * 8 + lrow;
        int64_t m = m0 + row;
        m = m < Mtot ? m : Mtot - 1;  // rows past the end re-read the last row; their results are never stored
        const uint32_t smp = fdiv((uint32_t)m, g.dOHOW), pix = (uint32_t)m - smp * (uint32_t)(g.OH * g.OW);
        asrc[i] = in + (int64_t)smp * in_stride + patch_origin<false>(g, pix) + ((lpos ^ ((row >> 1) & 7)) << 2);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int row = (i * 4 + wave) * 8 + lrow;
        int n = n0 + row;
        n = n < N ? n : N - 1;
        bsrc[i] = wt + (int64_t)n * K + ((lpos ^ ((row >> 1) & 7)) << 2);
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a_ = 0; a_ < TM; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < TN; ++b_)
#pragma unroll
            for (int r_ = 0; r_ < 16; ++r_) acc[a_][b_][r_] = 0.f;

    auto issue = [&](int k0, int stage) {
        // chunk k0..k0+31 = channels c0.. of filter tap (kh, kw): one wave-uniform offset inside the input patch
        const uint32_t tap = fdiv((uint32_t)k0, g.dCin), c0 = (uint32_t)k0 - tap * (uint32_t)g.Cin;
        const uint32_t kh = fdiv(tap, g.dKW), kw = tap - kh * (uint32_t)g.KW;
        const int aoff = (int)((kh * (uint32_t)g.W + kw) * (uint32_t)g.Cin + c0);
        float *sa = lds + stage * STAGE, *sb = sa + BM * 32;
#pragma unroll
        for (int i = 0; i < AI; ++i) GLDS16(asrc[i] + aoff, sa + (i * 4 + wave) * 256);
#pragma unroll
        for (int i = 0; i < BI; ++i) GLDS16(bsrc[i] + k0, sb + (i * 4 + wave) * 256);
    };
    // NS-stage DMA pipeline: chunk t+NS-1 is issued while chunk t is in the matrix pipe.  Loads retire in order, so
    // "at most (NS-2) chunks' worth of DMA instructions outstanding" == "chunk t has landed" — never a vmcnt(0)
    // inside the loop for NS = 3.
    static_assert(NS == 2, "two LDS stages");
    // Chunk ORDER of a 4x4 stride-2 window over 32 channels (conv2: a chunk = one filter tap).  An input element is read
    // by the four taps (kh, kw), (kh, kw+2), (kh+2, kw), (kh+2, kw+2) of four different output pixels; in tap order those
    // reads are up to 10 chunks (~25 us, ~20 MB streamed through the XCD's 4 MB L2) apart, and 74 % of the im2col
    // re-reads leave the L2 (profiles/r05_g_traffic_xcd_rows_0.json: 642 MB fetched per launch against 368 MB read
    // once).  tapperm != 0: visit the taps in groups of those four, so that an element's uses are at most 3 chunks
    // apart.  The reduction is a sum over the same products in another order.
    const bool tapperm = tap_perm && g.KH == 4 && g.KW == 4 && g.S == 2 && g.Cin == 32 && kbeg == 0 && kend == K;
    auto kord = [&](int k0) {  // position in the visiting order -> first reduction index of the chunk visited there
        if (!tapperm) return k0;
        const int p = k0 >> 5, grp = p >> 2, e = p & 3;              // grp = (kh & 1) * 2 + (kw & 1) in visiting order
        const int kh = ((grp >> 1) & 1) + 2 * (e >> 1), kw = (grp & 1) + 2 * (e & 1);
        return (kh * 4 + kw) << 5;
    };
    int stage = 0, k0 = kbeg;
    // data-gradient launches with a ReLU mask (dmask = the layer's input activation): the tile's 16*TM*TN mask words are
    // fetched in front of the LAST chunk's MFMAs instead of in the epilogue (where every group of 16 loads was a bare
    // round trip to memory with nothing of this wave to overlap it)
    constexpr bool MASK_PRE = ZL && TM * TN == 4;
    float mpre[MASK_PRE ? TM : 1][MASK_PRE ? TN : 1][16];
    bool mask_pre = false;
    auto prefetch_mask = [&]() {
        if constexpr (MASK_PRE) {
            if (dmask_on && dmask && g.relu == 1 && m0 + BM <= Mtot && n0 + BN <= N) {
                const float *mk = dmask + (m0 + wm * TM * 32) * N + (n0 + wn * TN * 32);
                const uint32_t vo = (uint32_t)(4 * (lane >> 5)) * (uint32_t)N + (uint32_t)(lane & 31);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            mpre[tm][tn][r] = (mk + (int64_t)(tm * 32 + (r & 3) + 8 * (r >> 2)) * N + tn * 32)[vo];
                mask_pre = true;
            }
        }
    };
    if constexpr (ZL) {

        // ---- per-lane 32-bit byte offsets of the DMA sources (relative to `in` / `wt`), LDS fragment pointers of stage 0
        uint32_t avoff[AI], bvoff[BI];
#pragma unroll
        for (int i = 0; i < AI; ++i) avoff[i] = (uint32_t)((asrc[i] - in) * (int64_t)sizeof(float));
#pragma unroll
        for (int i = 0; i < BI; ++i) bvoff[i] = (uint32_t)((bsrc[i] - wt) * (int64_t)sizeof(float));
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)lds;
        const int r = lane & 31, h = lane >> 5, sw = (r >> 1) & 7;
        const float *apl[4], *bpl[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int pos = (((2 * c + h) ^ sw) << 2);
            apl[c] = lds + (wm * TM * 32 + r) * 32 + pos;
            bpl[c] = lds + BM * 32 + (wn * TN * 32 + r) * 32 + pos;
        }
        auto chunk_off = [&](int kn) {  // element offset of chunk kn inside an input patch (uniform)
            const uint32_t tap = fdiv((uint32_t)kn, g.dCin), c0 = (uint32_t)kn - tap * (uint32_t)g.Cin;
            const uint32_t kh = fdiv(tap, g.dKW), kw = tap - kh * (uint32_t)g.KW;
            return (int)((kh * (uint32_t)g.W + kw) * (uint32_t)g.Cin + c0);
        };
        auto dma = [&](int q, const float *abase, const float *bbase, int st) {  // DMA instruction q of a chunk
            if (q < AI) glds16_s(abase, avoff[q < AI ? q : 0], lds0 + (uint32_t)((st * STAGE + (q * 4 + wave) * 256) * 4));
            else glds16_s(bbase, bvoff[q >= AI ? q - AI : 0], lds0 + (uint32_t)((st * STAGE + BM * 32 + ((q - AI) * 4 + wave) * 256) * 4));
        };
        {
            const int kf = kord(kbeg);
            const float *ab = in + chunk_off(kf), *bb = wt + kf;
#pragma unroll
            for (int q = 0; q < AI + BI; ++q) dma(q, ab, bb, 0);
        }
        auto step = [&](auto stc, bool prefetch) {  // multiply the chunk in stage ST; stream the next one into ST ^ 1
            constexpr int ST = decltype(stc)::value;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            BARRIER_NOFENCE();
            const int kn = prefetch ? kord(k0 + 32) : 0;
            const float *ab = in + chunk_off(kn), *bb = wt + kn;
            if (!prefetch) prefetch_mask();  // (last chunk of the tile)
            if constexpr (TM * TN <= 2) {  // DMA instructions spread over the first three MFMA groups
                mma_chunk_ptrs_mid<TM, TN, ST * STAGE>(apl, bpl, acc, [&](int c) {
                    constexpr int TOT = AI + BI, PER = (TOT + 2) / 3;
                    if (prefetch) {
#pragma unroll
                        for (int q = 0; q < TOT; ++q)
                            if (q / PER == c) dma(q, ab, bb, ST ^ 1);
                    }
                });
            } else {  // 64 x 64 wave tiles: burst in front of the chunk (see the plain form below)
                auto burst = [&](int qlo, int qhi) {
                    if (prefetch) {
#pragma unroll
                        for (int q = 0; q < AI + BI; ++q)
                            if (q >= qlo && q < qhi) dma(q, ab, bb, ST ^ 1);
                    }
                };
                if constexpr (SF_FWD_FRAG_DB >= 2) {
                    mma_chunk_ptrs_db<TM, TN, ST * STAGE, false, SF_FWD_FRAG_DB == 3>(apl, bpl, acc, [&](auto pt) {
                        constexpr int P = decltype(pt)::value, HALF = (AI + BI) / 2;
                        if constexpr (SF_FWD_FRAG_DB == 2) {
                            if constexpr (P == 0) burst(0, AI + BI);
                        } else {
                            if constexpr (P == 1) burst(0, HALF);
                            if constexpr (P == 2) burst(HALF, AI + BI);
                        }
                    });
                } else {
                    burst(0, AI + BI);
                    if constexpr (SF_FWD_FRAG_DB == 1) mma_chunk_ptrs_db<TM, TN, ST * STAGE>(apl, bpl, acc, [](auto) {});
                    else mma_chunk_ptrs<TM, TN, ST * STAGE>(apl, bpl, acc);
                }
            }
            k0 += 32;
        };
        while (k0 + 64 < kend) {  // two chunks per trip: stages 0 and 1 as compile-time constants
            step(std::integral_constant<int, 0>{}, true);
            step(std::integral_constant<int, 1>{}, true);
        }
        if (k0 + 32 < kend) {
            step(std::integral_constant<int, 0>{}, true);
            step(std::integral_constant<int, 1>{}, false);
        } else {
            step(std::integral_constant<int, 0>{}, false);
        }
    } else {
    issue(kord(kbeg), 0);
    if constexpr (TM * TN <= 2) {
        // chunk k0 is multiplied out of `stage` while chunk k0+32 streams into the other stage; its DMA instructions
        // are spread over the first three MFMA groups (mma_chunk_rows_mid).  Measured: +0..+3 % for the 64x32 wave
        // tile, -11 % for the 64x64 one (16 MFMAs per group: the fences cost more fragment-read overlap than the DMA
        // placement wins), which therefore keeps the burst form below.
        for (; k0 + 32 < ((SF_GLDS_ABLATE & 8) ? kbeg + 64 : kend); k0 += 32, stage ^= 1) {
            if (!(SF_GLDS_ABLATE & 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            BARRIER_NOFENCE();  // chunk k0 is visible to all waves, the stage about to be refilled is no longer read
            }
            const int kn = kord(k0 + 32);
            const uint32_t tap = fdiv((uint32_t)kn, g.dCin), c0 = (uint32_t)kn - tap * (uint32_t)g.Cin;
            const uint32_t kh = fdiv(tap, g.dKW), kw = tap - kh * (uint32_t)g.KW;
            const int aoff = (int)((kh * (uint32_t)g.W + kw) * (uint32_t)g.Cin + c0);
            float *na = lds + (stage ^ 1) * STAGE, *nb = na + BM * 32;
            const float *sa = lds + stage * STAGE;
            mma_chunk_rows_mid<TM, TN>(sa, sa + BM * 32, wm * TM * 32, wn * TN * 32, lane, acc, [&](int c) {
                constexpr int TOT = AI + BI, PER = (TOT + 2) / 3;
#pragma unroll
                for (int q = 0; q < TOT; ++q) {
                    if (q / PER != c || (SF_GLDS_ABLATE & 1)) continue;
                    if (q < AI) GLDS16(asrc[q < AI ? q : 0] + aoff, na + (q * 4 + wave) * 256);
                    else GLDS16(bsrc[q >= AI ? q - AI : 0] + kn, nb + ((q - AI) * 4 + wave) * 256);
                }
            });
        }
    } else {
        for (; k0 + 32 < kend; k0 += 32, stage ^= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            BARRIER_NOFENCE();
            issue(k0 + 32, stage ^ 1);
            const float *sa = lds + stage * STAGE;
            mma_chunk_rows<TM, TN>(sa, sa + BM * 32, wm * TM * 32, wn * TN * 32, lane, acc);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BARRIER_NOFENCE();
    {
        const float *sa = lds + stage * STAGE;
        mma_chunk_rows<TM, TN>(sa, sa + BM * 32, wm * TM * 32, wn * TN * 32, lane, acc);
    }
    }  // !ZL
    // epilogue: uniform base pointer + one 32-bit lane offset; the activation kind and the "tile is complete" test
    // are hoisted out of the 16*TM*TN element loop (per element: bias add, max, address add, store — the first
    // version re-derived a 64-bit m*N+n and branched on the kind per element: 13 VALU + 3 quarter-rate multiplies)
    float *ob = (partial ? partial + (int64_t)bz * Mtot * N : out) + (m0 + wm * TM * 32) * N + (n0 + wn * TN * 32);
    const int rows_left = (int)min((int64_t)(TM * 32), Mtot - m0 - wm * TM * 32) - 4 * (lane >> 5);
    const int cols_left = N - (n0 + wn * TN * 32) - (lane & 31);
    const uint32_t voff = (uint32_t)(4 * (lane >> 5)) * (uint32_t)N + (uint32_t)(lane & 31);
    const bool full = m0 + BM <= Mtot && n0 + BN <= N;
    if (dmask_on) {  // data gradient of a linear layer (sf_conv_dgrad): out = acc * act'(dmask), no bias
        const float *mk = dmask ? dmask + (m0 + wm * TM * 32) * N + (n0 + wn * TN * 32) : nullptr;
        if (!dmask) {
            if (full) store_dgrad_tile<TM, TN, 0, true>(acc, ob, mk, voff, N, rows_left, cols_left, 0);
            else store_dgrad_tile<TM, TN, 0, false>(acc, ob, mk, voff, N, rows_left, cols_left, 0);
        } else if (MASK_PRE && mask_pre) {  // (complete tile, ReLU) mask words already in registers
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = tm * 32 + (r & 3) + 8 * (r >> 2);
                        (ob + (int64_t)rr * N + tn * 32)[voff] = act_bwd_mul<1>(acc[tm][tn][r], mpre[MASK_PRE ? tm : 0][MASK_PRE ? tn : 0][r], 1);
                    }
        } else if (g.relu == 1) {
            if (full) store_dgrad_tile<TM, TN, 1, true>(acc, ob, mk, voff, N, rows_left, cols_left, 1);
            else store_dgrad_tile<TM, TN, 1, false>(acc, ob, mk, voff, N, rows_left, cols_left, 1);
        } else {
            store_dgrad_tile<TM, TN, -1, false>(acc, ob, mk, voff, N, rows_left, cols_left, g.relu);
        }
    } else if (partial) {
        if (full) store_fwd_tile<TM, TN, 0, true>(acc, ob, voff, N, rows_left, cols_left, nullptr, lane & 31, 0);
        else store_fwd_tile<TM, TN, 0, false>(acc, ob, voff, N, rows_left, cols_left, nullptr, lane & 31, 0);
    } else if (g.relu == 1) {
        if (full) store_fwd_tile<TM, TN, 1, true>(acc, ob, voff, N, rows_left, cols_left, bias ? bias + n0 + wn * TN * 32 : nullptr, lane & 31, 1);
        else store_fwd_tile<TM, TN, 1, false>(acc, ob, voff, N, rows_left, cols_left, bias ? bias + n0 + wn * TN * 32 : nullptr, lane & 31, 1);
    } else {
        store_fwd_tile<TM, TN, -1, false>(acc, ob, voff, N, rows_left, cols_left, bias ? bias + n0 + wn * TN * 32 : nullptr, lane & 31, g.relu);
    }
    if (PERSIST) BARRIER_NOFENCE();  // every wave is past its last fragment read before the next tile's DMA lands
    }  // ptile
}

template <int BM, int BN, int WM, int WN, int NS>
__global__ __launch_bounds__(256) void k_fwd_glds(ConvG g, const float *__restrict__ in, int64_t in_stride,
                                                  const float *__restrict__ wt, const float *__restrict__ bias,
                                                  float *__restrict__ out, int64_t Mtot, int k_per_split,
                                                  float *__restrict__ partial, const float *__restrict__ dmask,
                                                  int dmask_on, int rx = 0, int ry = 0, int rtot = 0, int tap_perm = 0) {
    fwd_glds_body<BM, BN, WM, WN, NS, false>(g, in, in_stride, wt, bias, out, Mtot, k_per_split, partial, dmask, dmask_on, rx,
                                             ry, rtot, tap_perm);
}
// ... and persistent (experiment: SF_GLDS_PERSIST)
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void k_fwd_glds_zp(ConvG g, const float *__restrict__ in, int64_t in_stride,
                                                       const float *__restrict__ wt, const float *__restrict__ bias,
                                                       float *__restrict__ out, int64_t Mtot, int k_per_split) {
    fwd_glds_body<BM, BN, WM, WN, 2, true, true>(g, in, in_stride, wt, bias, out, Mtot, k_per_split, nullptr, nullptr, 0, 0, 0, 0, 0);
}
// TAIL SPLIT (single-column unsplit launches): every tile of such a launch costs the same, so a launch of T tiles on R
// resident work-groups runs floor(T / R) full rounds and one more for the T mod R tiles left over — conv2 at a rollout
// step: 2592 tiles on 512 = five rounds + 32 tiles that keep a sixteenth of the chip busy for a sixth round.  Here the
// last `T mod R` 128-row tiles are run as twice as many 64-row tiles (blocks main_tiles .. of the same launch, the
// <BM/2, BN> instantiation of the same body): the last round is half as long (rollout-size launch of conv2 184 - 191 ->
// 179 - 181 us, profiles/r05_ad_fwd_tail_split.log).  Every single-column unsplit launch takes this kernel; one whose tile
// count leaves nothing to split (n = 32768: 27 full rounds) passes main_tiles = all of them.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void k_fwd_glds_zt(ConvG g, const float *__restrict__ in, int64_t in_stride,
                                                       const float *__restrict__ wt, const float *__restrict__ bias,
                                                       float *__restrict__ out, int64_t Mtot, int k_per_split, int main_tiles,
                                                       int tap_perm) {
    if ((int)blockIdx.x < main_tiles)
        fwd_glds_body<BM, BN, WM, WN, 2, true, false, true>(g, in, in_stride, wt, bias, out, Mtot, k_per_split, nullptr, nullptr, 0,
                                                            0, 0, 0, tap_perm);
    else
        fwd_glds_body<BM / 2, BN, WM, WN, 2, true, false, true>(g, in, in_stride, wt, bias, out, Mtot, k_per_split, nullptr,
                                                                nullptr, 0, 0, 0, 0, tap_perm, main_tiles);
}
// the same kernel with the zero-VALU k-loop (see fwd_glds_body, ZL)
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void k_fwd_glds_z(ConvG g, const float *__restrict__ in, int64_t in_stride,
                                                    const float *__restrict__ wt, const float *__restrict__ bias,
                                                    float *__restrict__ out, int64_t Mtot, int k_per_split,
                                                    float *__restrict__ partial, const float *__restrict__ dmask,
                                                    int dmask_on, int rx = 0, int ry = 0, int rtot = 0, int tap_perm = 0) {
    fwd_glds_body<BM, BN, WM, WN, 2, true>(g, in, in_stride, wt, bias, out, Mtot, k_per_split, partial, dmask, dmask_on, rx, ry,
                                           rtot, tap_perm);
}

// out[m][n] = sum_k A1[m][k] W1t[n][k] + sum_k A2[m][k] W2t[n][k] + bias1[n] + bias2[n]: TWO linear layers into one
// accumulator (one inference step of an LSTM: x W_ih^T + h W_hh^T + b_ih + b_hh; sf_linear_fwd_dual).  The k_fwd_glds
// pipeline with a segment switch per 32-chunk: chunks below K1 stream (A1, W1t), the rest (A2, W2t) — the short first
// product (K1 = 64: two chunks) rides in the long one's pipeline instead of being a launch of its own that is all fill
// and drain.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void k_fwd_glds2(const float *__restrict__ a1, int64_t lda1, const float *__restrict__ w1t,
                                                   const float *__restrict__ bias1, int K1, const float *__restrict__ a2,
                                                   int64_t lda2, const float *__restrict__ w2t,
                                                   const float *__restrict__ bias2, int K2, float *__restrict__ out,
                                                   int64_t Mtot, int N, int gru_H) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int AI = BM / 32, BI = BN / 32;
    constexpr int STAGE = (BM + BN) * 32;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 waves per block");
    __shared__ __attribute__((aligned(1024))) float lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    // gru_H > 0 (one GRU inference step, N = 4*gru_H, BN divides gru_H): columns [0, 2H) = x W_i{r,z}^T + h W_h{r,z}^T + both
    // biases; [2H, 3H) = x W_in^T + b_in only; [3H, 4H) = h W_hn^T + b_hn only (weight / bias row n - H of the second layer):
    // the candidate gate needs its recurrent part separately, n = tanh(x_n + r * h_n).  Uniform per work-group.
    const bool use1 = gru_H == 0 || n0 < 3 * gru_H, use2 = gru_H == 0 || n0 < 2 * gru_H || n0 >= 3 * gru_H;
    const int row2_off = (gru_H > 0 && n0 >= 3 * gru_H) ? -gru_H : 0;
    const int kbeg = use1 ? 0 : K1, K = use2 ? K1 + K2 : K1;
    const int lrow = lane >> 3, lpos = lane & 7;
    const float *as1[AI], *as2[AI], *bs1[BI], *bs2[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int row = (i * 4 + wave) * 8 + lrow;
        int64_t m = m0 + row;
        m = m < Mtot ? m : Mtot - 1;
        const int sw = (lpos ^ ((row >> 1) & 7)) << 2;
        as1[i] = a1 + m * lda1 + sw;
        as2[i] = a2 + m * lda2 + sw - K1;  // indexed with the global k
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int row = (i * 4 + wave) * 8 + lrow;
        int n = n0 + row;
        n = n < N ? n : N - 1;
        const int sw = (lpos ^ ((row >> 1) & 7)) << 2;
        bs1[i] = w1t + (int64_t)(use1 ? n : 0) * K1 + sw;
        bs2[i] = w2t + (int64_t)(use2 ? n + row2_off : 0) * K2 + sw - K1;
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a_ = 0; a_ < TM; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < TN; ++b_)
#pragma unroll
            for (int r_ = 0; r_ < 16; ++r_) acc[a_][b_][r_] = 0.f;
    auto issue = [&](int k0, int stage) {
        float *sa = lds + stage * STAGE, *sb = sa + BM * 32;
        const bool first = k0 < K1;  // (uniform)
#pragma unroll
        for (int i = 0; i < AI; ++i) GLDS16((first ? as1[i] : as2[i]) + k0, sa + (i * 4 + wave) * 256);
#pragma unroll
        for (int i = 0; i < BI; ++i) GLDS16((first ? bs1[i] : bs2[i]) + k0, sb + (i * 4 + wave) * 256);
    };
    issue(kbeg, 0);
    int stage = 0, k0 = kbeg;
    for (; k0 + 32 < K; k0 += 32, stage ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        BARRIER_NOFENCE();
        issue(k0 + 32, stage ^ 1);
        const float *sa = lds + stage * STAGE;
        mma_chunk_rows<TM, TN>(sa, sa + BM * 32, wm * TM * 32, wn * TN * 32, lane, acc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BARRIER_NOFENCE();
    {
        const float *sa = lds + stage * STAGE;
        mma_chunk_rows<TM, TN>(sa, sa + BM * 32, wm * TM * 32, wn * TN * 32, lane, acc);
    }
    // epilogue: (acc + bias1) + bias2, the order of "x W_ih^T + b_ih" + "h W_hh^T + b_hh" up to the product sums
    const int lcol = lane & 31;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int col = n0 + wn * TN * 32 + tn * 32 + lcol;
        if (col >= N) continue;
        const float b1 = (bias1 && use1) ? bias1[col] : 0.f, b2 = (bias2 && use2) ? bias2[col + row2_off] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t row = m0 + wm * TM * 32 + tm * 32 + FRAG_ROW(r, lane);
                if (row < Mtot) out[row * N + col] = (acc[tm][tn][r] + b1) + b2;
            }
    }
}

// wt[n][k] = w[k][n]  (weights are kept K-major for the data-gradient; the glds forward wants them Cout-major)
__global__ __launch_bounds__(256) void k_transpose(const float *__restrict__ w, float *__restrict__ wt, int K, int N) {
    __shared__ float t[32][33];
    const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int k = k0 + ty + i, n = n0 + tx;
        t[ty + i][tx] = (k < K && n < N) ? w[(int64_t)k * N + n] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        const int n = n0 + ty + i, k = k0 + tx;
        if (n < N && k < K) wt[(int64_t)n * K + k] = t[tx][ty + i];
    }
}

// A DMA source for rows that must contribute nothing (taps outside dY, reduction rows past the end).
__device__ __attribute__((aligned(128))) const float sf_zero_page[32] = {};

// ============================================================================================== DATA GRADIENT, pixel-major
// The class-decomposed gather form above still multiplies by structural zeros at the image border: a row tile mixes
// pixels for which a tap is inside dY with pixels for which it is not (conv3: 81 input pixels x 9 taps, only 49 x 9
// pairs exist: 40 % of the MFMA work is 0 * w).  Here a row tile is BM SAMPLES at ONE input pixel, so whether a tap
// exists is uniform over the tile and non-existent taps are skipped: the MFMA work equals the algorithmic
// 2 * n*OH*OW * Cout * KH*KW*Cin exactly.  A block walks one input row (ih fixed, iw = 0..W-1) with the DMA pipeline
// running across pixel boundaries; rows of both operands are contiguous 128-byte runs of output channels:
//   A[s][co] = dY[s, oh, ow, co]   (per-lane base = sample, wave-uniform offset = (oh, ow, co-chunk))
//   B[c][co] = W[(kh*KW + kw)*Cin + c][co]
// Work-group ids are dealt round-robin to the 8 XCDs; the ids are re-mapped so that all pixel rows of one sample tile
// run on the same XCD and share its L2 copy of that tile's dY.
template <int BM, int BN, int WM, int WN, bool ZL>
__device__ __forceinline__ void dgrad_pix_body(ConvG g, const float *__restrict__ dy, const float *__restrict__ w,
                                               const float *__restrict__ in_act, float *__restrict__ din,
                                               int nsamples, int ntiles, int tiles8, int lpt) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int AI = BM / 32, BI = BN / 32;
    constexpr int STAGE = (BM + BN) * 32;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1 && BI >= 1, "4 waves per block");
    __shared__ __attribute__((aligned(1024))) float lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: LDS-DMA bases (M0) stay on the scalar unit
    const int wm = wave / WN, wn = wave % WN;
    const int Cin = g.Cin, Cout = g.Cout, S = g.S, OH = g.OH, OW = g.OW;
    // (sample tile, input row, Cin tile) from the XCD-swizzled linear id
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
    const int CC = Cout >> 5;  // 32-chunks per tap

    const int lrow = lane >> 3, lpos = lane & 7;
    const float *asrc[AI], *bsrc[BI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int row = (i * 4 + wave) * 8 + lrow;
        int s = s0 + row;
        s = s < nsamples ? s : nsamples - 1;
        asrc[i] = dy + (int64_t)s * (OH * OW) * Cout + ((lpos ^ ((row >> 1) & 7)) << 2);
    }
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int row = (i * 4 + wave) * 8 + lrow;
        int c = n0 + row;
        c = c < Cin ? c : Cin - 1;
        bsrc[i] = w + (int64_t)c * Cout + ((lpos ^ ((row >> 1) & 7)) << 2);
    }
    // taps of input row ih: kh = ph + S*a, oh = ihc - a in [0, OH)  ->  a in [a_lo, a_hi]
    const int ph = ih % S, ihc = ih / S, KHs = (g.KH - ph + S - 1) / S;
    const int a_lo = ihc - OH + 1 > 0 ? ihc - OH + 1 : 0, a_hi = ihc < KHs - 1 ? ihc : KHs - 1;

    f32x16 acc[TM][TN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a_ = 0; a_ < TM; ++a_)
#pragma unroll
            for (int b_ = 0; b_ < TN; ++b_)
#pragma unroll
                for (int r_ = 0; r_ < 16; ++r_) acc[a_][b_][r_] = 0.f;
    };
    // epilogue addressing: 32-bit element offsets (launcher guarantees n*H*W*Cin < 2^31); everything that depends on
    // the accumulator register index is wave-uniform, so one VGPR offset per lane suffices
    const uint32_t sstride = (uint32_t)(g.H * g.W * Cin);
    const int srow = s0 + wm * TM * 32 + 4 * (lane >> 5), ccol = n0 + wn * TN * 32 + (lane & 31);
    const uint32_t obase = (uint32_t)srow * sstride + (uint32_t)(ih * g.W) * (uint32_t)Cin + (uint32_t)ccol;
    const int slim = nsamples - srow;  // rows r with rowconst(r) < slim exist
    // Epilogue, software-pipelined against the MFMA phases (a plain "load in_act, multiply, store" tail at the end of
    // every pixel cost 40 % of the kernel: scattered 128-byte rows, nothing to overlap with).  The activation values of
    // pixel p are prefetched into registers during p's first chunk; at the end of p the masked result is parked in
    // registers and written out during the first chunk of the NEXT pixel, so neither the load latency nor the store
    // drain sits in front of a vmcnt(0).
    float pend[TM][TN][16], actv[TM][TN][16];
    // Complete tiles (every row a real sample, every column a real channel — all but the last tile) take the paths
    // without per-element predicates: address = uniform (pixel, fragment row, fragment column) part + obase.
    const bool full = s0 + BM <= nsamples && n0 + BN <= Cin;
    auto prefetch_act = [&](int iw) {
        if (!in_act) return;
        const uint32_t pix = (uint32_t)iw * (uint32_t)Cin;
        if (full) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rc = tm * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        actv[tm][tn][r] = (in_act + (size_t)(pix + (uint32_t)rc * sstride + (uint32_t)(tn * 32)))[obase];
                }
            return;
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rc = tm * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const bool ok = rc < slim && ccol + tn * 32 < Cin;
                    const uint32_t o = obase + pix + (uint32_t)rc * sstride + (uint32_t)(tn * 32);
                    actv[tm][tn][r] = in_act[ok ? o : 0u];  // rows past the last sample: any valid address
                }
            }
    };
    auto park_pixel = [&]() {
        const int akind = g.relu;
        auto park = [&](auto kc) {
            constexpr int KIND = decltype(kc)::value;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        pend[tm][tn][r] = act_bwd_mul<KIND>(acc[tm][tn][r], actv[tm][tn][r], akind);
        };
        if (!in_act) park(std::integral_constant<int, 0>{});
        else if (akind == 1) park(std::integral_constant<int, 1>{});
        else park(std::integral_constant<int, -1>{});
    };
    auto store_pixel = [&](int iw) {
        const uint32_t pix = (uint32_t)iw * (uint32_t)Cin;
        if (full) {
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rc = tm * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        (din + (size_t)(pix + (uint32_t)rc * sstride + (uint32_t)(tn * 32)))[obase] = pend[tm][tn][r];
                }
            return;
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rc = tm * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    if (rc < slim && ccol + tn * 32 < Cin) {
                        const uint32_t o = obase + pix + (uint32_t)rc * sstride + (uint32_t)(tn * 32);
                        din[o] = pend[tm][tn][r];
                    }
                }
            }
    };
    auto zero_pixel = [&](int iw) {  // an input pixel no filter tap reaches: gradient 0
        const uint32_t pix = (uint32_t)iw * (uint32_t)Cin;
        for (int rc = 0; rc < TM * 32; rc += 1) {
            const int rr = (rc & 3) + 8 * ((rc & 15) >> 2) + 32 * (rc >> 4);  // same row set as the fragments
            for (int tn = 0; tn < TN; ++tn)
                if (rr < slim && ccol + tn * 32 < Cin) din[obase + pix + (uint32_t)rr * sstride + (uint32_t)(tn * 32)] = 0.f;
        }
    };
    // wave-uniform walk over the pixels of this input row; per pixel the reduction runs over q = (a, b, cc)
    struct Px { int iw, pw, iwc, b_lo, nb, total; };
    auto pixel = [&](int iw) {  // first pixel >= iw that some tap reaches (iw = W: none left)
        Px p;
        for (p.iw = iw; p.iw < g.W; ++p.iw) {
            p.pw = p.iw % S; p.iwc = p.iw / S;
            const int KWs = (g.KW - p.pw + S - 1) / S;
            p.b_lo = p.iwc - OW + 1 > 0 ? p.iwc - OW + 1 : 0;
            const int b_hi = p.iwc < KWs - 1 ? p.iwc : KWs - 1;
            p.nb = b_hi - p.b_lo + 1;
            p.total = (a_hi - a_lo + 1) * p.nb * CC;
            if (p.nb > 0 && a_hi >= a_lo) return p;
        }
        p.total = 0;
        return p;
    };
    // ZL (k_dgrad_pix_z): DMA as uniform base + 32-bit lane offset (glds16_s), fragments through per-lane LDS pointers with
    // the stage as an immediate (mma_chunk_stage): no vector-ALU instruction in the reduction loop (see fwd_glds_body)
    uint32_t avoff[AI], bvoff[BI];
    const float *apl[4], *bpl[4];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)lds;
    const uint32_t ldsw = lds0 + (uint32_t)wave * 1024u;  // (SF_PIX_LDSIMM) this wave's 8 rows of every 32-row DMA group
    if constexpr (ZL) {
#pragma unroll
        for (int i = 0; i < AI; ++i) avoff[i] = (uint32_t)((asrc[i] - dy) * (int64_t)sizeof(float));
#pragma unroll
        for (int i = 0; i < BI; ++i) bvoff[i] = (uint32_t)((bsrc[i] - w) * (int64_t)sizeof(float));
        const int r = lane & 31, h = lane >> 5, sw = (r >> 1) & 7;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int pos = (((2 * c + h) ^ sw) << 2);
            apl[c] = lds + (wm * TM * 32 + r) * 32 + pos;
            bpl[c] = lds + BM * 32 + (wn * TN * 32 + r) * 32 + pos;
        }
    }
    // SF_PIX_INCR: the chunk to fetch next is carried as (filter-row index a, filter-column index b, channel chunk cc) and
    // stepped (cc fastest, then b, then a = the order of q) instead of being decoded from q -- q / CC, tap / nb and tap % nb
    // are divisions by run-time values, ~60 scalar instructions (and SGPRs spilled to VGPR lanes around them) between the
    // chunk's barrier and its DMA instructions in a kernel whose chunk is only 32 MFMAs per wave.  Same addresses.
    struct Cq { int a, b, cc; };
    auto first_chunk = [&](const Px &p) { return Cq{a_lo, p.b_lo, 0}; };
    auto next_chunk = [&](Cq &c, const Px &p) {
        if (++c.cc == CC) {
            c.cc = 0;
            if (++c.b == p.b_lo + p.nb) { c.b = p.b_lo; ++c.a; }
        }
    };
    auto issue_c = [&](const Px &p, int a, int b, int cc, int stage) {
        const int oh = ihc - a, ow = p.iwc - b, kh = ph + a * S, kw = p.pw + b * S;
        const int64_t aoff = (int64_t)(oh * OW + ow) * Cout + cc * 32;
        const int64_t boff = (int64_t)((kh * g.KW + kw) * Cin) * Cout + cc * 32;
        if constexpr (ZL) {
            const float *ab = dy + aoff, *bb = w + boff;
            if constexpr (SF_PIX_LDSIMM) {
                // destination = (lds0 + wave * 1 KiB + stage * STAGE) [one scalar add per chunk] + an immediate per instruction
                const uint32_t sb = ldsw + (uint32_t)stage * (uint32_t)(STAGE * 4);
                sf_static_for<0, AI>([&](auto ic) {
                    constexpr int I = decltype(ic)::value;
                    glds16_si<I * 4096>(ab, avoff[I], sb);
                });
                sf_static_for<0, BI>([&](auto ic) {
                    constexpr int I = decltype(ic)::value;
                    glds16_si<BM * 128 + I * 4096>(bb, bvoff[I], sb);
                });
                return;
            }
#pragma unroll
            for (int i = 0; i < AI; ++i) glds16_s(ab, avoff[i], lds0 + (uint32_t)((stage * STAGE + (i * 4 + wave) * 256) * 4));
#pragma unroll
            for (int i = 0; i < BI; ++i)
                glds16_s(bb, bvoff[i], lds0 + (uint32_t)((stage * STAGE + BM * 32 + (i * 4 + wave) * 256) * 4));
            return;
        }
        float *sa = lds + stage * STAGE, *sb = sa + BM * 32;
#pragma unroll
        for (int i = 0; i < AI; ++i) GLDS16(asrc[i] + aoff, sa + (i * 4 + wave) * 256);
#pragma unroll
        for (int i = 0; i < BI; ++i) GLDS16(bsrc[i] + boff, sb + (i * 4 + wave) * 256);
    };
    auto issue = [&](const Px &p, int q, int stage) {
        const int tap = q / CC, cc = q - tap * CC;  // CC, nb: small wave-uniform divisors (scalar unit)
        issue_c(p, a_lo + tap / p.nb, p.b_lo + tap % p.nb, cc, stage);
    };
    // (preparing a chunk's DMA bases one chunk ahead as k_dgrad_quadrow_z does costs hipcc 256 registers + 550 bytes of
    // scratch in this kernel: not used)
    Px cur = pixel(0);
    for (int z = 0; z < cur.iw; ++z) zero_pixel(z);
    Cq nq{0, 0, 0};  // SF_PIX_INCR: the chunk the next DMA round fetches
    if (cur.iw < g.W) {
        if constexpr (SF_PIX_INCR) {
            nq = first_chunk(cur);
            issue_c(cur, nq.a, nq.b, nq.cc, 0);
            next_chunk(nq, cur);
        } else issue(cur, 0, 0);
    }
    int stage = 0, parked = -1;  // parked: pixel whose result waits in pend[]
    while (cur.iw < g.W) {
        const Px nx = pixel(cur.iw + 1);
        auto issue_next = [&](int q, int stg) {  // chunk q + 1 of this pixel, or the first one of the next pixel
            if constexpr (SF_PIX_INCR) {
                if (q + 1 < cur.total) {
                    issue_c(cur, nq.a, nq.b, nq.cc, stg);
                    next_chunk(nq, cur);
                } else if (nx.iw < g.W) {  // the DMA pipeline runs across the pixel boundary
                    nq = first_chunk(nx);
                    issue_c(nx, nq.a, nq.b, nq.cc, stg);
                    next_chunk(nq, nx);
                }
            } else {
                if (q + 1 < cur.total) issue(cur, q + 1, stg);
                else if (nx.iw < g.W) issue(nx, 0, stg);
            }
        };
        zero_acc();
        if constexpr (ZL && !SF_DGRAD_PIX_ZL_LITE) {
            // every tap is CC chunks and CC is even (the launcher's contract for this form: Cout % 64 == 0), so a pixel
            // always starts in stage 0: two chunks per trip with the stage as a compile-time constant
            auto chunk = [&](int q, auto stc) {
                constexpr int ST = decltype(stc)::value;  // (only the even chunk of a pair can be a pixel's first)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (ST == 0 && q == 0 && parked >= 0) store_pixel(parked);
                issue_next(q, ST ^ 1);
                if (ST == 0 && q == 0) prefetch_act(cur.iw);
                mma_chunk_ptrs<TM, TN, ST * STAGE>(apl, bpl, acc);
            };
            for (int q = 0; q < cur.total; q += 2) {
                chunk(q, std::integral_constant<int, 0>{});
                chunk(q + 1, std::integral_constant<int, 1>{});
            }
        } else
        // (counted waits — chunk 1's DMA issued in front of the parked pixel's stores, s_waitcnt vmcnt(63 / 32) instead of 0 —
        // measured 1 - 3 % SLOWER here and 1.6 % slower in the row-walking kernel: profiles/r05_z_dgrad_lazy_wait.log)
        for (int q = 0; q < cur.total; ++q, stage ^= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (q == 0 && parked >= 0) store_pixel(parked);
            issue_next(q, stage ^ 1);
            if (q == 0) prefetch_act(cur.iw);
            const float *sa = lds + stage * STAGE;
            mma_chunk_rows<TM, TN>(sa, sa + BM * 32, wm * TM * 32, wn * TN * 32, lane, acc);
        }
        park_pixel();
        parked = cur.iw;
        for (int z = cur.iw + 1; z < nx.iw; ++z) zero_pixel(z);  // pixels no tap reaches
        cur = nx;
    }
    if (parked >= 0) store_pixel(parked);
}
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void k_dgrad_pix(ConvG g, const float *__restrict__ dy, const float *__restrict__ w,
                                                   const float *__restrict__ in_act, float *__restrict__ din,
                                                   int nsamples, int ntiles, int tiles8, int lpt) {
    dgrad_pix_body<BM, BN, WM, WN, false>(g, dy, w, in_act, din, nsamples, ntiles, tiles8, lpt);
}
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void k_dgrad_pix_z(ConvG g, const float *__restrict__ dy, const float *__restrict__ w,
                                                     const float *__restrict__ in_act, float *__restrict__ din,
                                                     int nsamples, int ntiles, int tiles8, int lpt) {
    dgrad_pix_body<BM, BN, WM, WN, true>(g, dy, w, in_act, din, nsamples, ntiles, tiles8, lpt);
}

// ============================================================================================== WEIGHT GRADIENT (glds)
// partial[z][k][n] = sum_{m in split z} A[m][k] * dY[m][n]   (A = im2col view of the f32 NHWC input)
// Both operands are reduction-major in memory already (a row = one reduction index m, free index contiguous), so the
// DMA image is As[32 m][BK k], Bs[32 m][BN n].  A lane of the MFMA A operand owns an output row; output rows are
// relabelled so that lane i owns the TM CONSECUTIVE k's TM*i .. TM*i+TM-1 (one per fragment): one ds_read_b32/b64/
// b128 feeds TM MFMAs (same for the TN fragments of dY).  Odd reduction rows are stored with their two 128-byte
// halves swapped (chunk ^ 8) so the two half-waves of a 4-byte read hit different banks.
template <int W> struct VecW;
template <> struct VecW<1> { typedef float T; };
template <> struct VecW<2> { typedef float2 T; };
template <> struct VecW<4> { typedef float4 T; };
template <int W>
__device__ __forceinline__ float vec_elem(const typename VecW<W>::T &v, int i) {
    if constexpr (W == 1) return v;
    else if constexpr (W == 2) return i == 0 ? v.x : v.y;
    else return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w;
}

// ZL (k_wgrad_glds_z, LINEAR layers only: 1x1 window on a 1x1 image, i.e. reduction row m = sample m): the row decode that
// sits in the general kernel's k-loop (two fast divisions, the patch origin, one ds_bpermute per DMA instruction) collapses
// to "uniform base + per-lane constant": both operands' DMA in SADDR form (glds16_s), fragment reads as VGPR + immediate
// with two chunks per trip (static stages) — no vector-ALU instruction in the reduction loop besides the bias column sums.
template <int BK, int BN, int WM, int WN, bool ZL>
__device__ __forceinline__ void wgrad_glds_body(ConvG g, const float *__restrict__ in, int64_t in_stride,
                                                const float *__restrict__ dy, float *__restrict__ partial,
                                                float *__restrict__ partial_b, int64_t Mtot, int64_t m_per_split,
                                                int rx, int ry, int rtot) {
    constexpr int TM = BK / WM / 32, TN = BN / WN / 32;
    static_assert(WM * WN == 4 && (TM == 1 || TM == 2 || TM == 4) && (TN == 1 || TN == 2 || TN == 4), "tile");
    // XCD-aware block order (rtot > 0: 1-D launch of 8 * ceil(rtot / 8) blocks).  The hardware deals consecutive block ids
    // to the 8 XCDs round-robin, so in a plain 3-D launch the tiles that share an operand strip — the ry column tiles of a
    // weight-row strip, the rx row tiles of a dY column strip, all of one reduction slice — land on 8 different L2s and
    // every strip is fetched from memory by each of them (fc layer, n = 32768: 2.29 GB per launch against 0.48 GB of
    // operands).  Here XCD c takes the CONTIGUOUS run [c * per, (c + 1) * per) of the logical order (x fastest, then y,
    // then z): a slice's tiles sit on one or two XCDs and walk their shared rows through the same L2 together.
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (rtot > 0) {
        const int per = (rtot + 7) >> 3, L = (int)(blockIdx.x & 7u) * per + (int)(blockIdx.x >> 3);
        if (L >= rtot) return;
        bx = L % rx;
        const int t = L / rx;
        by = t % ry;
        bz = t / ry;
    }
    constexpr int A_RPI = 256 / BK, B_RPI = 256 / BN;  // reduction rows per 1-KiB DMA instruction
    constexpr int AI = 32 / A_RPI / 4, BI = 32 / B_RPI / 4;  // DMA instructions per wave and chunk
    constexpr int STAGE = (BK + BN) * 32;
    __shared__ __attribute__((aligned(1024))) float lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: LDS-DMA bases (M0) stay on the scalar unit
    const int wm = wave / WN, wn = wave % WN;
    const int k0row = bx * BK, n0 = by * BN;
    const int N = g.Cout, K = g.K;
    const int64_t mbeg = (int64_t)bz * m_per_split;
    const int64_t mend = (mbeg + m_per_split < Mtot) ? mbeg + m_per_split : Mtot;

    // DMA lanes: A instruction j covers reduction rows j*A_RPI.. ; lane -> (row-in-instr, 16-byte position).
    // A reduction row m = (sample, oh, ow) moves every chunk, so its decode (two fast divisions, the patch origin)
    // sits inside the k-loop.  A wave's AI instructions touch only NROW = AI*A_RPI = 8 distinct rows per chunk: lane
    // L decodes row L & 7 ONCE and every instruction fetches its row's byte offset with one ds_bpermute, then adds the
    // lane's loop-invariant tap offset.  (First version: every lane decoded the row of every instruction — 206 VALU
    // incl. 62 quarter-rate multiplies per 64 MFMAs, 4.4 VALU per MFMA in the SQ counters.  Doing the decode on the
    // scalar unit instead — rows are wave-uniform for BK = 256 — was SLOWER: 280 dependent SALU instructions per
    // chunk delay the DMA issue.)  Offsets are 32-bit bytes: the launcher guarantees n * in_stride * 4 < 2^32.
    constexpr int ACH = BK / 4, BCH = BN / 4;  // 16-byte chunks per row
    constexpr int NROW = AI * A_RPI;
    static_assert(B_RPI % 2 == 0 && (A_RPI == 1 || A_RPI % 2 == 0) && NROW == 8, "row parity must be a lane constant");
    const int a_r = lane / ACH, a_p = lane % ACH, b_r = lane / BCH, b_p = lane % BCH;
    const int dl = lane & (NROW - 1);
    const int drow = ((dl / A_RPI) * 4 + wave) * A_RPI + (dl % A_RPI);  // row (inside a chunk) this lane decodes
    // byte offset inside the input patch of this lane's k-chunk (row parity: A_RPI == 1 -> parity of j = wave & 1)
    uint32_t a_tapb;
    {
        const int par = A_RPI == 1 ? (wave & 1) : (a_r & 1);
        int k = k0row + ((a_p ^ (par << 3)) << 2);
        k = k < K ? k : 0;  // weight rows past K are never stored
        a_tapb = (uint32_t)tap_offset<false>(g, (uint32_t)k) << 2;
    }
    uint32_t b_offb[BI];  // byte offset of (row inside the chunk, column chunk) inside dY, per instruction
    int b_row[BI];
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        b_row[i] = (i * 4 + wave) * B_RPI + b_r;
        int n = n0 + ((b_p ^ ((b_r & 1) << 3)) << 2);
        n = n < N ? n : 0;  // columns past N are never stored
        b_offb[i] = (uint32_t)(b_row[i] * N + n) << 2;
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int a_ = 0; a_ < TM; ++a_)
#pragma unroll
        for (int b_ = 0; b_ < TN; ++b_)
#pragma unroll
            for (int r_ = 0; r_ < 16; ++r_) acc[a_][b_][r_] = 0.f;

    const char *inb = reinterpret_cast<const char *>(in);
    auto issue = [&](int64_t mc, int stage) {
        float *sa = lds + stage * STAGE, *sb = sa + BK * 32;
        int64_t m = mc + drow;
        m = m < mend ? m : mend - 1;  // annihilated by the zero dY row below
        const uint32_t smp = fdiv((uint32_t)m, g.dOHOW), pix = (uint32_t)m - smp * (uint32_t)(g.OH * g.OW);
        const int rowb = (int)((smp * (uint32_t)in_stride + (uint32_t)patch_origin<false>(g, pix)) << 2);
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const uint32_t ob = (uint32_t)__builtin_amdgcn_ds_bpermute((i * A_RPI + a_r) << 2, rowb) + a_tapb;
            GLDS16(reinterpret_cast<const float *>(inb + (size_t)ob), sa + (i * 4 + wave) * 256);
        }
        const char *dyb = reinterpret_cast<const char *>(dy + mc * N);
        const int left = (int)min((int64_t)32, mend - mc);  // rows of this chunk that exist
#pragma unroll
        for (int i = 0; i < BI; ++i) {
            const float *src = b_row[i] < left ? reinterpret_cast<const float *>(dyb + (size_t)b_offb[i]) : sf_zero_page;
            GLDS16(src, sb + (i * 4 + wave) * 256);
        }
    };
    const bool do_colsum = partial_b != nullptr && bx == 0 && tid < BN;
    float colacc = 0.f;
    int stage = 0;
    int64_t mc0 = mbeg;
    if constexpr (ZL) {
        // ---- full 32-row chunks of a linear layer; a ragged last chunk (and nothing else) takes the general loop below
        const int64_t nfull = (mend - mbeg) / 32;
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)lds;
        uint32_t avoff[AI], bvoff[BI];
#pragma unroll
        for (int i = 0; i < AI; ++i)  // row (inside a chunk) of instruction i = ((i*4 + wave)*A_RPI + a_r); linear layer: origin 0
            avoff[i] = (uint32_t)(((i * 4 + wave) * A_RPI + a_r) * (int)in_stride * 4) + a_tapb;
#pragma unroll
        for (int i = 0; i < BI; ++i) bvoff[i] = b_offb[i];
        const int i_ = lane & 31, h_ = lane >> 5;
        const int acol = ((((wm * TM * 32 + TM * i_) >> 2) ^ (h_ << 3)) << 2) | ((TM * i_) & 3);
        const int bcol = ((((wn * TN * 32 + TN * i_) >> 2) ^ (h_ << 3)) << 2) | ((TN * i_) & 3);
        const float *ap0 = lds + h_ * BK + acol, *bp0 = lds + BK * 32 + h_ * BN + bcol;  // stage 0
        auto dma = [&](int64_t mc, int st) {
            const char *ab = reinterpret_cast<const char *>(in) + mc * in_stride * 4;
            const char *bb = reinterpret_cast<const char *>(dy + mc * N);
#pragma unroll
            for (int i = 0; i < AI; ++i) glds16_s(ab, avoff[i], lds0 + (uint32_t)((st * STAGE + (i * 4 + wave) * 256) * 4));
#pragma unroll
            for (int i = 0; i < BI; ++i)
                glds16_s(bb, bvoff[i], lds0 + (uint32_t)((st * STAGE + BK * 32 + (i * 4 + wave) * 256) * 4));
        };
        auto chunk = [&](int64_t mc, int64_t mlast, auto stc) {  // multiply chunk mc out of stage ST, stream mc + 32 into ST ^ 1
            constexpr int ST = decltype(stc)::value;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (mc + 32 < mlast) dma(mc + 32, ST ^ 1);
            if (do_colsum) {
                const float *sb = lds + ST * STAGE + BK * 32;
#pragma unroll 8
                for (int kk = 0; kk < 32; ++kk) colacc += sb[kk * BN + ((((tid >> 2) ^ ((kk & 1) << 3)) << 2) | (tid & 3))];
            }
#pragma unroll
            for (int s_ = 0; s_ < 16; ++s_) {
                const typename VecW<TM>::T a = *reinterpret_cast<const typename VecW<TM>::T *>(ap0 + ST * STAGE + 2 * s_ * BK);
                const typename VecW<TN>::T b = *reinterpret_cast<const typename VecW<TN>::T *>(bp0 + ST * STAGE + 2 * s_ * BN);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(vec_elem<TM>(a, tm), vec_elem<TN>(b, tn),
                                                                           acc[tm][tn], 0, 0, 0);
            }
        };
        const int64_t mlast = mbeg + nfull * 32;  // end of the full chunks
        if (nfull > 0) dma(mbeg, 0);
        int64_t mc = mbeg;
        for (; mc + 64 <= mlast; mc += 64) {
            chunk(mc, mlast, std::integral_constant<int, 0>{});
            chunk(mc + 32, mlast, std::integral_constant<int, 1>{});
        }
        if (mc < mlast) {
            chunk(mc, mlast, std::integral_constant<int, 0>{});
            mc += 32;
        }
        mc0 = mc;  // == mlast
        if (mc0 < mend) {  // ragged tail: every wave is past its last read of both stages only after a barrier
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    if (mc0 < mend) issue(mc0, 0);
    for (int64_t mc = mc0; mc < mend; mc += 32, stage ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (mc + 32 < mend) issue(mc + 32, stage ^ 1);
        const float *sa = lds + stage * STAGE, *sb = sa + BK * 32;
        if (do_colsum) {
#pragma unroll 8
            for (int kk = 0; kk < 32; ++kk) colacc += sb[kk * BN + ((((tid >> 2) ^ ((kk & 1) << 3)) << 2) | (tid & 3))];
        }
        // lane (i, h) reads words TM*i.. of A row 2s+h and TN*i.. of dY row 2s+h; odd rows: halves swapped
        const int i = lane & 31, h = lane >> 5;
        const int acol = ((((wm * TM * 32 + TM * i) >> 2) ^ (h << 3)) << 2) | ((TM * i) & 3);
        const int bcol = ((((wn * TN * 32 + TN * i) >> 2) ^ (h << 3)) << 2) | ((TN * i) & 3);
        const float *ap = sa + h * BK + acol, *bp = sb + h * BN + bcol;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const typename VecW<TM>::T a = *reinterpret_cast<const typename VecW<TM>::T *>(ap + 2 * s * BK);
            const typename VecW<TN>::T b = *reinterpret_cast<const typename VecW<TN>::T *>(bp + 2 * s * BN);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(vec_elem<TM>(a, tm), vec_elem<TN>(b, tn),
                                                                       acc[tm][tn], 0, 0, 0);
        }
    }
    if (do_colsum && n0 + tid < N) partial_b[(int64_t)bz * N + n0 + tid] = colacc;
    float *dst = partial + (int64_t)bz * K * N;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = n0 + wn * TN * 32 + TN * (lane & 31) + tn;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0row + wm * TM * 32 + TM * FRAG_ROW(r, lane) + tm;
                if (k < K && n < N) dst[(int64_t)k * N + n] = acc[tm][tn][r];
            }
        }
}

template <int BK, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void k_wgrad_glds(ConvG g, const float *__restrict__ in, int64_t in_stride,
                                                    const float *__restrict__ dy, float *__restrict__ partial,
                                                    float *__restrict__ partial_b, int64_t Mtot, int64_t m_per_split,
                                                    int rx, int ry, int rtot) {
    wgrad_glds_body<BK, BN, WM, WN, false>(g, in, in_stride, dy, partial, partial_b, Mtot, m_per_split, rx, ry, rtot);
}
template <int BK, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void k_wgrad_glds_z(ConvG g, const float *__restrict__ in, int64_t in_stride,
                                                      const float *__restrict__ dy, float *__restrict__ partial,
                                                      float *__restrict__ partial_b, int64_t Mtot, int64_t m_per_split,
                                                      int rx, int ry, int rtot) {
    wgrad_glds_body<BK, BN, WM, WN, true>(g, in, in_stride, dy, partial, partial_b, Mtot, m_per_split, rx, ry, rtot);
}

// ============================================================================================== FORWARD, raw u8 frames
// First conv layer on raw NCHW u8 observations (Nature-CNN conv1: 4x84x84 -> 32, 8x8 stride 4, K = 256).
// Measured on MI355X (tools/ubench/mfma_peak.hip): f32 MFMA alone sustains 153-155 TFLOP/s, but every VALU
// instruction issued next to it costs ~3-4 cycles of matrix-pipe time (3 VALU per 16x16x4 MFMA: 118 TFLOP/s).  The
// im2col kernel converts every input byte u8 -> f32 KH*KW/S^2 = 4 times (cvt + scale per MFMA operand) and re-gathers
// it 4 times through the vector-memory path; a version that kept the frames as BYTES in LDS and converted at
// fragment-read time hit the same 90 TFLOP/s wall (2 VALU per MFMA).  So: convert ONCE.
//   A block owns SMP = 2 samples and walks the image in strips of R = 4 output rows (20 input rows).  The strip's
//   bytes are loaded with plain coalesced 4-byte loads (prefetched into registers during the previous strip's MFMAs),
//   converted ((x - mean) * 1/scale, the arithmetic of k_conv_fwd's loader) and stored to LDS as f32
//   [smp][c][20][W].  The MFMA A operand is then read straight out of that image: lane (row = output pixel, kg)
//   fetches the 4 floats (c, kh, kw..kw+3) of its patch with ONE ds_read_b128 and feeds 4 MFMAs — the k-loop contains
//   no VALU work besides the address add, and no global memory traffic at all.
//   The weight fragment of the wave's 16 output channels lives in 64 VGPRs for the whole kernel.
// v_mfma_f32_16x16x4_f32 (same peak as 32x32x2) gives 16-row fragments: SMP * R * OW = 160 rows = 10 fragments, split
// evenly over 2 (row) x 2 (16-channel column) waves = TMF = 5 per wave and strip — no ragged tile anywhere.
// Reduction order inside a 16-group: MFMA j consumes k = 16g + 4*kg + j from lane group kg (A and B agree on it).
typedef float f32x4 __attribute__((ext_vector_type(4)));

// NORM (sf_conv_fwd_norm): cfg.normalize_input=True on raw frames — the running-statistics normalisation of
// utils/normalize.py:51-70 + running_mean_std.py:79-110, v = clamp(((x - sub) * 1/scale - mu[d]) * rstd[d], +-5), happens
// HERE, where the byte becomes an f32 in LDS, instead of in a pass of its own that writes (and conv1 then re-reads) a
// 113 KB f32 copy of every 28 KB frame.  d = the byte's NCHW offset inside the frame, mu / rstd = the normaliser's f32
// tables [Cin*H*W] (g.nmu, g.nrstd).  The table words of a strip depend on the strip only, so with NORM the work-group
// walks its units STRIP-major (all its sample pairs at strip 0, then strip 1, ...) and keeps the strip's table words
// in registers: the tables are fetched 5 times per work-group and launch, not once per unit.
template <int SMP, int R, int TMF, int KG, bool SUB, bool NORM>
__device__ __forceinline__ void conv_u8_img_body(ConvG g, const uint8_t *__restrict__ in, int64_t in_stride,
                                                 const int32_t *__restrict__ index, int64_t offset,
                                                 const float *__restrict__ w, const float *__restrict__ bias,
                                                 float *__restrict__ out, int nsamples) {
    extern __shared__ __attribute__((aligned(16))) float strip[];  // [SMP][Cin][RS][W] f32
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: LDS-DMA bases (M0) stay on the scalar unit
    const int wm = wave >> 1, wn = wave & 1;
    // Geometry is compile-time (the launcher checks it): the 80 fragment addresses of a strip then differ from 5
    // per-lane bases only by IMMEDIATE ds_read offsets; with run-time geometry hipcc materialised all 80 in VGPRs
    // (256 registers, one wave per SIMD).
    constexpr int H = 84, W = 84, Cin = 4, KH = 8, KW = 8, S = 4, OH = 20, OW = 20, OHOW = OH * OW;
    static_assert(KG * 16 == Cin * KH * KW && SMP * R * OW == 32 * TMF && OH % R == 0 && (R * OW) % 16 == 0,
                  "Nature-CNN conv1 geometry");
    const int N = g.Cout;
    constexpr int RS = (R - 1) * S + KH;  // input rows per strip
    constexpr int W4 = W >> 2;            // 4-byte words per input row
    const int words = Cin * RS * W4;     // u32 words per strip and sample
    constexpr int NLD = 7;               // u32 loads per thread, sample and strip (launcher: words <= 256 * NLD)
    // PERSISTENT work-groups: block b takes the sample pairs b, b + gridDim.x, ... — the 64 weight registers, the tap
    // table and the strip decode below are set up once per work-group instead of once per pair, and the byte prefetch
    // runs across pair boundaries.
    constexpr int nstrips = OH / R;
    const int npairs = (nsamples + SMP - 1) / SMP;
    const int my_pairs = (int)blockIdx.x < npairs ? (npairs - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int total_units = my_pairs * nstrips;  // unit = (local pair, strip)
    if (total_units == 0) return;
    // ---- per-thread strip words: word q -> (c, row, x4), the same for every sample of the block; the sample bases
    // are wave-uniform, so a load is "SGPR base + 32-bit VGPR offset"
    int gofs[NLD], lofs[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        int q = tid + 256 * i;
        const bool ok = q < words;
        q = ok ? q : 0;
        const int x4 = q % W4, t1 = q / W4, row = t1 % RS, c = t1 / RS;
        gofs[i] = (c * H + row) * W + x4 * 4;
        lofs[i] = ok ? (c * RS + row) * W + x4 * 4 : -1;
    }
    uint32_t pre[SMP][NLD];
    f32x4 tmu[NORM ? NLD : 1], trs[NORM ? NLD : 1];  // NORM: this strip's table words of the thread's NLD image words
    auto decode = [&](int unit, int &lp, int &st) {  // unit -> (local pair, strip): pair-major, NORM strip-major
        if (NORM) { st = unit / my_pairs; lp = unit - st * my_pairs; }
        else { lp = unit / nstrips; st = unit - lp * nstrips; }
    };
    auto load_strip = [&](int unit) {
        int lp, st;
        decode(unit, lp, st);
        const int s0u = ((int)blockIdx.x + lp * (int)gridDim.x) * SMP;
        const int rowoff = st * R * S * W;
#pragma unroll
        for (int z = 0; z < SMP; ++z) {
            int sg = s0u + z;
            sg = sg < nsamples ? sg : nsamples - 1;
            const uint8_t *sb = in + sample_base(g, index, offset, in_stride, (uint32_t)sg);  // wave-uniform
#pragma unroll
            for (int i = 0; i < NLD; ++i) pre[z][i] = *reinterpret_cast<const uint32_t *>(sb + rowoff + gofs[i]);
        }
        if (NORM && lp == 0) {  // first unit of a strip (the previous strip's last store_strip is over)
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                tmu[i] = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(g.nmu + rowoff + gofs[i], 16));
                trs[i] = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(g.nrstd + rowoff + gofs[i], 16));
            }
        }
    };
    const float sub = g.sub_mean, scl = g.inv_scale;
    auto store_strip = [&]() {
#pragma unroll
        for (int z = 0; z < SMP; ++z)
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float b8 = (float)((pre[z][i] >> (8 * j)) & 0xFFu);
                    v[j] = (SUB ? b8 - sub : b8) * scl;  // SUB = false: mean == 0 (x - 0 is exact anyway)
                    if (NORM) v[j] = clampf((v[j] - tmu[i][j]) * trs[i][j], -5.0f, 5.0f);  // k_obsnorm_apply's arithmetic
                }
                if (lofs[i] >= 0)
                    *reinterpret_cast<f32x4 *>(__builtin_assume_aligned(strip + z * Cin * RS * W + lofs[i], 16)) = v;
            }
    };
    load_strip(0);
    // ---- weight fragment of this wave's 16 output channels -> registers; tap offsets of this lane's kg slot
    const int col = wn * 16 + (lane & 15), kg = lane >> 4;
    const int colc = col < N ? col : N - 1;
    f32x4 breg[KG];
    int tapv[KG];
#pragma unroll
    for (int gi = 0; gi < KG; ++gi) {
        const int k = 16 * gi + 4 * kg;  // k = (c*KH + kh)*KW + kw
#pragma unroll
        for (int j = 0; j < 4; ++j) breg[gi][j] = w[(int64_t)(k + j) * N + colc];
        // 16*gi advances (c, kh) by whole rows, 4*kg stays inside two kw-rows: tap = uniform(gi) + lane(kg)
        tapv[gi] = ((16 * gi) / (KH * KW) * RS + ((16 * gi) % (KH * KW)) / KW) * W;
    }
    const int lanetap = (kg >> 1) * W + (kg & 1) * 4;
    const float bv = bias ? bias[colc] : 0.f;
    // fragment rows of this wave inside a strip: row = (smp, oh_l, ow), fixed for all strips
    int origin[TMF];
#pragma unroll
    for (int t = 0; t < TMF; ++t) {
        const int row = (2 * t + wm) * 16 + (lane & 15);  // < SMP * R * OW by construction
        const int smp = row / (R * OW), p = row - smp * (R * OW), ohl = p / OW, ow = p - ohl * OW;
        origin[t] = (smp * Cin * RS + ohl * S) * W + ow * S + lanetap;
    }
    const uint32_t voff = (uint32_t)(4 * (lane >> 4)) * (uint32_t)N + (uint32_t)col;
    for (int unit = 0; unit < total_units; ++unit) {
        int lp, st;
        decode(unit, lp, st);
        const int s0 = ((int)blockIdx.x + lp * (int)gridDim.x) * SMP;
        store_strip();  // first use of the prefetched bytes
        __syncthreads();
        if (unit + 1 < total_units) load_strip(unit + 1);  // lands during the MFMA phase
        f32x4 acc[TMF];
#pragma unroll
        for (int t = 0; t < TMF; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        // fragments double-buffered one 16-group ahead; sched_barrier keeps hipcc from hoisting ALL 80 reads (320
        // VGPRs) to the top of the strip
        f32x4 a[2][TMF];
        auto fetch = [&](int gi) {
#pragma unroll
            for (int t = 0; t < TMF; ++t)
                a[gi & 1][t] = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(strip + origin[t] + tapv[gi], 16));
        };
        fetch(0);
#pragma unroll
        for (int gi = 0; gi < KG; ++gi) {
            if (gi + 1 < KG) fetch(gi + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < TMF; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[gi & 1][t][j], breg[gi][j], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // epilogue: R*OW is a multiple of 16, so a fragment never straddles a sample -> uniform base pointer per
        // fragment, one 32-bit lane offset, activation kind hoisted (see store_fwd_tile)
        auto epilogue = [&](auto kc) {
            constexpr int KIND = decltype(kc)::value;
#pragma unroll
            for (int t = 0; t < TMF; ++t) {
                const int f = 2 * t + wm, smp = f / (R * OW / 16), prow = f * 16 - smp * (R * OW);
                float *ob = out + ((int64_t)(s0 + smp) * OHOW + st * (R * OW) + prow) * N;
                const bool ok = s0 + smp < nsamples && col < N;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = act_fwd_c<KIND>(acc[t][r] + bv, g.relu);
                    if (ok) (ob + r * N)[voff] = v;
                }
            }
        };
        if (g.relu == 1) epilogue(std::integral_constant<int, 1>{});
        else epilogue(std::integral_constant<int, -1>{});
        __syncthreads();  // everybody is done reading this strip before it is overwritten
    }
}

template <int SMP, int R, int TMF, int KG, bool SUB>
__global__ __launch_bounds__(256) void k_conv_u8_img(ConvG g, const uint8_t *__restrict__ in, int64_t in_stride,
                                                     const int32_t *__restrict__ index, int64_t offset,
                                                     const float *__restrict__ w, const float *__restrict__ bias,
                                                     float *__restrict__ out, int nsamples) {
    conv_u8_img_body<SMP, R, TMF, KG, SUB, false>(g, in, in_stride, index, offset, w, bias, out, nsamples);
}
template <int SMP, int R, int TMF, int KG>
__global__ __launch_bounds__(256) void k_conv_u8_img_norm(ConvG g, const uint8_t *__restrict__ in, int64_t in_stride,
                                                          const int32_t *__restrict__ index, int64_t offset,
                                                          const float *__restrict__ w, const float *__restrict__ bias,
                                                          float *__restrict__ out, int nsamples) {
    conv_u8_img_body<SMP, R, TMF, KG, true, true>(g, in, in_stride, index, offset, w, bias, out, nsamples);
}

// ============================================================================================== DATA GRADIENT, stride groups
// Strided convolutions (conv2: 4x4 stride 2, Cin = 32): with one input pixel per tile the GEMM is only N = Cin = 32
// columns wide and K = 2x2 taps x Cout deep, so twice the operand traffic per MFMA of an N = 64 tile and a 8-chunk
// reduction per epilogue.  But the S x S input pixels (ihc*S + ph, iwc*S + pw) all read the SAME (KH/S) x (KW/S) block
// of dY pixels (ihc - a, iwc - b) — only the filter tap (ph + S*a, pw + S*b) differs.  So per S x S pixel group:
//   A[row][(a, b, co)]         = dY[s, ihc - a, iwc - b, co]
//   B[(ph, pw, c)][(a, b, co)] = W[((ph + S*a)*KW + pw + S*b)*Cin + c][co]         N = S*S*Cin = 128, K = 256 (conv2)
// i.e. every filter tap is used exactly once per group: a dense N = 128, K = 256 GEMM with 64 x 64 wave tiles.  Same
// DMA pipeline and swizzle as k_dgrad_pix.
//
// Tiling.  A first version tiled BM SAMPLES at one pixel group (like k_dgrad_pix); its epilogue then touches BM
// scattered 256-byte pieces (one per sample, 50 KB apart) and was no faster.  Here a tile is 128 consecutive rows
// m = (sample, iwc) — about 13 whole image rows of groups — and the block walks DOWN the image (ihc = 0..Hg-1): per
// step every sample of the tile reads/writes two complete contiguous image rows (2 x 2.5 KB for conv2), dY rows are
// contiguous 256-byte neighbours, and the per-row (sample, iwc) decode is done once per block (row offsets parked in
// LDS).  (a, b) blocks outside dY: a is uniform per step and skipped; b depends on iwc, i.e. on the row, so a row whose
// tap column falls outside dY takes its DMA from the zero page (10 % of the rows x taps for conv2 — the only
// structural-zero work left).  Measured (n = 32768, conv2): class-decomposed im2col kernel 2.98 ms, pixel-major
// 2.51 ms, this kernel 2.33 ms; with all memory traffic removed 1.65 ms — the epilogue's 64 loads + 64 stores per lane
// and step, not the matrix pipe, are what is left.
template <int BM, int BN, int WM, int WN, bool ZL>
__device__ __forceinline__ void dgrad_quadrow_body(ConvG g, const float *__restrict__ dy,
                                                   const float *__restrict__ w, const float *__restrict__ in_act,
                                                   float *__restrict__ din, int64_t Mrows, FastDiv dWg) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int AI = BM / 32, BI = BN / 32;
    constexpr int STAGE = (BM + BN) * 32;
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 waves per block");
    __shared__ __attribute__((aligned(1024))) float lds[2 * STAGE];
    __shared__ uint32_t rowoff[BM];  // element offset of (sample, iwc) inside din / in_act, 0xFFFFFFFF = row past M
    __shared__ int rot_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: LDS-DMA bases (M0) stay on the scalar unit
    const int wm = wave / WN, wn = wave % WN;
    // SF_QUADROW_ROT: the two work-groups that share a CU walk the group rows in different rotations (0, 1, .., Hg-1 and
    // 1, .., Hg-1, 0): the first and the last row have half the taps, so the second one's output stores fall half a step
    // behind the first one's for the whole tile, at the same total time.  The key is the hardware wave slot of wave 0
    // (HW_ID.wave_id: with two waves per SIMD the co-resident work-groups sit in slots of different parity, and a
    // work-group that replaces a finished one inherits its slot) -- a key from blockIdx cannot know who shares a CU.
    if (ZL && SF_QUADROW_ROT && tid == 0) rot_s = (int)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4) & 1u);
    const int Cin = g.Cin, Cout = g.Cout, S = g.S, OH = g.OH, OW = g.OW, H = g.H, W = g.W;
    const int Hg = (H + S - 1) / S, Wg = (int)dWg.d, N = S * S * Cin;
    const int KHs = g.KH / S, KWs = g.KW / S, CC = Cout >> 5;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const uint32_t sstride = (uint32_t)(H * W * Cin);
    if (tid < BM) {
        const int64_t m = m0 + tid;
        const uint32_t mm = m < Mrows ? (uint32_t)m : 0u;
        const uint32_t s = fdiv(mm, dWg), iwc = mm - s * (uint32_t)Wg;
        // ZL: BYTE offsets (the launcher guarantees n*H*W*Cin*4 < 2^32), so that "uniform base + 32-bit lane offset" stores and
        // loads need one v_add_u32 per element instead of a 64-bit address build-up
        rowoff[tid] = m < Mrows ? (s * sstride + iwc * (uint32_t)(S * Cin)) * (ZL ? 4u : 1u) : 0xFFFFFFFFu;
    }
    const int lrow = lane >> 3, lpos = lane & 7;
    const float *asrc[AI], *bsrc[BI];
    int aiwc[AI];
#pragma unroll
    for (int i = 0; i < AI; ++i) {
        const int row = (i * 4 + wave) * 8 + lrow;
        int64_t m = m0 + row;
        m = m < Mrows ? m : Mrows - 1;
        const uint32_t s = fdiv((uint32_t)m, dWg);
        aiwc[i] = (int)((uint32_t)m - s * (uint32_t)Wg);
        asrc[i] = dy + ((int64_t)s * (OH * OW) + aiwc[i]) * Cout + ((lpos ^ ((row >> 1) & 7)) << 2);
    }
    const int zpos = lpos << 2;  // any 16 bytes of the zero page
#pragma unroll
    for (int i = 0; i < BI; ++i) {
        const int row = (i * 4 + wave) * 8 + lrow;
        int n = n0 + row;
        n = n < N ? n : N - 1;
        const int pp = (int)fdiv((uint32_t)n, g.dCin), c = n - pp * Cin, ph = pp / S, pw = pp - ph * S;
        bsrc[i] = w + (int64_t)((ph * g.KW + pw) * Cin + c) * Cout + ((lpos ^ ((row >> 1) & 7)) << 2);
    }
    f32x16 acc[TM][TN];
    // column fragment tn of this lane = (pixel (ph, pw) of the group, channel c)
    uint32_t cbase[TN];
    bool colok[TN];
    int phv[TN], pwv[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + wn * TN * 32 + tn * 32 + (lane & 31);
        const int nc = n < N ? n : 0;
        const int pp = (int)fdiv((uint32_t)nc, g.dCin), c = nc - pp * Cin;
        phv[tn] = pp / S;
        pwv[tn] = pp - phv[tn] * S;
        colok[tn] = n < N;
        cbase[tn] = (uint32_t)((phv[tn] * W + pwv[tn]) * Cin + c) * (ZL ? 4u : 1u);
    }
    const int rbase = wm * TM * 32 + 4 * (lane >> 5);
    float actv[TM][TN][16];
    __syncthreads();  // rowoff visible
    // Complete tiles (all rows real, all columns real, no group row hanging over the image) skip the per-element
    // predicates; the activation kind is hoisted out of the element loops.
    const bool full = m0 + BM <= Mrows && n0 + BN <= N && H % S == 0;
    auto ld32 = [&](const float *base, uint32_t off) {  // ZL: off in bytes; else in elements
        if constexpr (ZL) return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + (size_t)off);
        else return base[off];
    };
    auto st32 = [&](float *base, uint32_t off, float v) {
        if constexpr (ZL) *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + (size_t)off) = v;
        else base[off] = v;
    };
    auto prefetch_act = [&](int ihc) {
        if (!in_act) return;
        const uint32_t gy = (uint32_t)(ihc * S * W * Cin) * (ZL ? 4u : 1u);
        if (full) {
            uint32_t cb[TN];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) cb[tn] = gy + cbase[tn];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t ro = rowoff[rbase + tm * 32 + (r & 3) + 8 * (r >> 2)];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) actv[tm][tn][r] = ld32(in_act, ro + cb[tn]);
                }
            return;
        }
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const uint32_t ro = rowoff[rbase + tm * 32 + (r & 3) + 8 * (r >> 2)];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const bool ok = ro != 0xFFFFFFFFu && colok[tn] && ihc * S + phv[tn] < H;
                    actv[tm][tn][r] = ld32(in_act, ok ? ro + gy + cbase[tn] : 0u);
                }
            }
    };
    auto store_step = [&](int ihc, bool zero) {
        const uint32_t gy = (uint32_t)(ihc * S * W * Cin) * (ZL ? 4u : 1u);
        const int akind = g.relu;
        auto body = [&](auto kc, auto fc) {
            constexpr int KIND = decltype(kc)::value;
            constexpr bool FULL = decltype(fc)::value;
            uint32_t cb[TN];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) cb[tn] = gy + cbase[tn];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const uint32_t ro = rowoff[rbase + tm * 32 + (r & 3) + 8 * (r >> 2)];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        if (FULL || (ro != 0xFFFFFFFFu && colok[tn] && ihc * S + phv[tn] < H)) {
                            float v = zero ? 0.f : acc[tm][tn][r];
                            if (!zero) v = act_bwd_mul<KIND>(v, actv[tm][tn][r], akind);
                            if (!(ZL && (SF_GLDS_ABLATE & 32)) || v == 1.2345e-30f) st32(din, ro + cb[tn], v);  // (ablation: the compare keeps the MFMAs alive)
                        }
                }
        };
        if (!in_act || zero) {
            if (full) body(std::integral_constant<int, 0>{}, std::true_type{});
            else body(std::integral_constant<int, 0>{}, std::false_type{});
        } else if (akind == 1) {
            if (full) body(std::integral_constant<int, 1>{}, std::true_type{});
            else body(std::integral_constant<int, 1>{}, std::false_type{});
        } else {
            body(std::integral_constant<int, -1>{}, std::false_type{});
        }
    };
    // (the last group column of an image whose width is not a multiple of S has pixels past W: never the case for the
    // launcher's contract W % S == 0)
    // (All work-groups of a launch have the same work and the resident ones start together; the ablation of
    // profiles/r05_aa_quadrow_ablation.log — time without MFMAs 565 us + MFMA time 1219 us = the kernel's 1784 us — suggested
    // they run in lock-step with nothing overlapping.  Walking the group rows rotated by one in half of the work-groups (the
    // first and last group row have half the taps: half a step of phase shift for free) changed nothing, 1654 vs 1654 us:
    // profiles/r05_ab_quadrow_rotation.log.)
    // v = position in the walk, ihc = the group row visited there (v + rot, wrapping; v >= Hg: past the end)
    const int rot = (ZL && SF_QUADROW_ROT && Hg > 1) ? __builtin_amdgcn_readfirstlane(rot_s) : 0;
    struct St { int v, ihc, a_lo, na, total; };
    auto step = [&](int v) {
        St p;
        p.v = v;
        const int ihc = v + rot < Hg ? v + rot : v + rot - Hg;
        p.ihc = ihc;
        p.a_lo = ihc - OH + 1 > 0 ? ihc - OH + 1 : 0;
        const int a_hi = ihc < KHs - 1 ? ihc : KHs - 1;
        p.na = a_hi - p.a_lo + 1;
        p.na = p.na > 0 ? p.na : 0;
        p.total = p.na * KWs * CC;
        return p;
    };
    auto next_step = [&](int v) {  // first position >= v with work (v = Hg: none)
        St p = step(v);
        while (p.v < Hg && p.total == 0) p = step(p.v + 1);
        return p;
    };
    auto row_at = [&](int v) { return v + rot < Hg ? v + rot : v + rot - Hg; };
    // ZL (k_dgrad_quadrow_z): fragments through per-lane LDS pointers with the stage as an immediate, the weight operand's
    // DMA as uniform base + 32-bit lane offset.  (The dY operand keeps 64-bit lane addresses: a lane whose tap column is
    // outside dY reads the zero page, which no 32-bit offset from dY can name.)
    uint32_t bvoff[BI];
    const float *apl[4], *bpl[4];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float *)lds;
    if constexpr (ZL) {
#pragma unroll
        for (int i = 0; i < BI; ++i) bvoff[i] = (uint32_t)((bsrc[i] - w) * (int64_t)sizeof(float));
        const int r = lane & 31, h = lane >> 5, sw = (r >> 1) & 7;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int pos = (((2 * c + h) ^ sw) << 2);
            apl[c] = lds + (wm * TM * 32 + r) * 32 + pos;
            bpl[c] = lds + BM * 32 + (wn * TN * 32 + r) * 32 + pos;
        }
    }
    auto issue = [&](const St &p, int q, int stage) {
        if (ZL && (SF_GLDS_ABLATE & 16)) return;
        const int blk = q / CC, cc = q - blk * CC;
        const int a = p.a_lo + blk / KWs, b = blk % KWs;
        const int64_t aoff = (int64_t)((p.ihc - a) * OW - b) * Cout + cc * 32;
        const int64_t boff = (int64_t)((S * a * g.KW + S * b) * Cin) * Cout + cc * 32;
        float *sa = lds + stage * STAGE, *sb = sa + BM * 32;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int ow = aiwc[i] - b;
            const float *src = (ow >= 0 && ow < OW) ? asrc[i] + aoff : sf_zero_page + zpos;
            GLDS16(src, sa + (i * 4 + wave) * 256);
        }
        if constexpr (ZL) {
            const float *bb = w + boff;
#pragma unroll
            for (int i = 0; i < BI; ++i)
                glds16_s(bb, bvoff[i], lds0 + (uint32_t)((stage * STAGE + BM * 32 + (i * 4 + wave) * 256) * 4));
            return;
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) GLDS16(bsrc[i] + boff, sb + (i * 4 + wave) * 256);
    };
    // ZL + SF_QUADROW_PREP: the addresses of a chunk's DMA are worked out ONE CHUNK AHEAD (scalar divisions of the chunk
    // index, the per-row zero-page select and its 64-bit adds: ~40 scalar + ~30 vector instructions that otherwise sit
    // between the barrier and the DMA with no MFMA of this wave in flight), so that after the barrier only the 8 DMA
    // instructions are left; the work for the chunk after next is placed behind them, among the MFMAs.
    struct It { St st; int q; };
    auto advance = [&](It &it) {
        if (it.q + 1 < it.st.total) ++it.q;
        else { it.st = next_step(it.st.v + 1); it.q = 0; }
    };
    struct Prep { const float *bb; const float *as[AI]; bool valid; };
    auto prep = [&](const It &it) {
        Prep r;
        r.valid = it.st.v < Hg;
        const int q = r.valid ? it.q : 0;
        const int blk = q / CC, cc = q - blk * CC;
        const int a = it.st.a_lo + blk / KWs, b = blk % KWs;
        const int64_t aoff = (int64_t)((it.st.ihc - a) * OW - b) * Cout + cc * 32;
        r.bb = w + (int64_t)((S * a * g.KW + S * b) * Cin) * Cout + cc * 32;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
            const int ow = aiwc[i] - b;
            r.as[i] = (ow >= 0 && ow < OW) ? asrc[i] + aoff : sf_zero_page + zpos;
        }
        return r;
    };
    auto fire = [&](const Prep &r, int stage, int part = 3) {  // part: 1 = the dY rows, 2 = the weight rows, 3 = both
        if (!r.valid || (SF_GLDS_ABLATE & 16)) return;
        float *sa = lds + stage * STAGE;
        if (part & 1) {
#pragma unroll
            for (int i = 0; i < AI; ++i) GLDS16(r.as[i], sa + (i * 4 + wave) * 256);
        }
        if (part & 2) {
#pragma unroll
            for (int i = 0; i < BI; ++i)
                glds16_s(r.bb, bvoff[i], lds0 + (uint32_t)((stage * STAGE + BM * 32 + (i * 4 + wave) * 256) * 4));
        }
    };
    St cur = next_step(0);
    for (int z = 0; z < cur.v && z < Hg; ++z) store_step(row_at(z), true);
    It ahead{cur, 0};
    Prep pend;
    if constexpr (ZL && SF_QUADROW_PREP) {
        fire(prep(ahead), 0);
        advance(ahead);
        pend = prep(ahead);
    } else if (cur.v < Hg) issue(cur, 0, 0);
    int stage = 0;
    while (cur.v < Hg) {
        const St nx = next_step(cur.v + 1);
        if constexpr (!ZL) {
#pragma unroll
            for (int a_ = 0; a_ < TM; ++a_)
#pragma unroll
                for (int b_ = 0; b_ < TN; ++b_)
#pragma unroll
                    for (int r_ = 0; r_ < 16; ++r_) acc[a_][b_][r_] = 0.f;
        }
        if constexpr (ZL) {  // CC even (launcher contract): a step starts in stage 0, two chunks per trip, static stages
            auto chunk = [&](int q, auto stc, auto firstc) {
                constexpr int ST = decltype(stc)::value;
                constexpr bool FIRST = decltype(firstc)::value;  // first chunk of the step: accumulation starts from the constant 0
                if (!(SF_GLDS_ABLATE & 64)) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                }
                auto refill = [&]() {
                    if constexpr (SF_QUADROW_PREP) {
                        fire(pend, ST ^ 1);  // the chunk after this one
                        advance(ahead);
                        pend = prep(ahead);  // ... and the addresses of the one after that
                    } else {
                        if (q + 1 < cur.total) issue(cur, q + 1, ST ^ 1);
                        else if (nx.v < Hg) issue(nx, 0, ST ^ 1);
                    }
                    if (FIRST) prefetch_act(cur.ihc);
                };
                if constexpr (SF_QUADROW_FRAG_DB == 2 && !(SF_GLDS_ABLATE & 128)) {
                    mma_chunk_ptrs_db<TM, TN, ST * STAGE, FIRST>(apl, bpl, acc, [&](auto pt) {
                        if constexpr (decltype(pt)::value == 0) refill();
                    });
                    return;
                }
                if constexpr (SF_QUADROW_FRAG_DB == 3 && SF_QUADROW_PREP && !(SF_GLDS_ABLATE & 128)) {
                    mma_chunk_ptrs_db<TM, TN, ST * STAGE, FIRST, true>(apl, bpl, acc, [&](auto pt) {
                        constexpr int P = decltype(pt)::value;
                        if constexpr (P == 0) { if (FIRST) prefetch_act(cur.ihc); }
                        if constexpr (P == 1) fire(pend, ST ^ 1, 1);
                        if constexpr (P == 2) fire(pend, ST ^ 1, 2);
                        if constexpr (P == 3) { advance(ahead); pend = prep(ahead); }
                    });
                    return;
                }
                refill();
                if constexpr (SF_QUADROW_FRAG_DB == 1 && !(SF_GLDS_ABLATE & 128)) {
                    mma_chunk_ptrs_db<TM, TN, ST * STAGE, FIRST>(apl, bpl, acc, [](auto) {});
                    return;
                }
                if (SF_GLDS_ABLATE & 128) {
                    if (FIRST) {
#pragma unroll
                        for (int a_ = 0; a_ < TM; ++a_)
#pragma unroll
                            for (int b_ = 0; b_ < TN; ++b_)
#pragma unroll
                                for (int r_ = 0; r_ < 16; ++r_) acc[a_][b_][r_] = 0.f;
                    }
                } else
                mma_chunk_ptrs<TM, TN, ST * STAGE, FIRST>(apl, bpl, acc);
            };
            chunk(0, std::integral_constant<int, 0>{}, std::true_type{});
            chunk(1, std::integral_constant<int, 1>{}, std::false_type{});
            for (int q = 2; q < cur.total; q += 2) {
                chunk(q, std::integral_constant<int, 0>{}, std::false_type{});
                chunk(q + 1, std::integral_constant<int, 1>{}, std::false_type{});
            }
        } else
        for (int q = 0; q < cur.total; ++q, stage ^= 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (q + 1 < cur.total) issue(cur, q + 1, stage ^ 1);
            else if (nx.v < Hg) issue(nx, 0, stage ^ 1);
            if (q == 0) prefetch_act(cur.ihc);
            const float *sa = lds + stage * STAGE;
            mma_chunk_rows<TM, TN>(sa, sa + BM * 32, wm * TM * 32, wn * TN * 32, lane, acc);
        }
        store_step(cur.ihc, false);
        for (int z = cur.v + 1; z < nx.v && z < Hg; ++z) store_step(row_at(z), true);
        cur = nx;
    }
}
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void k_dgrad_quadrow(ConvG g, const float *__restrict__ dy,
                                                         const float *__restrict__ w, const float *__restrict__ in_act,
                                                         float *__restrict__ din, int64_t Mrows, FastDiv dWg) {
    dgrad_quadrow_body<BM, BN, WM, WN, false>(g, dy, w, in_act, din, Mrows, dWg);
}
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void k_dgrad_quadrow_z(ConvG g, const float *__restrict__ dy,
                                                           const float *__restrict__ w, const float *__restrict__ in_act,
                                                           float *__restrict__ din, int64_t Mrows, FastDiv dWg) {
    dgrad_quadrow_body<BM, BN, WM, WN, true>(g, dy, w, in_act, din, Mrows, dWg);
}

// ============================================================================================== WEIGHT GRADIENT, raw u8 frames
// dW[(c, kh, kw)][n] = sum over (sample, oh, ow) of x[s, c, oh*S + kh, ow*S + kw] * dY[s, oh, ow, n]  for Nature-CNN conv1.
// Same strip image as k_conv_u8_img (bytes converted ONCE into an f32 LDS image; the im2col kernel converts every byte
// 4 times and spends 2 VALU per MFMA operand, which costs matrix-pipe time).  The reduction runs over output pixels,
// so the whole 256 x 32 gradient (8 KB per lane-set = 128 accumulator registers) stays resident in EVERY wave and
// each wave reduces its own quarter of the strip's pixels:
//   16x16x4 MFMA: 4 reduction indices per instruction = 4 consecutive output pixels m (lane group kg = lane >> 4).
//   A operand, lane (i, kg): ONE ds_read_b128 at image position (c, oh*S + i/2, ow*S + 4*(i%2)) of pixel m_kg gives
//     x for k = c*64 + 4i .. 4i+3 — four different weight ROWS.  Output rows are relabelled (tile j holds rows
//     c*64 + 4i + j) so that those four words feed four MFMAs on four accumulator tiles.
//   B operand, lane (n, kg): dY[m_kg][n] and dY[m_kg][n + 16] straight from a DMA'd copy of the strip's dY rows.
//   => per pixel quad: 4 ds_read_b128 + 2 ds_read_b32 feed 32 MFMAs, no VALU.
// Blocks are persistent over sample pairs; the four waves' accumulators are summed through LDS at the end and written
// as ONE partial per block (reduced by k_reduce_partials, deterministic).  The bias gradient (column sums of dY) rides
// along from the LDS copy of dY.
// NORM (sf_conv_wgrad_norm): x is the NORMALISED observation, formed from the byte and the normaliser's tables on the way
// into LDS exactly as in the forward kernel above (same strip-major unit order, same table registers).
template <int SMP, int R, bool SUB, bool NORM>
__device__ __forceinline__ void conv1_wgrad_img_body(ConvG g, const uint8_t *__restrict__ in, int64_t in_stride,
                                                     const int32_t *__restrict__ index, int64_t offset,
                                                     const float *__restrict__ dy, float *__restrict__ partial,
                                                     float *__restrict__ partial_b, int nsamples, int npairs) {
    constexpr int H = 84, W = 84, Cin = 4, KH = 8, KW = 8, S = 4, OH = 20, OW = 20, OHOW = OH * OW, N = 32, K = 256;
    constexpr int RS = (R - 1) * S + KH, W4 = W >> 2, NLD = 7;
    constexpr int ROWS = SMP * R * OW;       // output pixels per strip (160)
    constexpr int QW = ROWS / 4 / 4;         // pixel quads per wave and strip (10)
    static_assert(ROWS % 16 == 0 && Cin * RS * W4 <= 256 * NLD, "strip geometry");
    constexpr int IMG_F = SMP * Cin * RS * W;  // floats of the image strip
    extern __shared__ __attribute__((aligned(1024))) float smem[];
    float *dys = smem;                 // [ROWS][32] dY rows of the strip (DMA destination, 1 KiB granules)
    float *strip = smem + ROWS * N;    // [SMP][Cin][RS][W]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // uniform: LDS-DMA bases (M0) stay on the scalar unit
    const int i16 = lane & 15, kg = lane >> 4;
    const int words = Cin * RS * W4;
    int gofs[NLD], lofs[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        int q = tid + 256 * i;
        const bool ok = q < words;
        q = ok ? q : 0;
        const int x4 = q % W4, t1 = q / W4, row = t1 % RS, c = t1 / RS;
        gofs[i] = (c * H + row) * W + x4 * 4;
        lofs[i] = ok ? (c * RS + row) * W + x4 * 4 : -1;
    }
    // this lane's image origin for each of its wave's quads: pixel m = (wave*QW + q)*4 + kg inside the strip
    int origin[QW];
#pragma unroll
    for (int q = 0; q < QW; ++q) {
        const int m = (wave * QW + q) * 4 + kg;
        const int smp = m / (R * OW), p = m - smp * (R * OW), ohl = p / OW, ow = p - ohl * OW;
        origin[q] = (smp * Cin * RS + ohl * S + (i16 >> 1)) * W + ow * S + 4 * (i16 & 1);
    }
    f32x4 acc[Cin][4][2];
#pragma unroll
    for (int c = 0; c < Cin; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) acc[c][j][h] = f32x4{0.f, 0.f, 0.f, 0.f};
    float colacc = 0.f;
    const float sub = g.sub_mean, scl = g.inv_scale;
    uint32_t pre[SMP][NLD];
    const uint8_t *sbase[SMP];
    auto set_pair = [&](int pair) {
#pragma unroll
        for (int z = 0; z < SMP; ++z) {
            int sg = pair * SMP + z;
            sg = sg < nsamples ? sg : nsamples - 1;
            sbase[z] = in + sample_base(g, index, offset, in_stride, (uint32_t)sg);
        }
    };
    f32x4 tmu[NORM ? NLD : 1], trs[NORM ? NLD : 1];
    auto load_strip = [&](int st) {
        const int rowoff = st * R * S * W;
#pragma unroll
        for (int z = 0; z < SMP; ++z)
#pragma unroll
            for (int i = 0; i < NLD; ++i) pre[z][i] = *reinterpret_cast<const uint32_t *>(sbase[z] + rowoff + gofs[i]);
    };
    auto load_tables = [&](int st) {
        const int rowoff = st * R * S * W;
#pragma unroll
        for (int i = 0; i < (NORM ? NLD : 0); ++i) {
            tmu[i] = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(g.nmu + rowoff + gofs[i], 16));
            trs[i] = *reinterpret_cast<const f32x4 *>(__builtin_assume_aligned(g.nrstd + rowoff + gofs[i], 16));
        }
    };
    auto store_strip = [&]() {
#pragma unroll
        for (int z = 0; z < SMP; ++z)
#pragma unroll
            for (int i = 0; i < NLD; ++i) {
                f32x4 v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float b8 = (float)((pre[z][i] >> (8 * j)) & 0xFFu);
                    v[j] = (SUB ? b8 - sub : b8) * scl;
                    if (NORM) v[j] = clampf((v[j] - tmu[i][j]) * trs[i][j], -5.0f, 5.0f);
                }
                if (lofs[i] >= 0)
                    *reinterpret_cast<f32x4 *>(__builtin_assume_aligned(strip + z * Cin * RS * W + lofs[i], 16)) = v;
            }
    };
    // dY rows of (pair, strip): SMP runs of R*OW rows x 128 B; ROWS*128/1024 = 20 DMA instructions, 5 per wave
    auto dma_dy = [&](int pair, int st) {
#pragma unroll
        for (int i = 0; i < ROWS * N / 256 / 4; ++i) {
            const int j = i * 4 + wave, e = j * 256 + lane * 4;       // float index inside dys
            const int m = e >> 5, n4 = e & 31, smp = m / (R * OW), p = m - smp * (R * OW);
            int sg = pair * SMP + smp;
            const bool ok = sg < nsamples;
            sg = ok ? sg : nsamples - 1;
            const float *src = ok ? dy + ((int64_t)sg * OHOW + st * (R * OW) + p) * N + n4 : sf_zero_page;
            GLDS16(src, dys + j * 256);
        }
    };
    constexpr int nstrips = OH / R;
    // unit = (pair of this work-group, strip): pair-major (strip fastest), NORM strip-major (the table registers change 5
    // times per launch)
    const int my_pairs = (int)blockIdx.x < npairs ? (npairs - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
    const int total_units = my_pairs * nstrips;
    auto decode = [&](int unit, int &pr, int &st) {
        int lp;
        if (NORM) { st = unit / my_pairs; lp = unit - st * my_pairs; }
        else { lp = unit / nstrips; st = unit - lp * nstrips; }
        pr = (int)blockIdx.x + lp * (int)gridDim.x;
    };
    int pair = blockIdx.x, st = 0;
    if (total_units > 0) { set_pair(pair); load_tables(0); load_strip(0); dma_dy(pair, 0); }
    for (int unit = 0; unit < total_units; ++unit) {
        {
            decode(unit, pair, st);
            int npair = pair, nst = st;
            const bool more = unit + 1 < total_units;
            if (more) decode(unit + 1, npair, nst);
            store_strip();  // VALU + ds_write: runs while this strip's dY DMA (issued one phase ago) is in flight
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // prefetch the next unit's bytes (and, NORM, the next strip's table words: this strip's last use is over)
            if (more) {
                if (npair != pair) set_pair(npair);
                if (NORM && nst != st) load_tables(nst);
                load_strip(nst);
            }
            if (partial_b) {  // bias gradient: thread (n, part) sums every 8th row; parts are combined at the end
#pragma unroll 4
                for (int m = tid >> 5; m < ROWS; m += 8) colacc += dys[m * N + (tid & 31)];
            }
#pragma unroll
            for (int q = 0; q < QW; ++q) {
                const float *bp = dys + ((wave * QW + q) * 4 + kg) * N + i16;
                const float b0 = bp[0], b1 = bp[16];
#pragma unroll
                for (int c = 0; c < Cin; ++c) {
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(
                        __builtin_assume_aligned(strip + origin[q] + c * RS * W, 16));
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[c][j][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b0, acc[c][j][0], 0, 0, 0);
                        acc[c][j][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b1, acc[c][j][1], 0, 0, 0);
                    }
                }
            }
            __syncthreads();  // strip and dys are free again
            if (more) dma_dy(npair, nst);
        }
    }
    // ---- block reduction of the four waves' accumulators (fixed order: wave 0 + 1 + 2 + 3), then one partial per block
    float *red = smem;  // 4 x 8192 floats = 128 KiB would not fit: reduce in two rounds of pairs through 2 x 32 KiB
    // C layout of tile (c, j, h): reg r -> weight row k = c*64 + 4*(4*kg + r) + j, column n = 16*h + i16
    auto put = [&](float *dst) {
#pragma unroll
        for (int c = 0; c < Cin; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        dst[(c * 64 + 4 * (4 * kg + r) + j) * N + 16 * h + i16] = acc[c][j][h][r];
    };
    auto add = [&](const float *src) {
#pragma unroll
        for (int c = 0; c < Cin; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        acc[c][j][h][r] += src[(c * 64 + 4 * (4 * kg + r) + j) * N + 16 * h + i16];
    };
    if (wave & 1) put(red + (wave >> 1) * K * N);   // waves 1, 3 -> slots 0, 1
    __syncthreads();
    if (!(wave & 1)) add(red + (wave >> 1) * K * N);  // wave 0 += wave 1, wave 2 += wave 3
    __syncthreads();
    if (wave == 2) put(red);
    __syncthreads();
    if (wave == 0) {
        add(red);
        put(partial + (int64_t)blockIdx.x * K * N);
    }
    if (partial_b) {
        __syncthreads();
        red[tid] = colacc;
        __syncthreads();
        if (tid < N) {
            float sum = 0.f;
#pragma unroll
            for (int part = 0; part < 8; ++part) sum += red[part * 32 + tid];
            partial_b[(int64_t)blockIdx.x * N + tid] = sum;
        }
    }
}
template <int SMP, int R, bool SUB>
__global__ __launch_bounds__(256, 2) void k_conv1_wgrad_img(ConvG g, const uint8_t *__restrict__ in, int64_t in_stride,
                                                           const int32_t *__restrict__ index, int64_t offset,
                                                           const float *__restrict__ dy, float *__restrict__ partial,
                                                           float *__restrict__ partial_b, int nsamples, int npairs) {
    conv1_wgrad_img_body<SMP, R, SUB, false>(g, in, in_stride, index, offset, dy, partial, partial_b, nsamples, npairs);
}
template <int SMP, int R>
__global__ __launch_bounds__(256, 2) void k_conv1_wgrad_img_norm(ConvG g, const uint8_t *__restrict__ in, int64_t in_stride,
                                                                const int32_t *__restrict__ index, int64_t offset,
                                                                const float *__restrict__ dy, float *__restrict__ partial,
                                                                float *__restrict__ partial_b, int nsamples, int npairs) {
    conv1_wgrad_img_body<SMP, R, true, true>(g, in, in_stride, index, offset, dy, partial, partial_b, nsamples, npairs);
}

// libsf_hip.so — persistent LSTM sequence kernels for gfx950 (one launch per BPTT pass instead of ~6 launches per step).
//
// Reference path: model/core.py:19-64 (nn.LSTM over a PackedSequence) + algo/learning/rnn_utils.py:114-158; here the
// masked time loop over recurrence-length chunks (state zeroed after done/invalid steps), the loop form the reference's
// tests/algo/test_rnn.py proves equal to its packed path.
//
// The per-step recurrent GEMM of config 5 (512 chunk rows x 512 hidden x 2048 gate columns, 1.07 GFLOP) is too small
// to fill the chip as a launch of its own: measured 31 us forward / 65 us backward per step at 15-27 TFLOP/s, plus the
// cell and carry kernels (profiles/r02_c5_a_kernel_stats.csv).  Instead ONE launch walks all R steps:
//   * a work-group owns JB hidden units (= 4*JB gate columns; JB = 16 for H = 512) of one ROW GROUP of chunks and keeps
//     its slice of W_hh RESIDENT IN LDS for the whole pass (64 x 516 floats = 129 KB forward, 16 x 2052 backward);
//   * per step only the h (forward: 128 KB per group) / gate-gradient (backward: 512 KB) rows come from L2; the
//     products run on v_mfma_f32_16x16x4_f32 (exact f32), one ds_read_b128 + one 16-byte global load per 4 MFMAs;
//   * the cell non-linearity, the done/invalid masking, the carries of dL/dh and dL/dc and all saves for the backward
//     pass are fused into the epilogue (the MFMA accumulator layout hands every lane all four gates of its (row, unit));
//   * the work-groups of a row group (H/JB of them; block b -> group b % ngroups, i.e. one XCD per group under the
//     observed round-robin dispatch — a speed bonus only) exchange h_t / dgates_t through L2 with the placement-
//     independent write-through protocol of MI355X_MICROARCH.md (16-byte sc1 stores -> s_waitcnt vmcnt(0) -> barrier ->
//     relaxed agent-scope counter; consumer: relaxed poll -> sc1 loads), one hand-off per step, no grid-wide barrier.
// Work-groups spin on the counter, so all of them must be co-resident: the grid is ngroups * H/JB <= #CUs with one
// work-group per CU (LDS-limited).  A bounded spin turns a lost work-group (GPU shared with another process) into an
// error flag instead of a hang.
#include <stdlib.h>

#include "sf_common.h"

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((__vector_size__(16)));

namespace {

constexpr int SEQ_SYNC_STRIDE = 16;       // one counter per 64-byte line
constexpr int SEQ_ABORT_SLOT = 8 * SEQ_SYNC_STRIDE;
constexpr uint32_t SEQ_SPIN_LIMIT = 1u << 22;
constexpr int SEQ_MAX_SUB = 4;             // 64-row sub-tiles per work-group (register-resident state): Cn <= ngroups * 256
#ifndef SF_SEQ_NBUF
#define SF_SEQ_NBUF 2
#endif
#ifndef SF_SEQ_LOAD_AUX
#define SF_SEQ_LOAD_AUX 16  // cache policy of the hand-off payload loads: 16 = sc1 (served by L2), 0 = through the CU's L1
#endif
constexpr uint32_t OOB = 0x7FFFFFF0u;     // byte offset past every buffer: loads return 0, stores are dropped

// Cell transcendentals of the fused sequence passes.  SF_FAST_CELL = 1 (NOT the default; -DSF_FAST_CELL=1 through
// tools/build_variant.sh): hardware v_exp_f32 / v_rcp_f32 forms — ~4 instructions instead of the ~15 (expf) /
// ~25 (tanhf) of the accurate library code; absolute error <= 2e-7 (the relative error of tanh grows as 6e-8 / |x| near
// 0).  Vector instructions do not overlap with MFMAs on a SIMD, so they are matrix-pipe time: forward
// pass 0.440 -> 0.415 ms, configs[4] step 16.36 -> 16.09 ms over three alternations, the configs[4] LSTM / GRU replays
// against the reference and its float64 loop still within their bounds (profiles/r05_i_fastcell_ab.log) — but the GRU replay's
// distance from the float64 loop doubles (1.2e-4 -> 2.4e-4 of the largest weight delta; the reference's own fp32 run:
// 0.5e-4), and 1.6 % of a secondary line does not buy that: the accurate library forms stay the default.
#ifndef SF_FAST_CELL
#define SF_FAST_CELL 0
#endif
#if SF_FAST_CELL
__device__ __forceinline__ float sigm(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_c(float x) {
    const float t = __expf(-2.0f * fabsf(x));
    return copysignf((1.0f - t) * __frcp_rn(1.0f + t), x);
}
#else
__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanh_c(float x) { return tanhf(x); }
#endif

// arrive: every wave has drained its write-through stores, then one relaxed agent-scope increment
__device__ __forceinline__ void seq_arrive(unsigned *counter) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wait until `counter` >= target; returns false if the pass was aborted (bounded spin)
__device__ __forceinline__ bool seq_wait(unsigned *counter, unsigned target, unsigned *abort_flag, float *lds_flag) {
    if (threadIdx.x == 0) {
        uint32_t spins = 0;
        bool ok = true;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if ((++spins & 1023u) == 0 &&
                (spins >= SEQ_SPIN_LIMIT || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
        *lds_flag = ok ? 1.0f : 0.0f;
    }
    __syncthreads();
    if (SF_SEQ_LOAD_AUX == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // drop the CU's L1 copies of older payload lines
    const bool ok = *lds_flag != 0.0f;
    __syncthreads();  // the flag word may be rewritten by the next wait
    return ok;
}

struct LstmSeqFwd {
    const float *gx, *whh, *bhh, *keep;
    float *gates, *hprev, *hout, *cprev, *cout;
    unsigned *sync;
    int R, Cn, ngroups, rows_per_group;
    int64_t ho_rs, ho_ts;  // hout element (row, t) lives at row*ho_rs + t*ho_ts (+ unit): [R][Cn][H] or [Cn][R][H]
    // KXB > 0 (sf_lstm_seq_fwd_x): gx is not given but computed here, x [R][Cn][16*KXB] (time-major), wih_t [4H][16*KXB]
    // (gate-column-major copy of W_ih), bih [4H]
    const float *x, *wih_t, *bih;
};

// H hidden units, JB of them per work-group (JB = 16: the W_hh slice is 64 x (H + 4) floats of LDS — 129 KB at H = 512,
// 66.5 KB at H = 256; H = 1024 would need 263 KB, i.e. JB = 8 and a different accumulator-to-lane mapping: not built).
// KXB > 0: the input projection gx_t = x_t W_ih^T + b_ih (KXB 16-column blocks of x) is computed HERE instead of by a GEMM
// launch that writes [R][Cn][4H] floats for this kernel to read back: the W_ih fragments of the work-group's 64 gate
// columns live in registers, the x fragments of step t+1 are fetched behind step t's hand-off, and the 16*KXB MFMAs run
// while the first h rows of the step are on their way from L2 (placed between the hand-off and the wait they delayed
// every work-group alike: +1.8 us per step, measured).
template <int H, int JB, int NSUB, int KXB>
__global__ __launch_bounds__(256, 1) void k_lstm_seq_fwd(LstmSeqFwd p) {
    constexpr int G4 = 4 * H, NC = 4 * JB, NT = NC / 16, NU = JB / 16, LDW = H + 4, KX = 16 * KXB;
    constexpr int KU = 8, NKB = H / 16 / KU;
    constexpr int NBUF = NKB < SF_SEQ_NBUF ? NKB : SF_SEQ_NBUF;  // k-blocks of 8 x 16-byte loads per lane in flight (+ the one in the matrix pipe)
    constexpr int STG = 16 * JB;  // floats of one wave's staging tile
    static_assert(NKB * KU * 16 == H && NU >= 1, "shape");
    __shared__ __attribute__((aligned(16))) float lds[NC * LDW + 4 * STG + 4];
    float *wt = lds, *flag = lds + NC * LDW + 4 * STG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    float *stg = lds + NC * LDW + wave * STG;
    const int group = blockIdx.x % p.ngroups, j0 = (blockIdx.x / p.ngroups) * JB;
    const unsigned ncol = H / JB;
    const int Cn = p.Cn, R = p.R;
    const int rot = (int)(blockIdx.x / p.ngroups) % NKB;
    unsigned *counter = p.sync + group * SEQ_SYNC_STRIDE, *abort_flag = p.sync + SEQ_ABORT_SLOT;
    // ---- W_hh slice, transposed into LDS: wt[q*JB + u][k] = whh[k][q*H + j0 + u]
    for (int idx = tid; idx < NC * H; idx += 256) {
        const int lc = idx % NC, k = idx / NC, q = lc / JB, u = lc % JB;
        wt[lc * LDW + k] = p.whh[(int64_t)k * G4 + q * H + j0 + u];
    }
    float bias[4][NU], bih[4][NU];
    f32x4 bx[KXB > 0 ? KXB : 1][NT];  // lane (c, g): W_ih^T[gate column of tile nt, lane c][16*blk + 4*g ..]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            bias[q][u] = p.bhh[q * H + j0 + u * 16 + c];
            bih[q][u] = KXB > 0 ? p.bih[q * H + j0 + u * 16 + c] : 0.0f;
            if (KXB > 0) {
#pragma unroll
                for (int blk = 0; blk < KXB; ++blk)
                    bx[blk][q * NU + u] = *reinterpret_cast<const f32x4 *>(p.wih_t + (int64_t)(q * H + j0 + u * 16 + c) * KX + 16 * blk + 4 * g);
            }
        }
    __syncthreads();
    const auto h_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.hprev, 0, (int)((int64_t)(R + 1) * Cn * H * 4), 0x00020000);
    const int g_row0 = group * p.rows_per_group;
    const int g_rows_end = min(Cn, g_row0 + p.rows_per_group);

    // the cell state of this lane's (row, unit) elements stays in registers across the steps (the masked copy is also
    // stored to cprev[t+1] for the backward pass); sub-tiles are unrolled so that the register arrays index statically
    float cst[NSUB][4][NU];
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
        const int row0 = g_row0 + sub * 64 + wave * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + 4 * g + i, r = row < g_rows_end ? row : g_rows_end - 1;
#pragma unroll
            for (int u = 0; u < NU; ++u) cst[sub][i][u] = p.cprev[(int64_t)r * H + j0 + u * 16 + c];
        }
    }

    // operands of the cell epilogue for the next step: issued right after this step's hand-off, so they fly during the
    // wait for the other work-groups and the MFMA phase
    float xg[KXB > 0 ? 1 : NSUB][4][4][NU], kp[NSUB][4];
    f32x4 xa[NSUB][KXB > 0 ? KXB : 1];   // KXB > 0: x fragments of the NEXT projection (lane (c, g): row c, k = 16*blk + 4*g ..)
    f32x4 xacc[NSUB][KXB > 0 ? NT : 1];  // KXB > 0: x_t W_ih^T of the step about to run
    auto prefetch = [&](int t) {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int row0 = g_row0 + sub * 64 + wave * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 4 * g + i, r = row < g_rows_end ? row : g_rows_end - 1;
                const int64_t tr = (int64_t)t * Cn + r;
                kp[sub][i] = p.keep[tr];
                if constexpr (KXB == 0) {
#pragma unroll
                    for (int u = 0; u < NU; ++u)
#pragma unroll
                        for (int q = 0; q < 4; ++q) xg[sub][i][q][u] = p.gx[tr * G4 + q * H + j0 + u * 16 + c];
                }
            }
        }
    };
    auto load_x = [&](int t) {  // fragments of x_t
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int row = g_row0 + sub * 64 + wave * 16 + c, r = row < g_rows_end ? row : g_rows_end - 1;
#pragma unroll
            for (int blk = 0; blk < KXB; ++blk)
                xa[sub][blk] = *reinterpret_cast<const f32x4 *>(p.x + ((int64_t)t * Cn + r) * KX + 16 * blk + 4 * g);
        }
    };
    auto project_x = [&](int sub) {  // xacc = xa W_ih^T (this work-group's gate columns)
#pragma unroll
        for (int nt = 0; nt < (KXB > 0 ? NT : 1); ++nt) xacc[sub][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int blk = 0; blk < KXB; ++blk)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    xacc[sub][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[sub][blk][j], bx[blk][nt][j], xacc[sub][nt], 0, 0, 0);
    };
    prefetch(0);
    if (KXB > 0) load_x(0);

    for (int t = 0; t < R; ++t) {
        if (t > 0 && !seq_wait(counter, ncol * (unsigned)t, abort_flag, flag)) return;
        float sv[NSUB][4][NU][6];  // i, f, g, o, h, c of this step: stored AFTER the hand-off (nobody waits for them)
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int row0 = g_row0 + sub * 64 + wave * 16;
            if (row0 >= g_rows_end) continue;  // (wave-uniform; no block barrier inside the sub-tile body)
            // ---- gh = h_{t-1} W_hh: A rows from L2 (write-through hand-off: sc1 loads), B from the resident LDS slice
            f32x4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int arow = row0 + c;
            const uint32_t abase = arow < g_rows_end ? (uint32_t)((((int64_t)t * Cn + arow) * H + 4 * g) * 4) : OOB;
            i32x4 abuf[NBUF][KU];
            auto load_block = [&](int kb, i32x4 (&dst)[KU]) {
#pragma unroll
                for (int ku = 0; ku < KU; ++ku)
                    dst[ku] = __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, abase + (uint32_t)((kb * KU + ku) * 64), 0, SF_SEQ_LOAD_AUX);
            };
            // every work-group of the row group reads the SAME h rows: each starts its reduction at a different k-block
            // so that at any moment they pull different lines (different L2 channels) instead of all the same one
            auto kbe = [&](int kb) { const int k = kb + rot; return k >= NKB ? k - NKB : k; };
#pragma unroll
            for (int b = 0; b < NBUF - 1; ++b) load_block(kbe(b), abuf[b]);
            if constexpr (KXB > 0) {  // x_t W_ih^T while the first h rows are on their way from L2
                __builtin_amdgcn_sched_barrier(0);
                project_x(sub);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                if (kb + NBUF - 1 < NKB) load_block(kbe(kb + NBUF - 1), abuf[(kb + NBUF - 1) % NBUF]);
                if constexpr (KXB > 0) __builtin_amdgcn_sched_barrier(0);  // (with the projection's registers live hipcc sinks these loads to their use)
                const float *bpk = wt + c * LDW + kbe(kb) * (KU * 16) + 4 * g;
#pragma unroll
                for (int ku = 0; ku < KU; ++ku) {
                    const f32x4 a4 = __builtin_bit_cast(f32x4, abuf[kb % NBUF][ku]);
                    const float *bp = bpk + ku * 16;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bp + nt * 16 * LDW);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], b4[j], acc[nt], 0, 0, 0);
                    }
                }
            }
            // ---- LSTM cell (k_rnn_cell_fwd's arithmetic); masked state for step t+1 into the staging tile
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    float gxv[4];  // gx of this (row, unit): given, or the projection computed behind the last hand-off
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        gxv[q] = KXB > 0 ? xacc[sub][KXB > 0 ? q * NU + u : 0][i] + bih[q][u] : xg[KXB > 0 ? 0 : sub][i][q][u];
                    const float ig = sigm(gxv[0] + (acc[0 * NU + u][i] + bias[0][u]));
                    const float fg = sigm(gxv[1] + (acc[1 * NU + u][i] + bias[1][u]));
                    const float gg = tanh_c(gxv[2] + (acc[2 * NU + u][i] + bias[2][u]));
                    const float og = sigm(gxv[3] + (acc[3 * NU + u][i] + bias[3][u]));
                    const float cn = fg * cst[sub][i][u] + ig * gg;
                    const float h = og * tanh_c(cn);
                    cst[sub][i][u] = cn * kp[sub][i];
                    stg[(4 * g + i) * JB + u * 16 + c] = h * kp[sub][i];
                    sv[sub][i][u][0] = ig; sv[sub][i][u][1] = fg; sv[sub][i][u][2] = gg; sv[sub][i][u][3] = og;
                    sv[sub][i][u][4] = h; sv[sub][i][u][5] = cn;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            // h_t * keep -> hprev[t+1]: 16-byte write-through stores (the hand-off payload)
#pragma unroll
            for (int v = 0; v < NU; ++v) {
                const int f = v * 64 + lane, r = f / (JB / 4), c4 = f % (JB / 4), row = row0 + r;
                const f32x4 val = *reinterpret_cast<const f32x4 *>(stg + r * JB + c4 * 4);
                const uint32_t off = row < g_rows_end ? (uint32_t)((((int64_t)(t + 1) * Cn + row) * H + j0 + c4 * 4) * 4) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, val), h_rsrc, off, 0, 16);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (t + 1 < R) {
            seq_arrive(counter);
            prefetch(t + 1);
            if (KXB > 0) load_x(t + 1);  // fragments of the next step's input
        }
        // ---- saves for the backward pass (plain stores: they drain while this work-group waits for the others)
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int row0 = g_row0 + sub * 64 + wave * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 4 * g + i;
                if (row < g_rows_end) {
                    const int64_t tr = (int64_t)t * Cn + row;
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const int j = j0 + u * 16 + c;
                        float *go = p.gates + tr * G4 + j;
                        go[0] = sv[sub][i][u][0]; go[H] = sv[sub][i][u][1]; go[2 * H] = sv[sub][i][u][2]; go[3 * H] = sv[sub][i][u][3];
                        p.hout[(int64_t)row * p.ho_rs + (int64_t)t * p.ho_ts + j] = sv[sub][i][u][4];
                        p.cout[tr * H + j] = sv[sub][i][u][5];
                        p.cprev[(tr + Cn) * H + j] = cst[sub][i][u];
                    }
                }
            }
        }
    }
}

struct LstmSeqBwd {
    const float *dout, *gates, *cprev, *cout, *keep, *whh;
    float *dgx;
    unsigned *sync;
    int R, Cn, ngroups, rows_per_group;
    int64_t do_rs, do_ts;  // dout element (row, t) lives at row*do_rs + t*do_ts (+ unit)
    int ablate;  // timing experiments only (SF_LSTM_ABLATE, tools/lstm_bench.py): 1 no hand-off wait, 2 no phase-B loads,
                 // 4 no phase-B MFMAs, 8 no phase A — results are wrong with any bit set
};

template <int H, int JB, int NSUB>
__global__ __launch_bounds__(256, 1) void k_lstm_seq_bwd(LstmSeqBwd p) {
    constexpr int G4 = 4 * H, NC = 4 * JB, NU = JB / 16, LDK = G4 + 4;
    constexpr int KU = 8, NKB = G4 / 16 / KU;
    constexpr int NBUF = SF_SEQ_NBUF;  // k-blocks of 8 x 16-byte loads per lane: NBUF - 1 in flight behind the one in the matrix pipe
    constexpr int STG = 16 * NC;
    static_assert(NKB * KU * 16 == G4 && NKB >= NBUF - 1, "shape");
    __shared__ __attribute__((aligned(16))) float lds[JB * LDK + 4 * STG + 4];
    float *wk = lds, *flag = lds + JB * LDK + 4 * STG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    float *stg = lds + JB * LDK + wave * STG;
    const int group = blockIdx.x % p.ngroups, j0 = (blockIdx.x / p.ngroups) * JB;
    const unsigned ncol = H / JB;
    const int Cn = p.Cn, R = p.R;
    const int rot = (int)(blockIdx.x / p.ngroups) % NKB;  // staggered reduction start (see k_lstm_seq_fwd)
    unsigned *counter = p.sync + group * SEQ_SYNC_STRIDE, *abort_flag = p.sync + SEQ_ABORT_SLOT;
    // ---- W_hh rows j0 .. j0+JB-1 (all 4H gate columns of this group's hidden units): wk[kk][n] = whh[j0 + kk][n]
    for (int idx = tid; idx < JB * (G4 / 4); idx += 256) {
        const int kk = idx / (G4 / 4), n4 = idx % (G4 / 4);
        *reinterpret_cast<f32x4 *>(wk + kk * LDK + n4 * 4) =
            *reinterpret_cast<const f32x4 *>(p.whh + (int64_t)(j0 + kk) * G4 + n4 * 4);
    }
    __syncthreads();
    const auto d_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.dgx, 0, (int)((int64_t)R * Cn * G4 * 4), 0x00020000);
    const int g_row0 = group * p.rows_per_group;
    const int g_rows_end = min(Cn, g_row0 + p.rows_per_group);

    // dL/dh and dL/dc carried from step t+1 to step t for this lane's (row, unit) elements: registers
    float car_h[NSUB][4][NU], car_c[NSUB][4][NU];
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int u = 0; u < NU; ++u) car_h[sub][i][u] = car_c[sub][i][u] = 0.0f;

    // operands of the cell backward of the NEXT step (t-1): issued right after this step's hand-off, so they fly during
    // the wait and the MFMA phase (only the carry of dL/dh links phase A of step t-1 to phase B of step t)
    float pg[NSUB][4][NU][4], pdo[NSUB][4][NU], pco[NSUB][4][NU], pcp[NSUB][4][NU], pkp[NSUB][4];
    auto prefetch = [&](int t) {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int row0 = g_row0 + sub * 64 + wave * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 4 * g + i, r = row < g_rows_end ? row : g_rows_end - 1;
                const int64_t tr = (int64_t)t * Cn + r;
                pkp[sub][i] = t > 0 ? p.keep[tr - Cn] : 0.0f;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int j = j0 + u * 16 + c;
                    const float *go = p.gates + tr * G4 + j;
                    pg[sub][i][u][0] = go[0]; pg[sub][i][u][1] = go[H]; pg[sub][i][u][2] = go[2 * H]; pg[sub][i][u][3] = go[3 * H];
                    pdo[sub][i][u] = p.dout[(int64_t)r * p.do_rs + (int64_t)t * p.do_ts + j];
                    pco[sub][i][u] = p.cout[tr * H + j];
                    pcp[sub][i][u] = p.cprev[tr * H + j];
                }
            }
        }
    };
    prefetch(R - 1);

    for (int s = 0; s < R; ++s) {
        const int t = R - 1 - s;
        // ---- phase A: cell backward (k_rnn_cell_bwd's arithmetic) for this group's (rows, units); dgates -> dgx[t]
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int row0 = g_row0 + sub * 64 + wave * 16;
            if (row0 >= g_rows_end || (p.ablate & 8)) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float kprev = pkp[sub][i];
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const float ig = pg[sub][i][u][0], fg = pg[sub][i][u][1], gg = pg[sub][i][u][2], og = pg[sub][i][u][3];
                    float d = pdo[sub][i][u];
                    float dc_in = 0.0f;
                    if (s > 0) {
                        d = d + car_h[sub][i][u];
                        dc_in = car_c[sub][i][u];
                    }
                    const float tc = tanh_c(pco[sub][i][u]);
                    const float dc = d * og * (1.0f - tc * tc) + dc_in;
                    const float di = (dc * gg) * (ig * (1.0f - ig)), df = (dc * pcp[sub][i][u]) * (fg * (1.0f - fg));
                    const float dg = (dc * ig) * (1.0f - gg * gg), dob = (d * tc) * (og * (1.0f - og));
                    float *sp = stg + (4 * g + i) * NC + u * 16 + c;
                    sp[0] = di; sp[JB] = df; sp[2 * JB] = dg; sp[3 * JB] = dob;
                    car_c[sub][i][u] = (dc * fg) * kprev;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int v = 0; v < STG / 4 / 64; ++v) {  // 16 rows x NC floats as 16-byte write-through stores
                const int f = v * 64 + lane, r = f / (NC / 4), c4 = f % (NC / 4), row = row0 + r;
                const int q = (c4 * 4) / JB, u4 = (c4 * 4) % JB;
                const f32x4 val = *reinterpret_cast<const f32x4 *>(stg + r * NC + c4 * 4);
                const uint32_t off = row < g_rows_end ? (uint32_t)((((int64_t)t * Cn + row) * G4 + q * H + j0 + u4) * 4) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, val), d_rsrc, off, 0, 16);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (t == 0) break;  // no state in front of step 0
        seq_arrive(counter);
        prefetch(t - 1);
        if (!(p.ablate & 1) && !seq_wait(counter, ncol * (unsigned)(s + 1), abort_flag, flag)) return;
        // ---- phase B: dL/dh_{t-1}[rows, own units] = dgates_t[rows, :] W_hh[own units, :]^T, masked by keep[t-1]
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int row0 = g_row0 + sub * 64 + wave * 16;
            if (row0 >= g_rows_end) continue;
            f32x4 acc[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int arow = row0 + c;
            const uint32_t abase = (arow < g_rows_end && !(p.ablate & 2)) ? (uint32_t)((((int64_t)t * Cn + arow) * G4 + 4 * g) * 4) : OOB;
            i32x4 abuf[NBUF][KU];
            auto load_block = [&](int kb, i32x4 (&dst)[KU]) {
#pragma unroll
                for (int ku = 0; ku < KU; ++ku)
                    dst[ku] = __builtin_amdgcn_raw_buffer_load_b128(d_rsrc, abase + (uint32_t)((kb * KU + ku) * 64), 0, SF_SEQ_LOAD_AUX);
            };
            auto mma_block = [&](int kb, const i32x4 (&src)[KU]) {
                if (p.ablate & 4) return;
#pragma unroll
                for (int ku = 0; ku < KU; ++ku) {
                    const f32x4 a4 = __builtin_bit_cast(f32x4, src[ku]);
                    const float *bp = wk + c * LDK + (kb * KU + ku) * 16 + 4 * g;
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bp + u * 16 * LDK);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], b4[j], acc[u], 0, 0, 0);
                    }
                }
            };
            auto kbe = [&](int kb) { const int k = kb + rot; return k >= NKB ? k - NKB : k; };
            // block n lives in slot n % NBUF; NBUF - 1 blocks are in flight behind the one in the matrix pipe
#pragma unroll
            for (int b = 0; b < NBUF - 1; ++b) load_block(kbe(b), abuf[b]);
            for (int kb = 0; kb < NKB; kb += NBUF) {
#pragma unroll
                for (int b = 0; b < NBUF; ++b) {
                    if (kb + b + NBUF - 1 < NKB) load_block(kbe(kb + b + NBUF - 1), abuf[(b + NBUF - 1) % NBUF]);
                    if (kb + b < NKB) mma_block(kbe(kb + b), abuf[b]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 4 * g + i, r = row < g_rows_end ? row : g_rows_end - 1;
                const float kprev = p.keep[(int64_t)(t - 1) * Cn + r];
#pragma unroll
                for (int u = 0; u < NU; ++u) car_h[sub][i][u] = acc[u][i] * kprev;
            }
        }
    }
}


// ================================================================================================ GRU (torch gate order r, z, n)
// The reference's DEFAULT core (cfg rnn_type = gru, rnn_size = 512).  Same decomposition and hand-off protocol as the
// LSTM passes; a work-group's 16 hidden units are 48 gate columns (r, z, n), the candidate gate needs the recurrent part
// hn = h W_hn + b_hn separately (n = tanh(x_n + r * hn)), so the MFMA result is NOT pre-added to gx, and the backward
// pass has two gate-gradient arrays: dgx (for W_ih and the encoder) and dgh = {dr, dz, dn * r} (for W_hh; it is the
// hand-off payload), plus the direct path dL/dh_prev += dh * z.
struct GruSeqFwd {
    const float *gx, *whh, *bhh, *keep;
    float *gates, *hprev, *hout;
    unsigned *sync;
    int R, Cn, ngroups, rows_per_group;
    int64_t ho_rs, ho_ts;
    const float *x, *wih_t, *bih;  // KXB > 0 (sf_gru_seq_fwd_x): see LstmSeqFwd
};

template <int H, int JB, int NSUB, int KXB>
__global__ __launch_bounds__(256, 1) void k_gru_seq_fwd(GruSeqFwd p) {
    constexpr int G3 = 3 * H, G4 = 4 * H, NC = 3 * JB, NT = NC / 16, NU = JB / 16, LDW = H + 4, KX = 16 * KXB;
    constexpr int KU = 8, NKB = H / 16 / KU;
    constexpr int NBUF = NKB < SF_SEQ_NBUF ? NKB : SF_SEQ_NBUF;  // k-blocks of 8 x 16-byte loads per lane in flight (+ the one in the matrix pipe)
    constexpr int STG = 16 * JB;
    static_assert(NKB * KU * 16 == H && NU >= 1, "shape");
    __shared__ __attribute__((aligned(16))) float lds[NC * LDW + 4 * STG + 4];
    float *wt = lds, *flag = lds + NC * LDW + 4 * STG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    float *stg = lds + NC * LDW + wave * STG;
    const int group = blockIdx.x % p.ngroups, j0 = (blockIdx.x / p.ngroups) * JB;
    const unsigned ncol = H / JB;
    const int Cn = p.Cn, R = p.R;
    const int rot = (int)(blockIdx.x / p.ngroups) % NKB;
    unsigned *counter = p.sync + group * SEQ_SYNC_STRIDE, *abort_flag = p.sync + SEQ_ABORT_SLOT;
    for (int idx = tid; idx < NC * H; idx += 256) {  // wt[q*JB + u][k] = whh[k][q*H + j0 + u]
        const int lc = idx % NC, k = idx / NC, q = lc / JB, u = lc % JB;
        wt[lc * LDW + k] = p.whh[(int64_t)k * G3 + q * H + j0 + u];
    }
    float bias[3][NU], bih[3][NU];
    f32x4 bx[KXB > 0 ? KXB : 1][NT];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            bias[q][u] = p.bhh[q * H + j0 + u * 16 + c];
            bih[q][u] = KXB > 0 ? p.bih[q * H + j0 + u * 16 + c] : 0.0f;
            if (KXB > 0) {
#pragma unroll
                for (int blk = 0; blk < KXB; ++blk)
                    bx[blk][q * NU + u] = *reinterpret_cast<const f32x4 *>(p.wih_t + (int64_t)(q * H + j0 + u * 16 + c) * KX + 16 * blk + 4 * g);
            }
        }
    __syncthreads();
    const auto h_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.hprev, 0, (int)((int64_t)(R + 1) * Cn * H * 4), 0x00020000);
    const int g_row0 = group * p.rows_per_group;
    const int g_rows_end = min(Cn, g_row0 + p.rows_per_group);
    float hst[NSUB][4][NU];  // masked state entering the step, for this lane's (row, unit) elements
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub) {
        const int row0 = g_row0 + sub * 64 + wave * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + 4 * g + i, r = row < g_rows_end ? row : g_rows_end - 1;
#pragma unroll
            for (int u = 0; u < NU; ++u) hst[sub][i][u] = p.hprev[(int64_t)r * H + j0 + u * 16 + c];
        }
    }
    float xg[KXB > 0 ? 1 : NSUB][4][3][NU], kp[NSUB][4];
    f32x4 xa[NSUB][KXB > 0 ? KXB : 1], xacc[NSUB][KXB > 0 ? NT : 1];  // (see k_lstm_seq_fwd)
    auto prefetch = [&](int t) {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int row0 = g_row0 + sub * 64 + wave * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 4 * g + i, r = row < g_rows_end ? row : g_rows_end - 1;
                const int64_t tr = (int64_t)t * Cn + r;
                kp[sub][i] = p.keep[tr];
                if constexpr (KXB == 0) {
#pragma unroll
                    for (int u = 0; u < NU; ++u)
#pragma unroll
                        for (int q = 0; q < 3; ++q) xg[sub][i][q][u] = p.gx[tr * G3 + q * H + j0 + u * 16 + c];
                }
            }
        }
    };
    auto load_x = [&](int t) {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int row = g_row0 + sub * 64 + wave * 16 + c, r = row < g_rows_end ? row : g_rows_end - 1;
#pragma unroll
            for (int blk = 0; blk < KXB; ++blk)
                xa[sub][blk] = *reinterpret_cast<const f32x4 *>(p.x + ((int64_t)t * Cn + r) * KX + 16 * blk + 4 * g);
        }
    };
    auto project_x = [&](int sub) {
#pragma unroll
        for (int nt = 0; nt < (KXB > 0 ? NT : 1); ++nt) xacc[sub][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int blk = 0; blk < KXB; ++blk)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    xacc[sub][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[sub][blk][j], bx[blk][nt][j], xacc[sub][nt], 0, 0, 0);
    };
    prefetch(0);
    if (KXB > 0) load_x(0);

    for (int t = 0; t < R; ++t) {
        if (t > 0 && !seq_wait(counter, ncol * (unsigned)t, abort_flag, flag)) return;
        float sv[NSUB][4][NU][5];  // r, z, n, hn, h
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int row0 = g_row0 + sub * 64 + wave * 16;
            if (row0 >= g_rows_end) continue;
            f32x4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int arow = row0 + c;
            const uint32_t abase = arow < g_rows_end ? (uint32_t)((((int64_t)t * Cn + arow) * H + 4 * g) * 4) : OOB;
            i32x4 abuf[NBUF][KU];
            auto load_block = [&](int kb, i32x4 (&dst)[KU]) {
#pragma unroll
                for (int ku = 0; ku < KU; ++ku)
                    dst[ku] = __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, abase + (uint32_t)((kb * KU + ku) * 64), 0, SF_SEQ_LOAD_AUX);
            };
            auto kbe = [&](int kb) { const int k = kb + rot; return k >= NKB ? k - NKB : k; };
#pragma unroll
            for (int b = 0; b < NBUF - 1; ++b) load_block(kbe(b), abuf[b]);
            if constexpr (KXB > 0) {  // x_t W_ih^T while the first h rows are on their way from L2
                __builtin_amdgcn_sched_barrier(0);
                project_x(sub);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                if (kb + NBUF - 1 < NKB) load_block(kbe(kb + NBUF - 1), abuf[(kb + NBUF - 1) % NBUF]);
                if constexpr (KXB > 0) __builtin_amdgcn_sched_barrier(0);  // (with the projection's registers live hipcc sinks these loads to their use)
                const float *bpk = wt + c * LDW + kbe(kb) * (KU * 16) + 4 * g;
#pragma unroll
                for (int ku = 0; ku < KU; ++ku) {
                    const f32x4 a4 = __builtin_bit_cast(f32x4, abuf[kb % NBUF][ku]);
                    const float *bp = bpk + ku * 16;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bp + nt * 16 * LDW);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], b4[j], acc[nt], 0, 0, 0);
                    }
                }
            }
            // ---- GRU cell (k_rnn_cell_fwd's arithmetic)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    float gxv[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        gxv[q] = KXB > 0 ? xacc[sub][KXB > 0 ? q * NU + u : 0][i] + bih[q][u] : xg[KXB > 0 ? 0 : sub][i][q][u];
                    const float r = sigm(gxv[0] + (acc[0 * NU + u][i] + bias[0][u]));
                    const float z = sigm(gxv[1] + (acc[1 * NU + u][i] + bias[1][u]));
                    const float hn = acc[2 * NU + u][i] + bias[2][u];
                    const float n = tanh_c(gxv[2] + r * hn);
                    const float h = (1.0f - z) * n + z * hst[sub][i][u];
                    hst[sub][i][u] = h * kp[sub][i];
                    stg[(4 * g + i) * JB + u * 16 + c] = hst[sub][i][u];
                    sv[sub][i][u][0] = r; sv[sub][i][u][1] = z; sv[sub][i][u][2] = n; sv[sub][i][u][3] = hn; sv[sub][i][u][4] = h;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int v = 0; v < NU; ++v) {
                const int f = v * 64 + lane, r = f / (JB / 4), c4 = f % (JB / 4), row = row0 + r;
                const f32x4 val = *reinterpret_cast<const f32x4 *>(stg + r * JB + c4 * 4);
                const uint32_t off = row < g_rows_end ? (uint32_t)((((int64_t)(t + 1) * Cn + row) * H + j0 + c4 * 4) * 4) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, val), h_rsrc, off, 0, 16);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (t + 1 < R) {
            seq_arrive(counter);
            prefetch(t + 1);
            if (KXB > 0) load_x(t + 1);  // fragments of the next step's input
        }
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int row0 = g_row0 + sub * 64 + wave * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 4 * g + i;
                if (row < g_rows_end) {
                    const int64_t tr = (int64_t)t * Cn + row;
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const int j = j0 + u * 16 + c;
                        float *go = p.gates + tr * G4 + j;
                        go[0] = sv[sub][i][u][0]; go[H] = sv[sub][i][u][1]; go[2 * H] = sv[sub][i][u][2]; go[3 * H] = sv[sub][i][u][3];
                        p.hout[(int64_t)row * p.ho_rs + (int64_t)t * p.ho_ts + j] = sv[sub][i][u][4];
                    }
                }
            }
        }
    }
}

struct GruSeqBwd {
    const float *dout, *gates, *hprev, *keep, *whh;
    float *dgx, *dgh;
    unsigned *sync;
    int R, Cn, ngroups, rows_per_group;
    int64_t do_rs, do_ts;
};

template <int H, int JB, int NSUB>
__global__ __launch_bounds__(256, 1) void k_gru_seq_bwd(GruSeqBwd p) {
    constexpr int G3 = 3 * H, G4 = 4 * H, NC = 3 * JB, NU = JB / 16, LDK = G3 + 4;
    constexpr int KU = 8, NKB = G3 / 16 / KU;
    constexpr int NBUF = SF_SEQ_NBUF;
    constexpr int STG = 16 * NC;
    static_assert(NKB * KU * 16 == G3 && NKB >= NBUF - 1, "shape");
    __shared__ __attribute__((aligned(16))) float lds[JB * LDK + 4 * STG + 4];
    float *wk = lds, *flag = lds + JB * LDK + 4 * STG;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15, g = lane >> 4;
    float *stg = lds + JB * LDK + wave * STG;
    const int group = blockIdx.x % p.ngroups, j0 = (blockIdx.x / p.ngroups) * JB;
    const unsigned ncol = H / JB;
    const int Cn = p.Cn, R = p.R;
    const int rot = (int)(blockIdx.x / p.ngroups) % NKB;
    unsigned *counter = p.sync + group * SEQ_SYNC_STRIDE, *abort_flag = p.sync + SEQ_ABORT_SLOT;
    for (int idx = tid; idx < JB * (G3 / 4); idx += 256) {  // wk[kk][n] = whh[j0 + kk][n]
        const int kk = idx / (G3 / 4), n4 = idx % (G3 / 4);
        *reinterpret_cast<f32x4 *>(wk + kk * LDK + n4 * 4) =
            *reinterpret_cast<const f32x4 *>(p.whh + (int64_t)(j0 + kk) * G3 + n4 * 4);
    }
    __syncthreads();
    const auto d_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.dgh, 0, (int)((int64_t)R * Cn * G3 * 4), 0x00020000);
    const int g_row0 = group * p.rows_per_group;
    const int g_rows_end = min(Cn, g_row0 + p.rows_per_group);
    float car_h[NSUB][4][NU];
#pragma unroll
    for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int u = 0; u < NU; ++u) car_h[sub][i][u] = 0.0f;
    float pg[NSUB][4][NU][4], pdo[NSUB][4][NU], php[NSUB][4][NU], pkp[NSUB][4];
    auto prefetch = [&](int t) {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int row0 = g_row0 + sub * 64 + wave * 16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 4 * g + i, r = row < g_rows_end ? row : g_rows_end - 1;
                const int64_t tr = (int64_t)t * Cn + r;
                pkp[sub][i] = t > 0 ? p.keep[tr - Cn] : 0.0f;
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int j = j0 + u * 16 + c;
                    const float *go = p.gates + tr * G4 + j;
                    pg[sub][i][u][0] = go[0]; pg[sub][i][u][1] = go[H]; pg[sub][i][u][2] = go[2 * H]; pg[sub][i][u][3] = go[3 * H];
                    pdo[sub][i][u] = p.dout[(int64_t)r * p.do_rs + (int64_t)t * p.do_ts + j];
                    php[sub][i][u] = p.hprev[tr * H + j];
                }
            }
        }
    };
    prefetch(R - 1);

    for (int s = 0; s < R; ++s) {
        const int t = R - 1 - s;
        float dir[NSUB][4][NU];  // dL/dh_prev that does not go through W_hh: dh * z
        // ---- phase A: GRU cell backward (k_rnn_cell_bwd's arithmetic); dgx plain, dgh = the hand-off payload
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int row0 = g_row0 + sub * 64 + wave * 16;
            if (row0 >= g_rows_end) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 4 * g + i;
                const bool ok = row < g_rows_end;
                const int64_t tr = (int64_t)t * Cn + (ok ? row : g_rows_end - 1);
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const float r = pg[sub][i][u][0], z = pg[sub][i][u][1], n = pg[sub][i][u][2], hn = pg[sub][i][u][3];
                    float d = pdo[sub][i][u];
                    if (s > 0) d = d + car_h[sub][i][u];
                    const float dn_pre = (d * (1.0f - z)) * (1.0f - n * n);
                    const float dz_pre = (d * (php[sub][i][u] - n)) * (z * (1.0f - z));
                    const float dr_pre = (dn_pre * hn) * (r * (1.0f - r));
                    float *sp = stg + (4 * g + i) * NC + u * 16 + c;
                    sp[0] = dr_pre; sp[JB] = dz_pre; sp[2 * JB] = dn_pre * r;
                    dir[sub][i][u] = d * z;
                    if (ok) {
                        float *x = p.dgx + tr * G3 + j0 + u * 16 + c;
                        x[0] = dr_pre; x[H] = dz_pre; x[2 * H] = dn_pre;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int v = 0; v < STG / 4 / 64; ++v) {  // 16 rows x NC floats as 16-byte write-through stores
                const int f = v * 64 + lane, r = f / (NC / 4), c4 = f % (NC / 4), row = row0 + r;
                const int q = (c4 * 4) / JB, u4 = (c4 * 4) % JB;
                const f32x4 val = *reinterpret_cast<const f32x4 *>(stg + r * NC + c4 * 4);
                const uint32_t off = row < g_rows_end ? (uint32_t)((((int64_t)t * Cn + row) * G3 + q * H + j0 + u4) * 4) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, val), d_rsrc, off, 0, 16);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (t == 0) break;
        seq_arrive(counter);
        float kcur[NSUB][4];
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub)
#pragma unroll
            for (int i = 0; i < 4; ++i) kcur[sub][i] = pkp[sub][i];  // keep[t-1] of THIS step (prefetch overwrites pkp)
        prefetch(t - 1);
        if (!seq_wait(counter, ncol * (unsigned)(s + 1), abort_flag, flag)) return;
        // ---- phase B: dL/dh_{t-1}[rows, own units] = (dgh_t[rows, :] W_hh[own units, :]^T + dh * z) * keep[t-1]
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int row0 = g_row0 + sub * 64 + wave * 16;
            if (row0 >= g_rows_end) continue;
            f32x4 acc[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
            const int arow = row0 + c;
            const uint32_t abase = arow < g_rows_end ? (uint32_t)((((int64_t)t * Cn + arow) * G3 + 4 * g) * 4) : OOB;
            i32x4 abuf[NBUF][KU];
            auto load_block = [&](int kb, i32x4 (&dst)[KU]) {
#pragma unroll
                for (int ku = 0; ku < KU; ++ku)
                    dst[ku] = __builtin_amdgcn_raw_buffer_load_b128(d_rsrc, abase + (uint32_t)((kb * KU + ku) * 64), 0, SF_SEQ_LOAD_AUX);
            };
            auto mma_block = [&](int kb, const i32x4 (&src)[KU]) {
#pragma unroll
                for (int ku = 0; ku < KU; ++ku) {
                    const f32x4 a4 = __builtin_bit_cast(f32x4, src[ku]);
                    const float *bp = wk + c * LDK + (kb * KU + ku) * 16 + 4 * g;
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const f32x4 b4 = *reinterpret_cast<const f32x4 *>(bp + u * 16 * LDK);
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], b4[j], acc[u], 0, 0, 0);
                    }
                }
            };
            auto kbe = [&](int kb) { const int k = kb + rot; return k >= NKB ? k - NKB : k; };
            // block n lives in slot n % NBUF; NBUF - 1 blocks are in flight behind the one in the matrix pipe
#pragma unroll
            for (int b = 0; b < NBUF - 1; ++b) load_block(kbe(b), abuf[b]);
            for (int kb = 0; kb < NKB; kb += NBUF) {
#pragma unroll
                for (int b = 0; b < NBUF; ++b) {
                    if (kb + b + NBUF - 1 < NKB) load_block(kbe(kb + b + NBUF - 1), abuf[(b + NBUF - 1) % NBUF]);
                    if (kb + b < NKB) mma_block(kbe(kb + b), abuf[b]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int u = 0; u < NU; ++u) car_h[sub][i][u] = (acc[u][i] + dir[sub][i][u]) * kcur[sub][i];
        }
    }
}

int seq_plan(int Cn, int H, int *ngroups, int *rows_per_group, int *jb) {
    // widths with a compiled instantiation (JB = 16; the k-blocking needs H % 128 == 0 forward, an even number of
    // 128-column blocks backward: 256 and 512 satisfy both for LSTM and GRU); other widths take the per-step path
    if (H != 512 && H != 256) return 0;
    *jb = 16;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return 0;
    const int ncol = H / *jb;
    int ng = cus / ncol;
    if (ng > 8) ng = 8;
    const int need = (Cn + 15) / 16;
    if (ng > need) ng = need;
    if (ng < 1) return 0;
    *ngroups = ng;
    *rows_per_group = ((Cn + ng - 1) / ng + 15) / 16 * 16;
    return *rows_per_group <= 64 * SEQ_MAX_SUB;
}

#include "sf_rnn_regw.h"

}  // namespace

// backward passes with the weights in registers (sf_rnn_regw.h): one instantiation per (width, row tiles per group)
#define SEQ_DISPATCH_R(KERN)                                                         \
    do {                                                                             \
        if (H == 512) {                                                              \
            if (rpg == 32) KERN<512, 2><<<grid, block, 0, STREAM(stream)>>>(p);      \
            else KERN<512, 4><<<grid, block, 0, STREAM(stream)>>>(p);                \
        } else {                                                                     \
            if (rpg == 32) KERN<256, 2><<<grid, block, 0, STREAM(stream)>>>(p);      \
            else KERN<256, 4><<<grid, block, 0, STREAM(stream)>>>(p);                \
        }                                                                            \
    } while (0)

// one instantiation per (width, sub-tiles per work-group)
#define SEQ_DISPATCH(KERN, ...)                                                                       \
    do {                                                                                              \
        if (H == 512) {                                                                               \
            if (nsub == 1) KERN<512, 16, 1 __VA_ARGS__><<<grid, block, 0, STREAM(stream)>>>(p);       \
            else if (nsub == 2) KERN<512, 16, 2 __VA_ARGS__><<<grid, block, 0, STREAM(stream)>>>(p);  \
            else KERN<512, 16, 4 __VA_ARGS__><<<grid, block, 0, STREAM(stream)>>>(p);                 \
        } else {                                                                                      \
            if (nsub == 1) KERN<256, 16, 1 __VA_ARGS__><<<grid, block, 0, STREAM(stream)>>>(p);       \
            else if (nsub == 2) KERN<256, 16, 2 __VA_ARGS__><<<grid, block, 0, STREAM(stream)>>>(p);  \
            else KERN<256, 16, 4 __VA_ARGS__><<<grid, block, 0, STREAM(stream)>>>(p);                 \
        }                                                                                             \
    } while (0)
// forward passes with the input projection fused in (x has 64 columns; register budget: one or two sub-tiles)
#define SEQ_DISPATCH_X(KERN)                                                                  \
    do {                                                                                      \
        if (H == 512) {                                                                       \
            if (nsub == 1) KERN<512, 16, 1, 4><<<grid, block, 0, STREAM(stream)>>>(p);        \
            else KERN<512, 16, 2, 4><<<grid, block, 0, STREAM(stream)>>>(p);                  \
        } else {                                                                              \
            if (nsub == 1) KERN<256, 16, 1, 4><<<grid, block, 0, STREAM(stream)>>>(p);        \
            else KERN<256, 16, 2, 4><<<grid, block, 0, STREAM(stream)>>>(p);                  \
        }                                                                                     \
    } while (0)

extern "C" int sf_lstm_seq_supported(int Cn, int H) {
    int a, b, c;
    return Cn > 0 && seq_plan(Cn, H, &a, &b, &c);
}

static int lstm_seq_fwd_impl(const float *gx, const float *x, const float *wih_t, const float *bih, int Kx,
                             const float *whh, const float *bhh, const float *keep, float *gates, float *hprev, float *hout,
                             float *cprev, float *cout, uint32_t *sync, int R, int Cn, int H, int env_major, void *stream) {
    SF_REQUIRE((gx || (x && wih_t && bih)) && whh && bhh && keep && gates && hprev && hout && cprev && cout && sync && R > 0 && Cn > 0,
               "sf_lstm_seq_fwd: bad args");
    int ng, rpg, jb;
    SF_REQUIRE(seq_plan(Cn, H, &ng, &rpg, &jb), "sf_lstm_seq_fwd: unsupported shape Cn=%d H=%d (see sf_lstm_seq_supported)", Cn, H);
    SF_REQUIRE((int64_t)(R + 1) * Cn * H * 4 < 0x7FFFFFF0LL, "sf_lstm_seq_fwd: state buffer exceeds 2 GiB");
    const int nsub = (rpg + 63) / 64;  // 64-row sub-tiles per work-group, unrolled at compile time (register-resident state)
    if (!gx)
        SF_REQUIRE(Kx == 64 && nsub <= 2 && (((uintptr_t)x | (uintptr_t)wih_t) & 15) == 0,
                   "sf_lstm_seq_fwd_x: unsupported shape Cn=%d H=%d Kx=%d (see sf_seq_fwd_x_supported)", Cn, H, Kx);
    int rc = sf_hip_status(hipMemsetAsync(sync, 0, SEQ_ABORT_SLOT * sizeof(uint32_t), STREAM(stream)), "sf_lstm_seq_fwd memset");
    if (rc) return rc;
    // hout [R][Cn][H] (time-major) or, env_major, [Cn][R][H] = the row order of the minibatch itself (no transpose copy)
    LstmSeqFwd p{gx, whh, bhh, keep, gates, hprev, hout, cprev, cout, sync, R, Cn, ng, rpg,
                 env_major ? (int64_t)R * H : (int64_t)H, env_major ? (int64_t)H : (int64_t)Cn * H, x, wih_t, bih};
    const dim3 grid((unsigned)(ng * (H / jb))), block(256);
    if (gx) SEQ_DISPATCH(k_lstm_seq_fwd, , 0);
    else SEQ_DISPATCH_X(k_lstm_seq_fwd);
    return sf_launch_status("sf_lstm_seq_fwd");
}
extern "C" int sf_lstm_seq_fwd(const float *gx, const float *whh, const float *bhh, const float *keep, float *gates,
                               float *hprev, float *hout, float *cprev, float *cout, uint32_t *sync, int R, int Cn, int H,
                               int env_major, void *stream) {
    SF_REQUIRE(gx, "sf_lstm_seq_fwd: bad args");
    return lstm_seq_fwd_impl(gx, nullptr, nullptr, nullptr, 0, whh, bhh, keep, gates, hprev, hout, cprev, cout, sync, R, Cn, H,
                             env_major, stream);
}
extern "C" int sf_seq_fwd_x_supported(int Cn, int H, int Kx) {
    static const int on = getenv("SF_SEQ_FWD_X") ? atoi(getenv("SF_SEQ_FWD_X")) : 1;
    int ng, rpg, jb;
    return on && Cn > 0 && Kx == 64 && seq_plan(Cn, H, &ng, &rpg, &jb) && (rpg + 63) / 64 <= 2;
}
extern "C" int sf_lstm_seq_fwd_x(const float *x, const float *wih_t, const float *bih, int Kx, const float *whh,
                                 const float *bhh, const float *keep, float *gates, float *hprev, float *hout, float *cprev,
                                 float *cout, uint32_t *sync, int R, int Cn, int H, int env_major, void *stream) {
    SF_REQUIRE(x && wih_t && bih, "sf_lstm_seq_fwd_x: bad args");
    return lstm_seq_fwd_impl(nullptr, x, wih_t, bih, Kx, whh, bhh, keep, gates, hprev, hout, cprev, cout, sync, R, Cn, H,
                             env_major, stream);
}

extern "C" int sf_lstm_seq_bwd(const float *dout, const float *gates, const float *cprev, const float *cout,
                               const float *keep, const float *whh, float *dgx, uint32_t *sync, int R, int Cn, int H,
                               int env_major, void *stream) {
    SF_REQUIRE(dout && gates && cprev && cout && keep && whh && dgx && sync && R > 0 && Cn > 0,
               "sf_lstm_seq_bwd: bad args");
    int ng, rpg, jb;
    SF_REQUIRE(seq_plan(Cn, H, &ng, &rpg, &jb), "sf_lstm_seq_bwd: unsupported shape Cn=%d H=%d (see sf_lstm_seq_supported)", Cn, H);
    SF_REQUIRE((int64_t)R * Cn * 4 * H * 4 < 0x7FFFFFF0LL, "sf_lstm_seq_bwd: gate-gradient buffer exceeds 2 GiB");
    int rc = sf_hip_status(hipMemsetAsync(sync, 0, SEQ_ABORT_SLOT * sizeof(uint32_t), STREAM(stream)), "sf_lstm_seq_bwd memset");
    if (rc) return rc;
    static const int ablate = getenv("SF_LSTM_ABLATE") ? atoi(getenv("SF_LSTM_ABLATE")) : 0;
    int ngr, rpgr;
    if (seq_plan_r(Cn, H, &ngr, &rpgr)) {  // 32 hidden units per work-group, W_hh slice in registers
        const int rpg = rpgr;
        LstmSeqBwd p{dout, gates, cprev, cout, keep, whh, dgx, sync, R, Cn, ngr, rpg,
                     env_major ? (int64_t)R * H : (int64_t)H, env_major ? (int64_t)H : (int64_t)Cn * H, ablate};
        const dim3 grid((unsigned)(ngr * (H / 32))), block(256);
        SEQ_DISPATCH_R(k_lstm_seq_bwd_r);
        return sf_launch_status("sf_lstm_seq_bwd");
    }
    LstmSeqBwd p{dout, gates, cprev, cout, keep, whh, dgx, sync, R, Cn, ng, rpg,
                 env_major ? (int64_t)R * H : (int64_t)H, env_major ? (int64_t)H : (int64_t)Cn * H, ablate};
    const dim3 grid((unsigned)(ng * (H / jb))), block(256);
    const int nsub = (rpg + 63) / 64;
    SEQ_DISPATCH(k_lstm_seq_bwd);
    return sf_launch_status("sf_lstm_seq_bwd");
}

static int gru_seq_fwd_impl(const float *gx, const float *x, const float *wih_t, const float *bih, int Kx, const float *whh,
                            const float *bhh, const float *keep, float *gates, float *hprev, float *hout, uint32_t *sync,
                            int R, int Cn, int H, int env_major, void *stream) {
    SF_REQUIRE((gx || (x && wih_t && bih)) && whh && bhh && keep && gates && hprev && hout && sync && R > 0 && Cn > 0,
               "sf_gru_seq_fwd: bad args");
    int ng, rpg, jb;
    SF_REQUIRE(seq_plan(Cn, H, &ng, &rpg, &jb), "sf_gru_seq_fwd: unsupported shape Cn=%d H=%d (see sf_lstm_seq_supported)", Cn, H);
    SF_REQUIRE((int64_t)(R + 1) * Cn * H * 4 < 0x7FFFFFF0LL, "sf_gru_seq_fwd: state buffer exceeds 2 GiB");
    const int nsub = (rpg + 63) / 64;
    if (!gx)
        SF_REQUIRE(Kx == 64 && nsub <= 2 && (((uintptr_t)x | (uintptr_t)wih_t) & 15) == 0,
                   "sf_gru_seq_fwd_x: unsupported shape Cn=%d H=%d Kx=%d (see sf_seq_fwd_x_supported)", Cn, H, Kx);
    int rc = sf_hip_status(hipMemsetAsync(sync, 0, SEQ_ABORT_SLOT * sizeof(uint32_t), STREAM(stream)), "sf_gru_seq_fwd memset");
    if (rc) return rc;
    GruSeqFwd p{gx, whh, bhh, keep, gates, hprev, hout, sync, R, Cn, ng, rpg,
                env_major ? (int64_t)R * H : (int64_t)H, env_major ? (int64_t)H : (int64_t)Cn * H, x, wih_t, bih};
    const dim3 grid((unsigned)(ng * (H / jb))), block(256);
    if (gx) SEQ_DISPATCH(k_gru_seq_fwd, , 0);
    else SEQ_DISPATCH_X(k_gru_seq_fwd);
    return sf_launch_status("sf_gru_seq_fwd");
}
extern "C" int sf_gru_seq_fwd(const float *gx, const float *whh, const float *bhh, const float *keep, float *gates,
                              float *hprev, float *hout, uint32_t *sync, int R, int Cn, int H, int env_major,
                              void *stream) {
    SF_REQUIRE(gx, "sf_gru_seq_fwd: bad args");
    return gru_seq_fwd_impl(gx, nullptr, nullptr, nullptr, 0, whh, bhh, keep, gates, hprev, hout, sync, R, Cn, H, env_major,
                            stream);
}
extern "C" int sf_gru_seq_fwd_x(const float *x, const float *wih_t, const float *bih, int Kx, const float *whh,
                                const float *bhh, const float *keep, float *gates, float *hprev, float *hout, uint32_t *sync,
                                int R, int Cn, int H, int env_major, void *stream) {
    SF_REQUIRE(x && wih_t && bih, "sf_gru_seq_fwd_x: bad args");
    return gru_seq_fwd_impl(nullptr, x, wih_t, bih, Kx, whh, bhh, keep, gates, hprev, hout, sync, R, Cn, H, env_major, stream);
}

extern "C" int sf_gru_seq_bwd(const float *dout, const float *gates, const float *hprev, const float *keep,
                              const float *whh, float *dgx, float *dgh, uint32_t *sync, int R, int Cn, int H,
                              int env_major, void *stream) {
    SF_REQUIRE(dout && gates && hprev && keep && whh && dgx && dgh && sync && R > 0 && Cn > 0, "sf_gru_seq_bwd: bad args");
    int ng, rpg, jb;
    SF_REQUIRE(seq_plan(Cn, H, &ng, &rpg, &jb), "sf_gru_seq_bwd: unsupported shape Cn=%d H=%d (see sf_lstm_seq_supported)", Cn, H);
    SF_REQUIRE((int64_t)R * Cn * 3 * H * 4 < 0x7FFFFFF0LL, "sf_gru_seq_bwd: gate-gradient buffer exceeds 2 GiB");
    int rc = sf_hip_status(hipMemsetAsync(sync, 0, SEQ_ABORT_SLOT * sizeof(uint32_t), STREAM(stream)), "sf_gru_seq_bwd memset");
    if (rc) return rc;
    int ngr, rpgr;
    if (seq_plan_r(Cn, H, &ngr, &rpgr)) {
        const int rpg = rpgr;
        GruSeqBwd p{dout, gates, hprev, keep, whh, dgx, dgh, sync, R, Cn, ngr, rpg,
                    env_major ? (int64_t)R * H : (int64_t)H, env_major ? (int64_t)H : (int64_t)Cn * H};
        const dim3 grid((unsigned)(ngr * (H / 32))), block(256);
        SEQ_DISPATCH_R(k_gru_seq_bwd_r);
        return sf_launch_status("sf_gru_seq_bwd");
    }
    GruSeqBwd p{dout, gates, hprev, keep, whh, dgx, dgh, sync, R, Cn, ng, rpg,
                env_major ? (int64_t)R * H : (int64_t)H, env_major ? (int64_t)H : (int64_t)Cn * H};
    const dim3 grid((unsigned)(ng * (H / jb))), block(256);
    const int nsub = (rpg + 63) / 64;
    SEQ_DISPATCH(k_gru_seq_bwd);
    return sf_launch_status("sf_gru_seq_bwd");
}

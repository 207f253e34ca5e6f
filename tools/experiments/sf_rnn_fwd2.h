// PARKED EXPERIMENT (round 4) — not compiled into libsf_hip.so.  Forward LSTM sequence pass with TWO 32-row groups per
// work-group (eight waves: every SIMD holds one wave of each group), meant to hide a group's hand-off latency behind
// the other group's matrix phase.  Correct (tests/test_gpu_rl_kernels.py sequence tests green with it dispatched for
// H = 512, 256 <= Cn <= 512) and SLOWER than k_lstm_seq_fwd: 19.4-20 us per step against 14.9 at configs[4]
// (profiles/r04_d_seq_fwd2_ablate.log).  Why (tools/ubench/simd_overlap.hip, profiles/r04_d_simd_overlap.log): on
// gfx950 the MFMAs of one wave and the VALU work of ANOTHER wave on the same SIMD do not overlap at all (0.87 ms + 1.04 ms
// -> 1.89 ms), so a step costs MFMA 7.7 us + cell VALU 4.1 us per SIMD however the waves are arranged; only memory
// latency hides, and per-wave arrivals / polls (37 us per step) or even the LDS-relayed ones cost more than that saves.
// A matrix-phase token that forces the two waves of a SIMD to alternate makes it slower still (22.6 us).
// To try it again: paste into sf_rnn.hip in front of `struct LstmSeqBwd`, add `int ablate;` to LstmSeqFwd, and dispatch
// from lstm_seq_fwd_impl with p.ngroups = Cn / 32, grid (Cn / 64) * 32, block 512.
// ---- forward pass with TWO row groups per work-group (H = 512, 256 <= Cn <= 512 rows in 64-row steps).
// k_lstm_seq_fwd above has one wave per SIMD: while a work-group waits for its group's hand-off (write-through drain,
// counter, poll, first h rows from L2, lockstep skew of the 32 work-groups) its matrix pipes idle.  Here the row
// groups are 32 rows (16 of them at Cn = 512) and a work-group of EIGHT waves serves the same 16 hidden units of two
// groups: waves 0-3 group 2p, waves 4-7 group 2p + 1, so every SIMD holds one wave of each group and multiplies one
// group's step while the other group's hand-off is in flight.  The waves never meet after the prologue: a wave owns 16
// rows x 32 gate columns (8 units x {i, f, g, o}); the four waves of a group meet through two LDS words only (arrival
// count, release flag: no block barrier in the loop), and the two groups drift apart by half a step on their own.  The LDS slice is the same 64 columns x (H + 4) floats; columns
// are ordered [unit half][{i | f}, {g | o}][8 units], so a 16-lane row of the accumulator holds i / g (lanes 0-7) and
// f / o (lanes 8-15) of the same 8 units: one DPP row rotation by 8 hands every lane the other two gates, and each half
// row finishes two of the four accumulator rows (all 64 lanes busy in the cell).  Grid: (Cn / 64) x 32 <= #CUs, block
// b -> pair b % (Cn / 64): both groups of a pair stay on one XCD under the round-robin dispatch (speed only).
constexpr int SEQ_SYNC_STRIDE2 = 8;  // 16 counters in front of the abort word

__device__ __forceinline__ float selp(int cond, float a, float b) { return cond ? a : b; }
__device__ __forceinline__ float dpp_row_ror8(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));
}
// one wave waits until `counter` >= target (every lane loads the same word; the exit test is wave-uniform)
__device__ __forceinline__ bool seq_wait_wave(unsigned *counter, unsigned target, unsigned *abort_flag) {
    uint32_t spins = 0;
    while ((unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < target) {
        __builtin_amdgcn_s_sleep(4);
        if ((++spins & 1023u) == 0 &&
            (spins >= SEQ_SPIN_LIMIT ||
             __builtin_amdgcn_readfirstlane((int)__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)))) {
            __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

template <int H, int KXB>
__global__ __launch_bounds__(512, 1) void k_lstm_seq_fwd2(LstmSeqFwd p) {
    constexpr int G4 = 4 * H, JB = 16, NC = 4 * JB, NT = 2, LDW = H + 4, KX = 16 * KXB;
    constexpr int KU = 8, NKB = H / 16 / KU, NBUF = 2, STG = 16 * 8;
    static_assert(NKB * KU * 16 == H && NKB >= NBUF, "shape");
    __shared__ __attribute__((aligned(16))) float lds[NC * LDW + 8 * STG];
    __shared__ unsigned relay[8];  // per group of this work-group: [sub] wave arrivals so far, [2 + sub] last step released;
                                   // [4 + (wave & 3)]: matrix-phase token of the two waves that share a SIMD
    float *wt = lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4, hi = c >> 3, cu = c & 7;
    float *stg = lds + NC * LDW + wave * STG;
    if (tid < 8) relay[tid] = 0u;
    const int npairs = p.ngroups >> 1;
    const int pair = blockIdx.x % npairs, j0 = (blockIdx.x / npairs) * JB;
    const int sub = wave >> 2, rt = (wave >> 1) & 1, ch = wave & 1;  // waves w and w + 4 share a SIMD: one of each group
    const int group = 2 * pair + sub, row0 = group * 32 + rt * 16, unit = j0 + 8 * ch + cu;
    const unsigned narrive = (unsigned)(H / JB);  // arrivals per step and group: one per work-group (its 4 waves meet in LDS)
    const int Cn = p.Cn, R = p.R;
    const int rot = (int)(blockIdx.x / npairs) % NKB;  // staggered reduction start (see k_lstm_seq_fwd)
    unsigned *counter = p.sync + group * SEQ_SYNC_STRIDE2, *abort_flag = p.sync + SEQ_ABORT_SLOT;
    // ---- W_hh slice, transposed into LDS; local column lc = [unit half][tile][gate-of-tile][unit & 7]
    for (int idx = tid; idx < NC * H; idx += 512) {
        const int lc = idx % NC, k = idx / NC;
        const int q = 2 * ((lc >> 4) & 1) + ((lc >> 3) & 1), u = 8 * (lc >> 5) + (lc & 7);
        wt[lc * LDW + k] = p.whh[(int64_t)k * G4 + q * H + j0 + u];
    }
    float bias[NT], bih[NT];
    f32x4 bx[KXB > 0 ? KXB : 1][NT];  // lane (c, g): W_ih^T[this lane's gate column of tile nt][16*blk + 4*g ..]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = (2 * nt + hi) * H + unit;
        bias[nt] = p.bhh[col];
        bih[nt] = KXB > 0 ? p.bih[col] : 0.0f;
        if (KXB > 0) {
#pragma unroll
            for (int blk = 0; blk < KXB; ++blk)
                bx[blk][nt] = *reinterpret_cast<const f32x4 *>(p.wih_t + (int64_t)col * KX + 16 * blk + 4 * g);
        }
    }
    __syncthreads();  // the only block barrier of the pass
    const auto h_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.hprev, 0, (int)((int64_t)(R + 1) * Cn * H * 4), 0x00020000);
    // this lane finishes the elements (row er0 + e, unit), e = 0, 1: accumulator rows 2*hi + e of its 4-row strip
    const int er0 = row0 + 4 * g + 2 * hi;
    float cst[2], kp[2], xg[NT][4];
    f32x4 xa[KXB > 0 ? KXB : 1], xacc[NT];
#pragma unroll
    for (int e = 0; e < 2; ++e) cst[e] = p.cprev[(int64_t)(er0 + e) * H + unit];
    auto prefetch = [&](int t) {
#pragma unroll
        for (int e = 0; e < 2; ++e) kp[e] = p.keep[(int64_t)t * Cn + er0 + e];
        if constexpr (KXB == 0) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    xg[nt][i] = p.gx[((int64_t)t * Cn + row0 + 4 * g + i) * G4 + (2 * nt + hi) * H + unit];
        }
    };
    auto load_x = [&](int t) {
#pragma unroll
        for (int blk = 0; blk < KXB; ++blk)
            xa[blk] = *reinterpret_cast<const f32x4 *>(p.x + ((int64_t)t * Cn + row0 + c) * KX + 16 * blk + 4 * g);
    };
    prefetch(0);
    if (KXB > 0) load_x(0);

    for (int t = 0; t < R; ++t) {
        if (t > 0) {
            // one wave per group polls the global counter and releases the other three through LDS: per-wave polling
            // (2048 waves on 16 counters in four 128-byte lines) and per-wave arrivals cost 37 us per step, measured
            if ((wave & 3) == 0) {
                const bool ok = (p.ablate & 1) ? true : seq_wait_wave(counter, narrive * (unsigned)t, abort_flag);
                if (lane == 0) __hip_atomic_store(&relay[2 + sub], ok ? (unsigned)t : 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (!ok) return;
            } else {
                unsigned v;
                while ((v = (unsigned)__builtin_amdgcn_readfirstlane(
                            (int)__hip_atomic_load(&relay[2 + sub], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) < (unsigned)t)
                    __builtin_amdgcn_s_sleep(1);
                if (v == 0xFFFFFFFFu) return;
            }
        }
        f32x4 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const uint32_t abase = (p.ablate & 2) ? OOB : (uint32_t)((((int64_t)t * Cn + row0 + c) * H + 4 * g) * 4);
        i32x4 abuf[NBUF][KU];
        auto load_block = [&](int kb, i32x4 (&dst)[KU]) {
#pragma unroll
            for (int ku = 0; ku < KU; ++ku)
                dst[ku] = __builtin_amdgcn_raw_buffer_load_b128(h_rsrc, abase + (uint32_t)((kb * KU + ku) * 64), 0, SF_SEQ_LOAD_AUX);
        };
        auto kbe = [&](int kb) { const int k = kb + rot; return k >= NKB ? k - NKB : k; };
        load_block(kbe(0), abuf[0]);
        // The two waves of a SIMD take turns in the matrix phase: left alone they fall IN phase (both multiply, then both
        // run the cell and drain their stores: 15.6 us per step even without any hand-off wait, measured) — with the
        // token one wave's cell / drain / hand-off runs under the other's MFMAs.
        if (!(p.ablate & 32)) {
            __builtin_amdgcn_sched_barrier(0);
            for (;;) {
                unsigned got = 1u;
                if (lane == 0) got = __hip_atomic_exchange(&relay[4 + (wave & 3)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (__builtin_amdgcn_readfirstlane((int)got) == 0) break;
                __builtin_amdgcn_s_sleep(1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (KXB > 0) {  // x_t W_ih^T while the first h rows are on their way from L2
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) xacc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int blk = 0; blk < KXB; ++blk)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        xacc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[blk][j], bx[blk][nt][j], xacc[nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            if (kb + 1 < NKB) load_block(kbe(kb + 1), abuf[(kb + 1) % NBUF]);
            __builtin_amdgcn_sched_barrier(0);  // keep the next block's loads in front of this block's MFMAs
            const float *bpk = wt + (ch * 32 + c) * LDW + kbe(kb) * (KU * 16) + 4 * g;
            if (p.ablate & 4) continue;
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
        if (!(p.ablate & 32)) {
            __builtin_amdgcn_sched_barrier(0);
            if (lane == 0) __hip_atomic_store(&relay[4 + (wave & 3)], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- pre-activations of this lane's gate columns, then the partner half row's (lane ^ 8) by DPP
        float sown[NT][4], soth[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float gxv = KXB > 0 ? xacc[nt][i] + bih[nt] : xg[nt][i];
                sown[nt][i] = gxv + (acc[nt][i] + bias[nt]);
                soth[nt][i] = dpp_row_ror8(sown[nt][i]);
            }
        // ---- LSTM cell (k_rnn_cell_fwd's arithmetic) of the two elements; masked h for step t+1 into the staging tile
        float sv[2][6];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            // (by value: a conditional on two array ELEMENTS is an lvalue, i.e. a lane-dependent address — hipcc then
            //  keeps the arrays in scratch / LDS)
            const float own0 = selp(hi, sown[0][2 + e], sown[0][e]), oth0 = selp(hi, soth[0][2 + e], soth[0][e]);
            const float own1 = selp(hi, sown[1][2 + e], sown[1][e]), oth1 = selp(hi, soth[1][2 + e], soth[1][e]);
            const float ig = sigm(selp(hi, oth0, own0)), fg = sigm(selp(hi, own0, oth0));
            const float gg = tanhf(selp(hi, oth1, own1)), og = sigm(selp(hi, own1, oth1));
            const float cn = fg * cst[e] + ig * gg;
            const float h = og * tanhf(cn);
            cst[e] = cn * kp[e];
            stg[(4 * g + 2 * hi + e) * 8 + cu] = h * kp[e];
            sv[e][0] = ig; sv[e][1] = fg; sv[e][2] = gg; sv[e][3] = og; sv[e][4] = h; sv[e][5] = cn;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        {   // h_t * keep -> hprev[t+1]: 16 rows x 8 units = 32 x 16-byte write-through stores (the hand-off payload)
            const int r = (lane & 31) >> 1, c4 = lane & 1;
            const f32x4 val = *reinterpret_cast<const f32x4 *>(stg + r * 8 + c4 * 4);
            const uint32_t off = lane < 32 ? (uint32_t)((((int64_t)(t + 1) * Cn + row0 + r) * H + j0 + 8 * ch + c4 * 4) * 4) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, val), h_rsrc, off, 0, 16);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (t + 1 < R) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have drained
            if (lane == 0 && !(p.ablate & 16)) {  // the last of the group's four waves (LDS count) arrives for the work-group
                const unsigned old = __hip_atomic_fetch_add(&relay[sub], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (old == 4u * (unsigned)t + 3u) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            prefetch(t + 1);
            if (KXB > 0) load_x(t + 1);
        }
        // ---- saves for the backward pass (plain stores: they drain while this wave waits for the others)
        if (p.ablate & 8) continue;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int row = er0 + e;
            const int64_t tr = (int64_t)t * Cn + row;
            float *go = p.gates + tr * G4 + unit;
            go[0] = sv[e][0]; go[H] = sv[e][1]; go[2 * H] = sv[e][2]; go[3 * H] = sv[e][3];
            p.hout[(int64_t)row * p.ho_rs + (int64_t)t * p.ho_ts + unit] = sv[e][4];
            p.cout[tr * H + unit] = sv[e][5];
            p.cprev[(tr + Cn) * H + unit] = cst[e];
        }
    }
}


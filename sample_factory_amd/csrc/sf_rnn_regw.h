// Backward sequence passes with the W_hh slice in REGISTERS (included by sf_rnn.hip, inside its anonymous namespace).
//
// k_lstm_seq_bwd / k_gru_seq_bwd keep 16 hidden units' rows of W_hh in LDS and every work-group of a row group pulls the
// group's complete gate-gradient rows (64 rows x 4H floats = 512 KB per step at H = 512) through L2: measured, that
// load stream is what the backward step waits for (tools/lstm_bench.py: 19.4 us per step, 13.5 without the loads,
// 17.2 without the MFMAs — the matrix pipe is hidden behind the loads, not the other way round).  The bytes a
// work-group pulls are rows_per_group x K, and rows_per_group = Cn * (H / JB) / #work-groups: twice the hidden units per
// work-group halve them.  32 units x 4H gate columns are 256 KB of weights — more than LDS, but not more than the
// register file (4 waves x 64 lanes x 512 VGPRs = 512 KB per CU with one wave per SIMD):
//   * wave w owns the gate columns [w*K/4, (w+1)*K/4) of the reduction for ALL rows and ALL 32 units of the work-group;
//     its B fragments (K/4 x 32 weights = 256 VGPRs at H = 512, K = 4H) are loaded once and stay in registers: the
//     k-loop has no LDS traffic at all (one 16-byte global load per 8 MFMAs);
//   * the four partial sums of a (row, unit) element meet in LDS and are added in wave order (fixed);
//   * the cell backward (phase A) of the 16-row x 16-unit tiles is dealt to the waves (tile ti -> wave ti % 4), carries
//     in registers as before.
// Row groups are 32 (RT = 2) or 64 (RT = 4) rows, 16 groups of H/32 work-groups each; hand-off protocol as in sf_rnn.hip
// (write-through stores, counter per group — 8 words apart here, 16 groups fit in front of the abort word).
constexpr int SEQ_SYNC_STRIDE_R = 8;
constexpr int SEQ_MAX_GROUPS_R = SEQ_ABORT_SLOT / SEQ_SYNC_STRIDE_R;

// LSTM at H = 512: 256 weight registers + the phase-A operands of the tiles do not fit 512 VGPRs (hipcc spills); the
// fragments of every 4th 16-column block (RT = 4: 3 of 8) live in LDS instead, lane-linear (one conflict-free
// ds_read_b128 each, issued at the top of the 64-MFMA block that consumes them).
template <int H, int RT>
struct RegwLds {
    static constexpr bool in_lds(int blk) { return H == 512 && (blk % 4 == 3 || (RT == 4 && blk % 8 == 1)); }
    static constexpr int slot(int blk) {
        int n = 0;
        for (int b = 0; b < blk; ++b) n += in_lds(b) ? 1 : 0;
        return n;
    }
};

template <int H, int RT>
__global__ __launch_bounds__(256, 1) void k_lstm_seq_bwd_r(LstmSeqBwd p) {
    constexpr int JB = 32, G4 = 4 * H, KW = G4 / 4, NBLK = KW / 16, TPW = RT / 2;
    constexpr int KU = 8, NKB = NBLK / KU, S = RT * NKB;
    constexpr int RED = RT * 16 * JB, STG = 16 * 64;
    typedef RegwLds<H, RT> WL;
    constexpr int NBL = WL::slot(NBLK);  // blocks per wave whose fragments live in LDS
    static_assert(RT % 2 == 0 && NBLK % KU == 0, "shape");
    __shared__ __attribute__((aligned(16))) float lds[4 * RED + 4 * STG + 4 + 4 * NBL * 2 * 256];
    float *red = lds, *flag = lds + 4 * RED + 4 * STG;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c = lane & 15, g = lane >> 4;
    float *stg = lds + 4 * RED + wave * STG;
    const int group = blockIdx.x % p.ngroups, j0 = (blockIdx.x / p.ngroups) * JB;
    const unsigned ncol = H / JB;
    const int Cn = p.Cn, R = p.R;
    unsigned *counter = p.sync + group * SEQ_SYNC_STRIDE_R, *abort_flag = p.sync + SEQ_ABORT_SLOT;
    // ---- B fragments: lane (c, g) holds W_hh[unit j0 + 16*ut + c][gate column wave*KW + 16*blk + 4*g + j]
    f32x4 breg[NBLK][2];
    float *wl = lds + 4 * RED + 4 * STG + 4 + wave * (NBL * 2 * 256) + lane * 4;  // [slot][ut][lane] f32x4
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
        for (int ut = 0; ut < 2; ++ut) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(p.whh + (int64_t)(j0 + 16 * ut + c) * G4 + wave * KW + 16 * blk + 4 * g);
            if (WL::in_lds(blk)) *reinterpret_cast<f32x4 *>(wl + (WL::slot(blk) * 2 + ut) * 256) = wv;
            else breg[blk][ut] = wv;
        }
    const auto d_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.dgx, 0, (int)((int64_t)R * Cn * G4 * 4), 0x00020000);
    const int g_row0 = group * p.rows_per_group;
    const int g_rows_end = min(Cn, g_row0 + p.rows_per_group);

    // this wave's tiles: ti = wave + 4*q -> row tile rt = ti >> 1, unit tile ut = ti & 1; elements (row 4g+i, unit c)
    float car_h[TPW][4], car_c[TPW][4];
#pragma unroll
    for (int q = 0; q < TPW; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) car_h[q][i] = car_c[q][i] = 0.0f;
    float pg[TPW][4][4], pdo[TPW][4], pco[TPW][4], pcp[TPW][4], pkp[TPW][4];
    auto prefetch = [&](int t) {
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            const int ti = wave + 4 * q, row0 = g_row0 + (ti >> 1) * 16, j = j0 + 16 * (ti & 1) + c;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 4 * g + i, r = row < g_rows_end ? row : g_rows_end - 1;
                const int64_t tr = (int64_t)t * Cn + r;
                pkp[q][i] = t > 0 ? p.keep[tr - Cn] : 0.0f;
                const float *go = p.gates + tr * G4 + j;
                pg[q][i][0] = go[0]; pg[q][i][1] = go[H]; pg[q][i][2] = go[2 * H]; pg[q][i][3] = go[3 * H];
                pdo[q][i] = p.dout[(int64_t)r * p.do_rs + (int64_t)t * p.do_ts + j];
                pco[q][i] = p.cout[tr * H + j];
                pcp[q][i] = p.cprev[tr * H + j];
            }
        }
    };
    prefetch(R - 1);

    for (int s = 0; s < R; ++s) {
        const int t = R - 1 - s;
        // ---- phase A: cell backward (k_rnn_cell_bwd's arithmetic) of this wave's tiles; dgates -> dgx[t] (the payload)
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            const int ti = wave + 4 * q, row0 = g_row0 + (ti >> 1) * 16, ut = ti & 1;
            if (row0 >= g_rows_end) continue;  // (wave-uniform)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float ig = pg[q][i][0], fg = pg[q][i][1], gg = pg[q][i][2], og = pg[q][i][3];
                float d = pdo[q][i];
                float dc_in = 0.0f;
                if (s > 0) {
                    d = d + car_h[q][i];
                    dc_in = car_c[q][i];
                }
                const float tc = tanh_c(pco[q][i]);
                const float dc = d * og * (1.0f - tc * tc) + dc_in;
                const float di = (dc * gg) * (ig * (1.0f - ig)), df = (dc * pcp[q][i]) * (fg * (1.0f - fg));
                const float dg = (dc * ig) * (1.0f - gg * gg), dob = (d * tc) * (og * (1.0f - og));
                float *sp = stg + (4 * g + i) * 64 + c;
                sp[0] = di; sp[16] = df; sp[32] = dg; sp[48] = dob;
                car_c[q][i] = (dc * fg) * pkp[q][i];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int v = 0; v < 4; ++v) {  // 16 rows x 64 floats as 16-byte write-through stores
                const int f = v * 64 + lane, r = f >> 4, c4 = f & 15, row = row0 + r;
                const f32x4 val = *reinterpret_cast<const f32x4 *>(stg + r * 64 + c4 * 4);
                const uint32_t off = row < g_rows_end
                    ? (uint32_t)((((int64_t)t * Cn + row) * G4 + (c4 >> 2) * H + j0 + 16 * ut + (c4 & 3) * 4) * 4) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, val), d_rsrc, off, 0, 16);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (t == 0) break;  // no state in front of step 0
        seq_arrive(counter);
        float kcur[TPW][4];
#pragma unroll
        for (int q = 0; q < TPW; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) kcur[q][i] = pkp[q][i];  // keep[t-1] of THIS step (prefetch overwrites pkp)
        prefetch(t - 1);
        if (!(p.ablate & 1) && !seq_wait(counter, ncol * (unsigned)(s + 1), abort_flag, flag)) return;
        // ---- phase B: partial dL/dh_{t-1}[all rows, 32 units] over this wave's quarter of the gate columns
        f32x4 acc[RT][2];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt][0] = acc[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        i32x4 abuf[2][KU];
        auto load_block = [&](int sidx, i32x4 (&dst)[KU]) {  // sidx = rt * NKB + kb
            const int rt = sidx / NKB, kb = sidx % NKB, arow = g_row0 + rt * 16 + c;
            const uint32_t abase = (arow < g_rows_end && !(p.ablate & 2))
                ? (uint32_t)((((int64_t)t * Cn + arow) * G4 + wave * KW + kb * KU * 16 + 4 * g) * 4) : OOB;
#pragma unroll
            for (int ku = 0; ku < KU; ++ku)
                dst[ku] = __builtin_amdgcn_raw_buffer_load_b128(d_rsrc, abase + (uint32_t)(ku * 64), 0, 16);
        };
        load_block(0, abuf[0]);
#pragma unroll
        for (int sidx = 0; sidx < S; ++sidx) {
            if (sidx + 1 < S) load_block(sidx + 1, abuf[(sidx + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);  // keep the next block's loads in FRONT of this block's MFMAs (hipcc sinks them to their use)
            const int rt = sidx / NKB, kb = sidx % NKB;
#pragma unroll
            for (int ku = 0; ku < KU; ++ku) {
                const f32x4 a4 = __builtin_bit_cast(f32x4, abuf[sidx & 1][ku]);
                const int blk = kb * KU + ku;
#pragma unroll
                for (int ut = 0; ut < 2; ++ut) {
                    const f32x4 b4 = WL::in_lds(blk) ? *reinterpret_cast<const f32x4 *>(wl + (WL::slot(blk) * 2 + ut) * 256)
                                                     : breg[blk][ut];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[rt][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], b4[j], acc[rt][ut], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- the four partials of every (row, unit) meet in LDS; the tile's owner adds them in wave order
        float *rw = red + wave * RED;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ut = 0; ut < 2; ++ut)
#pragma unroll
                for (int i = 0; i < 4; ++i) rw[(rt * 16 + 4 * g + i) * JB + 16 * ut + c] = acc[rt][ut][i];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            const int ti = wave + 4 * q;
            const float *rp = red + ((ti >> 1) * 16 + 4 * g) * JB + 16 * (ti & 1) + c;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float sum = ((rp[i * JB] + rp[RED + i * JB]) + rp[2 * RED + i * JB]) + rp[3 * RED + i * JB];
                car_h[q][i] = sum * kcur[q][i];
            }
        }
        // (the next write to `red` sits behind the barrier of the next seq_arrive)
    }
}

// GRU: payload dgh = {dr, dz, dn * r} ([R][Cn][3H]), K = 3H gate columns, wave w reduces columns [w*3H/4, (w+1)*3H/4)
template <int H, int RT>
__global__ __launch_bounds__(256, 1) void k_gru_seq_bwd_r(GruSeqBwd p) {
    constexpr int JB = 32, G3 = 3 * H, G4 = 4 * H, KW = G3 / 4, NBLK = KW / 16, TPW = RT / 2;
    constexpr int KU = NBLK % 8 == 0 ? 8 : 4, NKB = NBLK / KU, S = RT * NKB;
    constexpr int RED = RT * 16 * JB, STG = 16 * 48;
    static_assert(RT % 2 == 0 && NBLK % KU == 0 && KW % 16 == 0, "shape");
    __shared__ __attribute__((aligned(16))) float lds[4 * RED + 4 * STG + 4];
    float *red = lds, *flag = lds + 4 * RED + 4 * STG;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c = lane & 15, g = lane >> 4;
    float *stg = lds + 4 * RED + wave * STG;
    const int group = blockIdx.x % p.ngroups, j0 = (blockIdx.x / p.ngroups) * JB;
    const unsigned ncol = H / JB;
    const int Cn = p.Cn, R = p.R;
    unsigned *counter = p.sync + group * SEQ_SYNC_STRIDE_R, *abort_flag = p.sync + SEQ_ABORT_SLOT;
    f32x4 breg[NBLK][2];
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk)
#pragma unroll
        for (int ut = 0; ut < 2; ++ut)
            breg[blk][ut] = *reinterpret_cast<const f32x4 *>(p.whh + (int64_t)(j0 + 16 * ut + c) * G3 + wave * KW + 16 * blk + 4 * g);
    const auto d_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.dgh, 0, (int)((int64_t)R * Cn * G3 * 4), 0x00020000);
    const int g_row0 = group * p.rows_per_group;
    const int g_rows_end = min(Cn, g_row0 + p.rows_per_group);

    float car_h[TPW][4];
#pragma unroll
    for (int q = 0; q < TPW; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) car_h[q][i] = 0.0f;
    float pg[TPW][4][4], pdo[TPW][4], php[TPW][4], pkp[TPW][4];
    auto prefetch = [&](int t) {
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            const int ti = wave + 4 * q, row0 = g_row0 + (ti >> 1) * 16, j = j0 + 16 * (ti & 1) + c;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 4 * g + i, r = row < g_rows_end ? row : g_rows_end - 1;
                const int64_t tr = (int64_t)t * Cn + r;
                pkp[q][i] = t > 0 ? p.keep[tr - Cn] : 0.0f;
                const float *go = p.gates + tr * G4 + j;
                pg[q][i][0] = go[0]; pg[q][i][1] = go[H]; pg[q][i][2] = go[2 * H]; pg[q][i][3] = go[3 * H];
                pdo[q][i] = p.dout[(int64_t)r * p.do_rs + (int64_t)t * p.do_ts + j];
                php[q][i] = p.hprev[tr * H + j];
            }
        }
    };
    prefetch(R - 1);

    for (int s = 0; s < R; ++s) {
        const int t = R - 1 - s;
        float dir[TPW][4];  // dL/dh_prev that does not go through W_hh: dh * z
        // ---- phase A: GRU cell backward (k_rnn_cell_bwd's arithmetic); dgx plain, dgh = the hand-off payload
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            const int ti = wave + 4 * q, row0 = g_row0 + (ti >> 1) * 16, ut = ti & 1;
            if (row0 >= g_rows_end) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 4 * g + i;
                const bool ok = row < g_rows_end;
                const int64_t tr = (int64_t)t * Cn + (ok ? row : g_rows_end - 1);
                const float r = pg[q][i][0], z = pg[q][i][1], n = pg[q][i][2], hn = pg[q][i][3];
                float d = pdo[q][i];
                if (s > 0) d = d + car_h[q][i];
                const float dn_pre = (d * (1.0f - z)) * (1.0f - n * n);
                const float dz_pre = (d * (php[q][i] - n)) * (z * (1.0f - z));
                const float dr_pre = (dn_pre * hn) * (r * (1.0f - r));
                float *sp = stg + (4 * g + i) * 48 + c;
                sp[0] = dr_pre; sp[16] = dz_pre; sp[32] = dn_pre * r;
                dir[q][i] = d * z;
                if (ok) {
                    float *x = p.dgx + tr * G3 + j0 + 16 * ut + c;
                    x[0] = dr_pre; x[H] = dz_pre; x[2 * H] = dn_pre;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int v = 0; v < 3; ++v) {  // 16 rows x 48 floats as 16-byte write-through stores
                const int f = v * 64 + lane, r = f / 12, c4 = f % 12, row = row0 + r;
                const f32x4 val = *reinterpret_cast<const f32x4 *>(stg + r * 48 + c4 * 4);
                const uint32_t off = row < g_rows_end
                    ? (uint32_t)((((int64_t)t * Cn + row) * G3 + (c4 >> 2) * H + j0 + 16 * ut + (c4 & 3) * 4) * 4) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4, val), d_rsrc, off, 0, 16);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (t == 0) break;
        seq_arrive(counter);
        float kcur[TPW][4];
#pragma unroll
        for (int q = 0; q < TPW; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) kcur[q][i] = pkp[q][i];
        prefetch(t - 1);
        if (!seq_wait(counter, ncol * (unsigned)(s + 1), abort_flag, flag)) return;
        // ---- phase B: partial dgh_t[rows, own quarter of the columns] W_hh[32 units, the same columns]^T
        f32x4 acc[RT][2];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) acc[rt][0] = acc[rt][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        i32x4 abuf[2][KU];
        auto load_block = [&](int sidx, i32x4 (&dst)[KU]) {
            const int rt = sidx / NKB, kb = sidx % NKB, arow = g_row0 + rt * 16 + c;
            const uint32_t abase = arow < g_rows_end
                ? (uint32_t)((((int64_t)t * Cn + arow) * G3 + wave * KW + kb * KU * 16 + 4 * g) * 4) : OOB;
#pragma unroll
            for (int ku = 0; ku < KU; ++ku)
                dst[ku] = __builtin_amdgcn_raw_buffer_load_b128(d_rsrc, abase + (uint32_t)(ku * 64), 0, 16);
        };
        load_block(0, abuf[0]);
#pragma unroll
        for (int sidx = 0; sidx < S; ++sidx) {
            if (sidx + 1 < S) load_block(sidx + 1, abuf[(sidx + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);  // keep the next block's loads in FRONT of this block's MFMAs (hipcc sinks them to their use)
            const int rt = sidx / NKB, kb = sidx % NKB;
#pragma unroll
            for (int ku = 0; ku < KU; ++ku) {
                const f32x4 a4 = __builtin_bit_cast(f32x4, abuf[sidx & 1][ku]);
#pragma unroll
                for (int ut = 0; ut < 2; ++ut)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[rt][ut] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[j], breg[kb * KU + ku][ut][j], acc[rt][ut], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        float *rw = red + wave * RED;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int ut = 0; ut < 2; ++ut)
#pragma unroll
                for (int i = 0; i < 4; ++i) rw[(rt * 16 + 4 * g + i) * JB + 16 * ut + c] = acc[rt][ut][i];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < TPW; ++q) {
            const int ti = wave + 4 * q;
            const float *rp = red + ((ti >> 1) * 16 + 4 * g) * JB + 16 * (ti & 1) + c;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float sum = ((rp[i * JB] + rp[RED + i * JB]) + rp[2 * RED + i * JB]) + rp[3 * RED + i * JB];
                car_h[q][i] = (sum + dir[q][i]) * kcur[q][i];
            }
        }
    }
}

// 16 row groups of H/32 work-groups; a row group is 32 or 64 rows (RT = 2 / 4 row tiles)
int seq_plan_r(int Cn, int H, int *ngroups, int *rows_per_group) {
    static const int on = getenv("SF_SEQ_BWD_REGW") ? atoi(getenv("SF_SEQ_BWD_REGW")) : 1;
    if (!on || (H != 512 && H != 256)) return 0;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return 0;
    const int ncol = H / 32;
    int ng = cus / ncol;
    if (ng > SEQ_MAX_GROUPS_R) ng = SEQ_MAX_GROUPS_R;
    const int need = (Cn + 31) / 32;
    if (ng > need) ng = need;
    if (ng < 1) return 0;
    *ngroups = ng;
    *rows_per_group = ((Cn + ng - 1) / ng + 31) / 32 * 32;
    return *rows_per_group <= 64;
}

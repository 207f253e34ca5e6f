// Forward of NARROW linear layers (Cout <= 32: the fused value / action-parameter heads, 512 -> 8 or 20), included by
// sf_nn.hip.  out[m][n] = act(sum_k x[m][k] * wt[n][k] + bias[n]), wt = the Cout-major weight copy.
//
// The tiled kernel gives such a layer one 128-row work-group per 128 rows — 16 to 32 work-groups for a rollout step —
// and each walks K in 32-chunks with a global-load -> LDS -> MFMA round trip per chunk: 13 us (4096 x 512 x 8) and 23 us
// (2048 x 512 x 20) for 4 - 8 MB of input, launched once per rollout step.  Here ONE WAVE owns 16 rows: it fetches
// its operands straight into MFMA fragments (lane (c, g): 16 bytes of row / weight column c at k = 16*blk + 4*g — both
// operands are k-contiguous, so the k-permutation inside a 16-block is shared), a whole 256-deep chunk of loads in
// flight at once and the next chunk behind it, no LDS, no barrier: the launch is one memory round trip deep.
// Grid = ceil(n / 16) single-wave work-groups (256 for a 4096-env rollout step: one per CU).
#pragma once

template <int NT>
__global__ __launch_bounds__(64) void k_linear_narrow(const float *__restrict__ in, int64_t in_stride,
                                                      const float *__restrict__ wt, const float *__restrict__ bias,
                                                      float *__restrict__ out, int n, int N, int K, int kind) {
    constexpr int CB = 16;  // 16-column blocks per chunk (256 reduction elements)
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    const int row0 = (int)blockIdx.x * 16;
    const int arow = row0 + c < n ? row0 + c : n - 1;
    const float *ap = in + (int64_t)arow * in_stride + 4 * g;
    const float *bp[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = 16 * nt + c;
        bp[nt] = wt + (int64_t)(col < N ? col : N - 1) * K + 4 * g;
    }
    f32x4 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 a[2][CB], b[2][NT][CB];
    auto load = [&](int k0, f32x4 (&av)[CB], f32x4 (&bv)[NT][CB]) {
#pragma unroll
        for (int blk = 0; blk < CB; ++blk) {
            const int kk = k0 + 16 * blk;
            const bool ok = kk < K;  // (uniform) K is a multiple of 16, not necessarily of 256
            av[blk] = ok ? *reinterpret_cast<const f32x4 *>(ap + kk) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                bv[nt][blk] = ok ? *reinterpret_cast<const f32x4 *>(bp[nt] + kk) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto mma = [&](const f32x4 (&av)[CB], const f32x4 (&bv)[NT][CB]) {
#pragma unroll
        for (int blk = 0; blk < CB; ++blk)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[blk][j], bv[nt][blk][j], acc[nt], 0, 0, 0);
    };
    load(0, a[0], b[0]);
    for (int k0 = 0; k0 < K; k0 += 2 * CB * 16) {
        if (k0 + CB * 16 < K) load(k0 + CB * 16, a[1], b[1]);
        mma(a[0], b[0]);
        if (k0 + CB * 16 < K) {
            if (k0 + 2 * CB * 16 < K) load(k0 + 2 * CB * 16, a[0], b[0]);
            mma(a[1], b[1]);
        }
    }
    // C/D: element i of lane (c, g) = (row 4*g + i, column c)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int col = 16 * nt + c;
        if (col >= N) continue;
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + 4 * g + i;
            if (row < n) out[(int64_t)row * N + col] = kind == 0 ? acc[nt][i] + bv : act_fwd(acc[nt][i] + bv, kind);
        }
    }
}

// Weight gradient of SMALL linear layers (K <= 64 inputs, N <= 64 outputs: the MLP encoder of vector observations,
// 27 -> 64 -> 64): dW[k][n] = sum_m x[m][k] * dy[m][n], a 27 x 64 result reduced over 16384 rows.  The tiled MFMA kernel
// spends 66 us (K = 27, scalar loader) / 41 us (K = 64) on 57 / 134 MFLOP.  Here a work-group takes a run of rows in
// 64-row tiles through LDS; thread (q = t >> 6, n = t & 63) owns the 16 outputs k = 16q .. 16q+15 of column n: per row
// one conflict-free read of dy and four broadcast ds_read_b128 of x feed 16 fmaf.  One partial per work-group, summed
// in fixed order by k_reduce_partials like every other weight-gradient kernel; column sums of dy = the bias gradient.
__global__ __launch_bounds__(256) void k_linear_wgrad_small(const float *__restrict__ x, int64_t x_stride,
                                                            const float *__restrict__ dy, float *__restrict__ partial_w,
                                                            float *__restrict__ partial_b, int64_t M, int64_t m_per_split,
                                                            int K, int N) {
    constexpr int RT = 64, XP = 68;
    __shared__ __attribute__((aligned(16))) float sx[RT * XP];
    __shared__ float sdy[RT * 64];
    const int t = threadIdx.x, n = t & 63, q = t >> 6;
    const int64_t m0 = (int64_t)blockIdx.x * m_per_split, m1 = m0 + m_per_split < M ? m0 + m_per_split : M;
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float bsum = 0.f;
    for (int64_t mc = m0; mc < m1; mc += RT) {
        const int rows = (int)(m1 - mc < RT ? m1 - mc : RT);
        // all 32 global loads of the tile pair are issued before the first LDS store (a load -> store loop waits for
        // every load in turn: 32 serialised round trips per tile)
        float xv[16], dv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int idx = t + 256 * i, r = idx >> 6, c = idx & 63;
            const int64_t row = mc + (r < rows ? r : 0);
            xv[i] = x[row * x_stride + (c < K ? c : 0)];
            dv[i] = dy[row * N + (c < N ? c : 0)];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int idx = t + 256 * i, r = idx >> 6, c = idx & 63;
            sx[r * XP + c] = (r < rows && c < K) ? xv[i] : 0.f;  // zero-filled past K / N / the last row
            sdy[idx] = (r < rows && c < N) ? dv[i] : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int r = 0; r < RT; ++r) {
            const float d = sdy[r * 64 + n];
            bsum += d;
            const f32x4 *xr = reinterpret_cast<const f32x4 *>(sx + r * XP + 16 * q);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const f32x4 xv = xr[v];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[4 * v + j] = fmaf(xv[j], d, acc[4 * v + j]);
            }
        }
        __syncthreads();
    }
    if (n < N) {
        float *dst = partial_w + (int64_t)blockIdx.x * K * N + n;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (16 * q + i < K) dst[(int64_t)(16 * q + i) * N] = acc[i];
        if (partial_b && q == 0) partial_b[(int64_t)blockIdx.x * N + n] = bsum;
    }
}

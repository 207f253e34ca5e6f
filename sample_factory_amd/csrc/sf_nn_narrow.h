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

// sf_rl.hip — the HBM/latency-bound part of the APPO hot path for gfx950 (MI355X):
// validity mask, GAE/returns scan, running-mean-std, V-trace, fused PPO loss fwd+bwd, minibatch index sets,
// grad-norm + Adam, action sampling, trajectory-step writes and the synthetic vector env.
// Reference op sequences replaced by each kernel are cited in include/sf_hip.h.
//
// Compiled with -ffp-contract=off: these kernels keep the reference's fp32 op order (torch CPU/GPU elementwise ops
// round after every op), so results match the oracle to the last bit wherever libm does.
#include "sf_common.h"

thread_local char sf_err_buf[512] = "";

extern "C" const char *sf_last_error(void) { return sf_err_buf; }
extern "C" int sf_abi_version(void) { return 19; }

#define STREAM(s) reinterpret_cast<hipStream_t>(s)

__device__ __forceinline__ float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// =========================================================================================== K13 validity mask
__global__ __launch_bounds__(256) void k_valid_mask(const int32_t *__restrict__ policy_id,
                                                    const float *__restrict__ policy_version,
                                                    uint8_t *__restrict__ valids, uint8_t *__restrict__ valids_flat,
                                                    float *__restrict__ actions,
                                                    int num_actions, float *__restrict__ logp, int E, int T,
                                                    int my_pid, int train_step, int max_lag,
                                                    int32_t *__restrict__ num_invalid) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t N = (int64_t)E * T;
    const bool in = i < N;
    bool valid = true;
    if (in) {
        const int64_t e = i / T;
        const int t = (int)(i - e * T);
        valid = (policy_id[i] == my_pid) && (((float)train_step - policy_version[i]) < (float)max_lag);
        valids[e * (T + 1) + t] = (uint8_t)valid;
        if (valids_flat) valids_flat[i] = (uint8_t)valid;  // the [E*T] dataset view (learner.py:1009-1012) without a compaction copy
        if (t == T - 1) valids[e * (T + 1) + T] = (uint8_t)valid;  // learner.py:955
        if (!valid) {
            for (int a = 0; a < num_actions; ++a) actions[i * num_actions + a] = 0.0f;
            logp[i] = -1.0f;
        }
    }
    const unsigned long long b = __ballot(in && !valid);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(num_invalid, (int32_t)__popcll(b));
}

extern "C" int sf_valid_mask(const int32_t *policy_id, const float *policy_version, uint8_t *valids,
                             uint8_t *valids_flat, float *actions, int num_actions, float *log_prob_actions, int E, int T,
                             int my_policy_id, int train_step, int max_policy_lag, int32_t *num_invalid, void *stream) {
    SF_REQUIRE(E > 0 && T > 0 && num_actions > 0, "sf_valid_mask: bad shape E=%d T=%d na=%d", E, T, num_actions);
    SF_REQUIRE(policy_id && policy_version && valids && actions && log_prob_actions && num_invalid,
               "sf_valid_mask: null pointer");
    int rc = sf_hip_status(hipMemsetAsync(num_invalid, 0, sizeof(int32_t), STREAM(stream)), "sf_valid_mask memset");
    if (rc) return rc;
    const int64_t N = (int64_t)E * T;
    k_valid_mask<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, STREAM(stream)>>>(
        policy_id, policy_version, valids, valids_flat, actions, num_actions, log_prob_actions, E, T, my_policy_id,
        train_step, max_policy_lag, num_invalid);
    return sf_launch_status("sf_valid_mask");
}

// =========================================================================================== K10+K11 GAE / returns
// A 256-thread block owns 64 consecutive envs.  The env-major [E,T] rows are staged through LDS in [64 x 32]-step tiles
// (row stride 33 words -> the per-lane row walk of the scan is bank-conflict-free): all four waves issue the tile's
// loads as coalesced runs of consecutive words — every load of a tile is in flight before the first LDS write (the
// one-wave version waited for each load in turn: 330 serialised round trips, 68 us at (4096, 32)) — wave 0 runs the
// time recursion with one register per lane (= env) in the reference's op order, all four waves store the results.
constexpr int GAE_TC = 32;
constexpr int GAE_LD = GAE_TC + 1;
constexpr int GAE_NT = 256;
constexpr int GAE_IT = (64 * GAE_LD + GAE_NT - 1) / GAE_NT;  // loads per thread and array: 9

struct GaeTile {
    float v[GAE_IT];
};
__device__ __forceinline__ GaeTile gae_fetch_f32(const float *g, int64_t row_stride, int e0, int E, int t0, int ncols,
                                                 int tid) {
    GaeTile r;
#pragma unroll
    for (int i = 0; i < GAE_IT; ++i) {
        const int idx = i * GAE_NT + tid;
        const int row = idx / ncols, col = idx - row * ncols;
        const int e = e0 + row;
        const bool ok = row < 64 && e < E;
        const float x = g[ok ? (int64_t)e * row_stride + t0 + col : 0];  // branch-free: clamped address, late select
        r.v[i] = ok ? x : 0.0f;
    }
    return r;
}
__device__ __forceinline__ GaeTile gae_fetch_u8(const uint8_t *g, int64_t row_stride, int e0, int E, int t0, int ncols,
                                                int tid) {
    GaeTile r;
#pragma unroll
    for (int i = 0; i < GAE_IT; ++i) {
        const int idx = i * GAE_NT + tid;
        const int row = idx / ncols, col = idx - row * ncols;
        const int e = e0 + row;
        const bool ok = row < 64 && e < E;
        const uint8_t x = g[ok ? (int64_t)e * row_stride + t0 + col : 0];
        r.v[i] = (ok && x) ? 1.0f : 0.0f;
    }
    return r;
}
__device__ __forceinline__ void gae_put(float (*s)[GAE_LD], const GaeTile &r, int ncols, int tid) {
#pragma unroll
    for (int i = 0; i < GAE_IT; ++i) {
        const int idx = i * GAE_NT + tid;
        const int row = idx / ncols, col = idx - row * ncols;
        if (row < 64) s[row][col] = r.v[i];
    }
}
__device__ __forceinline__ void gae_tile_store_f32(const float (*s)[GAE_LD], float *g, int64_t row_stride, int e0,
                                                   int E, int t0, int ncols, int tid) {
#pragma unroll
    for (int i = 0; i < GAE_IT; ++i) {
        const int idx = i * GAE_NT + tid;
        const int row = idx / ncols, col = idx - row * ncols;
        const int e = e0 + row;
        if (row < 64 && e < E) g[(int64_t)e * row_stride + t0 + col] = s[row][col];
    }
}

__global__ __launch_bounds__(GAE_NT) void k_gae_returns(float *__restrict__ rewards, const uint8_t *__restrict__ dones,
                                                        const uint8_t *__restrict__ time_outs,
                                                        const float *__restrict__ values,
                                                        const uint8_t *__restrict__ valids,
                                                        const double *__restrict__ rms_stats, int E, int T, float gamma,
                                                        float gl, int value_bootstrap, float *__restrict__ adv_out,
                                                        float *__restrict__ ret_out) {
    __shared__ float s_r[64][GAE_LD], s_v[64][GAE_LD], s_d[64][GAE_LD], s_to[64][GAE_LD], s_va[64][GAE_LD],
        s_adv[64][GAE_LD], s_ret[64][GAE_LD];
    const int tid = threadIdx.x, lane = tid;  // the scan runs on wave 0: lane == tid < 64
    const int e0 = blockIdx.x * 64;
    const bool denorm = rms_stats != nullptr;
    float mu = 0.f, sigma = 1.f;
    if (denorm) {
        mu = (float)rms_stats[0];
        sigma = sqrtf((float)rms_stats[1] + 1e-5f);
    }
    float cum = 0.0f;
    for (int t0 = ((T - 1) / GAE_TC) * GAE_TC; t0 >= 0; t0 -= GAE_TC) {
        const int tc = min(GAE_TC, T - t0);
        // every load of the tile is issued before the first LDS write
        const GaeTile fr = gae_fetch_f32(rewards, T, e0, E, t0, tc, tid);
        const GaeTile fd = gae_fetch_u8(dones, T, e0, E, t0, tc, tid);
        const GaeTile fv = gae_fetch_f32(values, T + 1, e0, E, t0, tc + 1, tid);
        const GaeTile fa = gae_fetch_u8(valids, T + 1, e0, E, t0, tc + 1, tid);
        GaeTile ft;
        if (value_bootstrap) ft = gae_fetch_u8(time_outs, T, e0, E, t0, tc, tid);
        __syncthreads();  // the previous tile's stores have read s_adv / s_ret / s_r
        gae_put(s_r, fr, tc, tid);
        gae_put(s_d, fd, tc, tid);
        gae_put(s_v, fv, tc + 1, tid);
        gae_put(s_va, fa, tc + 1, tid);
        if (value_bootstrap) gae_put(s_to, ft, tc, tid);
        __syncthreads();
        if (tid < 64) {
            // per-lane backward scan over this tile (lane = env)
            for (int c = tc; c >= 0; --c) {  // de-normalise the values of this row first (learner.py:969-979)
                float v = s_v[lane][c];
                if (denorm) v = clampf(v, -5.0f, 5.0f) * sigma + mu;
                s_v[lane][c] = v;
            }
            for (int c = tc - 1; c >= 0; --c) {
                const float v = s_v[lane][c], vn = s_v[lane][c + 1];
                const float valid = s_va[lane][c], valid_n = s_va[lane][c + 1];
                const float done = s_d[lane][c];
                float r = s_r[lane][c];
                if (value_bootstrap) {  // learner.py:990
                    r = r + ((gamma * v) * s_to[lane][c]) * done;
                    s_r[lane][c] = r;
                }
                const float a = (r - v) * valid;
                const float b = (1.0f - done) * ((gamma * vn) * valid_n);
                const float delta = a + b;
                const float disc = gl * valid + (1.0f - valid);
                cum = delta + (disc * cum) * (1.0f - done);
                s_adv[lane][c] = cum;
                s_ret[lane][c] = cum + valid * v;  // learner.py:1003
            }
        }
        __syncthreads();
        gae_tile_store_f32(s_adv, adv_out, T, e0, E, t0, tc, tid);
        gae_tile_store_f32(s_ret, ret_out, T, e0, E, t0, tc, tid);
        if (value_bootstrap) gae_tile_store_f32(s_r, rewards, T, e0, E, t0, tc, tid);
    }
}

extern "C" int sf_gae_returns(float *rewards, const uint8_t *dones, const uint8_t *time_outs, const float *values,
                              const uint8_t *valids, const double *rms_stats, int E, int T, float gamma,
                              float gae_lambda, int value_bootstrap, float *advantages, float *returns,
                              void *stream) {
    SF_REQUIRE(E > 0 && T > 0, "sf_gae_returns: bad shape E=%d T=%d", E, T);
    SF_REQUIRE(rewards && dones && values && valids && advantages && returns, "sf_gae_returns: null pointer");
    SF_REQUIRE(!value_bootstrap || time_outs, "sf_gae_returns: value_bootstrap needs time_outs");
    // the reference multiplies the python doubles gamma*lambda before the cast to f32 (rl_utils.py:90)
    const float gl = (float)((double)gamma * (double)gae_lambda);
    k_gae_returns<<<dim3((unsigned)((E + 63) / 64)), dim3(GAE_NT), 0, STREAM(stream)>>>(
        rewards, dones, time_outs, values, valids, rms_stats, E, T, gamma, gl, value_bootstrap, advantages, returns);
    return sf_launch_status("sf_gae_returns");
}

// =========================================================================================== K12 running mean/std
__global__ __launch_bounds__(256) void k_moments(const float *__restrict__ x, const uint8_t *__restrict__ valids,
                                                 const int32_t *__restrict__ index, int64_t offset, int64_t n,
                                                 int dense_x, double *__restrict__ moments) {
    __shared__ double lds[4 * 3];
    double acc[3] = {0.0, 0.0, 0.0};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = index ? (int64_t)index[i] : offset + i;
        if (!valids || valids[j]) {
            const double v = (double)x[dense_x ? i : j];
            acc[0] += v;
            acc[1] += v * v;
            acc[2] += 1.0;
        }
    }
    sf_block_sum<3>(acc, lds);
    if (threadIdx.x == 0) {
        atomicAdd(&moments[0], acc[0]);
        atomicAdd(&moments[1], acc[1]);
        atomicAdd(&moments[2], acc[2]);
    }
}

extern "C" int sf_moments(const float *x, const uint8_t *valids, const int32_t *index, int64_t offset, int64_t n,
                          int dense_x, double *moments, void *stream) {
    SF_REQUIRE(x && moments && n >= 0 && offset >= 0, "sf_moments: bad args");
    int rc = sf_hip_status(hipMemsetAsync(moments, 0, 3 * sizeof(double), STREAM(stream)), "sf_moments memset");
    if (rc || n == 0) return rc;
    // latency-bound at minibatch sizes (32768 elements = 164 KB): one element per thread and one round trip, not an
    // 8-deep dependent loop on 16 CUs; large inputs grid-stride over 2048 blocks (8 per CU)
    const int64_t blocks = (n + 255) / 256;
    k_moments<<<dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, STREAM(stream)>>>(
        x, valids, index, offset, n, dense_x, moments);
    return sf_launch_status("sf_moments");
}

__global__ void k_rms_update(const double *__restrict__ stats_in, const double *__restrict__ moments,
                             double *__restrict__ stats_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double n = moments[2];
    const double mean = stats_in[0], var = stats_in[1], count = stats_in[2];
    if (n <= 0.0) { stats_out[0] = mean; stats_out[1] = var; stats_out[2] = count; return; }
    // batch mean / unbiased variance are fp32 tensors in the reference (x.mean(), x.var()) -> round to f32
    const double bm64 = moments[0] / n;
    const double bv64 = (moments[1] - moments[0] * bm64) / (n - 1.0);
    const double batch_mean = (double)(float)bm64, batch_var = (double)(float)bv64;
    const double delta = batch_mean - mean;
    const double tot = count + n;
    const double new_mean = mean + delta * n / tot;
    const double M2 = var * count + batch_var * n + (delta * delta) * count * n / tot;
    stats_out[0] = new_mean;
    stats_out[1] = M2 / tot;
    stats_out[2] = tot;
}

extern "C" int sf_rms_update(const double *stats_in, const double *moments, double *stats_out, void *stream) {
    SF_REQUIRE(stats_in && moments && stats_out, "sf_rms_update: null pointer");
    k_rms_update<<<dim3(1), dim3(64), 0, STREAM(stream)>>>(stats_in, moments, stats_out);
    return sf_launch_status("sf_rms_update");
}

__global__ __launch_bounds__(256) void k_rms_apply(float *__restrict__ x, int64_t n,
                                                   const double *__restrict__ stats, int denormalize) {
    const float mu = (float)stats[0];
    const float sigma = sqrtf((float)stats[1] + 1e-5f);
    const float inv = 1.0f / sigma;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float v = x[i];
        if (denormalize) v = clampf(v, -5.0f, 5.0f) * sigma + mu;
        else v = clampf((v - mu) * inv, -5.0f, 5.0f);
        x[i] = v;
    }
}

extern "C" int sf_rms_apply(float *x, int64_t n, const double *stats, int denormalize, void *stream) {
    SF_REQUIRE(x && stats && n >= 0, "sf_rms_apply: bad args");
    if (n == 0) return SF_OK;
    const int64_t blocks = (n + 255) / 256;
    k_rms_apply<<<dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, STREAM(stream)>>>(x, n, stats,
                                                                                                   denormalize);
    return sf_launch_status("sf_rms_apply");
}

// =========================================================================================== action distributions
// log-prob of the taken action under the current policy, for one sample (used by v-trace and the loss head).
template <int MAXA>
__device__ __forceinline__ float action_logp(const float *__restrict__ z, int A, int action_kind,
                                             const float *__restrict__ act_row) {
    if (action_kind == 0) {
        float zz[MAXA];
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < MAXA; ++k) { zz[k] = k < A ? z[k] : -INFINITY; mx = fmaxf(mx, zz[k]); }
        float se = 0.f;
#pragma unroll
        for (int k = 0; k < MAXA; ++k) if (k < A) se += expf(zz[k] - mx);
        const float lse = logf(se);
        const int a = (int)act_row[0];
        float r = 0.f;
#pragma unroll
        for (int k = 0; k < MAXA; ++k) if (k == a) r = (zz[k] - mx) - lse;
        return r;
    } else {
        const int D = A / 2;
        float r = 0.f;
        for (int k = 0; k < D; ++k) {
            const float mu = z[k], sd = clampf(expf(z[D + k]), 1e-4f, 1e4f);
            const float a = act_row[k];
            r += -((a - mu) * (a - mu)) / (2.f * (sd * sd)) - logf(sd) - 0.91893853320467274178f;
        }
        return r;
    }
}

// =========================================================================================== K17 V-trace
struct HeadsDev {
    int num_heads, head_n[8];
};
// action columns of a Tuple space: one per Discrete member (head_n > 0), D per Box(D) member (head_n = -D)
__device__ __host__ __forceinline__ int heads_action_cols(const int *head_n, int num_heads) {
    int c = 0;
    for (int h = 0; h < num_heads; ++h) c += head_n[h] > 0 ? 1 : -head_n[h];
    return c;
}
// log-prob of a Tuple action: sum over the members of log_softmax(z_h)[a_h] (Discrete) / Normal log-density (Box, head_n = -D)
__device__ __forceinline__ float tuple_logp(const float *__restrict__ z, const float *__restrict__ act, const HeadsDev &hd) {
    float lp = 0.f;
    int off = 0, aoff = 0;
    for (int h = 0; h < hd.num_heads; ++h) {
        const int nh = hd.head_n[h];
        if (nh < 0) {
            const int D = -nh;
            for (int k = 0; k < D; ++k) {
                const float mu = z[off + k], sd = clampf(expf(z[off + D + k]), 1e-4f, 1e4f);
                const float a = act[aoff + k];
                lp += -((a - mu) * (a - mu)) / (2.f * (sd * sd)) - logf(sd) - 0.91893853320467274178f;
            }
            off += 2 * D; aoff += D;
            continue;
        }
        float mx = -INFINITY;
        for (int k = 0; k < nh; ++k) mx = fmaxf(mx, z[off + k]);
        float se = 0.f;
        for (int k = 0; k < nh; ++k) se += expf(z[off + k] - mx);
        lp += (z[off + (int)act[aoff]] - mx) - logf(se);
        off += nh; aoff += 1;
    }
    return lp;
}

// phase 1 (parallel over samples): clamped importance ratio pi/pi_old of every step -> ratio_out[k]
template <int MAXA>
__global__ __launch_bounds__(256) void k_vtrace_ratio(const float *__restrict__ params, int ldp,
                                                      const float *__restrict__ actions,
                                                      const float *__restrict__ old_logp,
                                                      const int32_t *__restrict__ index, int64_t offset, int64_t n, int A,
                                                      int action_kind, float *__restrict__ ratio_out, HeadsDev hd) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int nact = hd.num_heads > 1 ? heads_action_cols(hd.head_n, hd.num_heads) : (action_kind == 0 ? 1 : A / 2);
    const int64_t d = index ? (int64_t)index[k] : offset + k;
    const float lp = hd.num_heads > 1 ? tuple_logp(params + k * ldp, actions + d * nact, hd)
                                      : action_logp<MAXA>(params + k * ldp, A, action_kind, actions + d * nact);
    ratio_out[k] = clampf(expf(lp - old_logp[d]), 0.05f, 20.0f);  // learner.py:591-594
}

// phase 2 (one lane per trajectory): the reverse recursion of learner.py:618-635; vs[k] holds the ratio of step k on
// entry (read before it is overwritten with the result)
template <int MAXA>
__global__ __launch_bounds__(64) void k_vtrace(const float *__restrict__ params, int ldp,
                                               const float *__restrict__ values, int ldv,
                                               const float *__restrict__ actions, const float *__restrict__ old_logp,
                                               const float *__restrict__ rewards, const uint8_t *__restrict__ dones,
                                               const int32_t *__restrict__ index, int64_t offset, int64_t ntraj, int A,
                                               int action_kind, int rec, float gamma, float rho_hat, float c_hat,
                                               float *__restrict__ vs, float *__restrict__ adv, HeadsDev hd) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ntraj) return;
    const int64_t base = j * rec;
    const int64_t last = base + rec - 1;
    const int64_t dl = index ? (int64_t)index[last] : offset + last;
    float next_values = (values[last * ldv] - rewards[dl]) / gamma;
    float next_vs = next_values;
    for (int i = rec - 1; i >= 0; --i) {
        const int64_t k = base + i;
        const int64_t d = index ? (int64_t)index[k] : offset + k;
        const float ratio = vs[k];  // written by k_vtrace_ratio
        const float rho = fminf(rho_hat, ratio), c = fminf(c_hat, ratio);
        const float not_done = 1.0f - (dones[d] ? 1.0f : 0.0f);
        const float ndg = not_done * gamma;
        const float cv = values[k * ldv], r = rewards[d];
        const float delta_s = rho * ((r + ndg * next_values) - cv);
        adv[k] = rho * ((r + ndg * next_vs) - cv);
        next_vs = (cv + delta_s) + (ndg * c) * (next_vs - next_values);
        vs[k] = next_vs;
        next_values = cv;
    }
}

extern "C" int sf_vtrace(const float *params, int ld_params, const float *values, int ld_values,
                         const float *actions, const float *old_logp, const float *rewards, const uint8_t *dones,
                         const int32_t *index, int64_t offset, int64_t n, int A, int action_kind, int recurrence,
                         float gamma, float rho_hat, float c_hat, float *vs, float *adv, const int32_t *head_n,
                         int num_heads, void *stream) {
    SF_REQUIRE(ld_params >= A && ld_values >= 1, "sf_vtrace: bad strides");
    HeadsDev hd = {};
    if (head_n && num_heads > 1) {
        SF_REQUIRE(num_heads <= 8 && action_kind == 0, "sf_vtrace: at most 8 heads (action_kind 0 with a head list)");
        int tot = 0;
        hd.num_heads = num_heads;
        for (int i = 0; i < num_heads; ++i) {  // head_n > 0: Discrete(n), n logits; < 0: Box(-n), 2 * (-n) parameters
            SF_REQUIRE(head_n[i] != 0, "sf_vtrace: empty action head");
            hd.head_n[i] = head_n[i];
            tot += head_n[i] > 0 ? head_n[i] : -2 * head_n[i];
        }
        SF_REQUIRE(tot == A, "sf_vtrace: head sizes sum to %d, A = %d", tot, A);
    }
    SF_REQUIRE(params && values && actions && old_logp && rewards && dones && vs && adv, "sf_vtrace: null pointer");
    SF_REQUIRE(recurrence > 0 && n % recurrence == 0, "sf_vtrace: n=%lld not a multiple of recurrence=%d",
               (long long)n, recurrence);
    SF_REQUIRE(A > 0 && A <= 128 && (action_kind == 0 || (action_kind == 1 && A % 2 == 0)),
               "sf_vtrace: unsupported action params A=%d kind=%d", A, action_kind);
    const int64_t ntraj = n / recurrence;
    if (ntraj == 0) return SF_OK;
    const dim3 grid((unsigned)((ntraj + 63) / 64)), block(64), rgrid((unsigned)((n + 255) / 256));
#define VT_LAUNCH(M)                                                                                             \
    k_vtrace_ratio<M><<<rgrid, dim3(256), 0, STREAM(stream)>>>(params, ld_params, actions, old_logp, index, offset, n, A,   \
                                                               action_kind, vs, hd);                              \
    k_vtrace<M><<<grid, block, 0, STREAM(stream)>>>(params, ld_params, values, ld_values, actions, old_logp, rewards, dones, index,    \
                                                    offset, ntraj, A, action_kind, recurrence, gamma, rho_hat,   \
                                                    c_hat, vs, adv, hd)
    if (A <= 8) { VT_LAUNCH(8); }
    else if (A <= 32) { VT_LAUNCH(32); }
    else { VT_LAUNCH(128); }
#undef VT_LAUNCH
    return sf_launch_status("sf_vtrace");
}

// =========================================================================================== K16 PPO loss fwd+bwd
// One lane per sample; the per-sample distribution lives in registers (template MAXA bounds the unrolled loops so no
// array is indexed dynamically).  Loss sums use wave shuffles -> LDS -> one double atomic per block per quantity.
struct LossDev {
    float clip_lo, clip_hi, clip_value, value_coeff, expl_coeff, kl_coeff;
    int expl_kind, action_kind, dense_adv;
    int num_heads, head_n[8];
    int ov_T;  // > 0: old_values is the slab's [E, ov_T + 1] array read in place (dataset row e*T+t -> e*(T+1)+t)
};
__device__ __forceinline__ int64_t ov_row(const LossDev &h, int64_t d) {
    if (h.ov_T <= 0) return d;
    const int64_t e = d / h.ov_T;
    return d + e;  // e*(T+1) + t = (e*T + t) + e
}

__device__ __forceinline__ void atomic_max_float(double *addr, float v) {
    // sums[4] holds the running max as a double; KL >= 0 up to rounding, compare on the bit pattern of (v+1) > 0
    unsigned long long *a = reinterpret_cast<unsigned long long *>(addr);
    const double dv = (double)v;
    unsigned long long old = *a, assumed;
    do {
        assumed = old;
        if (__longlong_as_double((long long)assumed) >= dv) break;
        old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(dv));
    } while (assumed != old);
}

template <int MAXA>
__global__ __launch_bounds__(256) void k_ppo_loss(const float *__restrict__ params, int ldp,
                                                  const float *__restrict__ values, int ldv,
                                                  const float *__restrict__ actions,
                                                  const float *__restrict__ old_logp,
                                                  const float *__restrict__ old_params,
                                                  const float *__restrict__ old_values, const float *__restrict__ adv,
                                                  const float *__restrict__ targets,
                                                  const uint8_t *__restrict__ valids,
                                                  const int32_t *__restrict__ index, int64_t offset, int64_t n, int A,
                                                  LossDev h, const double *__restrict__ moments,
                                                  double *__restrict__ sums, float *__restrict__ g_params,
                                                  float *__restrict__ g_values, float *__restrict__ ratio_out) {
    __shared__ double lds[4 * 4];
    __shared__ float lds_max[4];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // global (all-reduced) advantage moments -> mean / unbiased std, learner.py:646-647
    const double mn = moments[2];
    const double mean64 = moments[0] / mn;
    const double var64 = (moments[1] - moments[0] * mean64) / (mn - 1.0);
    const float adv_mean = (float)mean64;
    const float adv_std = (float)sqrt(var64 > 0.0 ? var64 : (mn > 1.0 ? 0.0 : NAN));
    const float denom = fmaxf(adv_std, 1e-7f);
    const float inv_n = 1.0f / (float)mn;
    // symmetric-KL exploration: clamp(max=30) / non-finite gate need the MEAN; it is passed through moments-like
    // slot sums[6] by a pre-pass only when that loss is selected (see sf_ppo_loss); gate defaults to open.
    const float symkl_gate = (h.expl_kind == 2) ? (float)sums[6] : 1.0f;

    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    float kl_max = -1.0f;
    if (i < n) {
        const int64_t d = index ? (int64_t)index[i] : offset + i;
        const bool valid = valids[d] != 0;
        const float *z = params + i * ldp;
        float *gz = g_params + i * ldp;
        float logp_a = 0.f, ent = 0.f, kl = 0.f, symkl = 0.f;
        const int D = A / 2;
        float zz[MAXA], zo[MAXA];
        float mx = -INFINITY, mxo = -INFINITY, lse = 0.f, lseo = 0.f;
        int act = 0;
        if (h.action_kind == 0) {
#pragma unroll
            for (int k = 0; k < MAXA; ++k) {
                zz[k] = k < A ? z[k] : -INFINITY;
                zo[k] = k < A ? old_params[d * A + k] : -INFINITY;
                mx = fmaxf(mx, zz[k]);
                mxo = fmaxf(mxo, zo[k]);
            }
            float se = 0.f, seo = 0.f;
#pragma unroll
            for (int k = 0; k < MAXA; ++k)
                if (k < A) { se += expf(zz[k] - mx); seo += expf(zo[k] - mxo); }
            lse = logf(se);
            lseo = logf(seo);
            act = (int)actions[d];
            const float u = 1.0f / (float)A, lu = logf(u);
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int k = 0; k < MAXA; ++k)
                if (k < A) {
                    const float lp = (zz[k] - mx) - lse, p = expf(lp), q = (zo[k] - mxo) - lseo;
                    if (k == act) logp_a = lp;
                    ent -= p * lp;
                    kl += p * (lp - q);
                    a1 += p * (lp - lu);
                    a2 += u * (lu - lp);
                }
            symkl = 0.5f * (a1 + a2);
        } else {
            for (int k = 0; k < D; ++k) {
                const float mu = z[k], sd = clampf(expf(z[D + k]), 1e-4f, 1e4f);
                const float a = actions[d * D + k];
                const float var = sd * sd;
                logp_a += -((a - mu) * (a - mu)) / (2.f * var) - logf(sd) - 0.91893853320467274178f;
                ent += 0.5f + 0.91893853320467274178f + logf(sd);
                const float muo = old_params[d * A + k], sdo = clampf(expf(old_params[d * A + D + k]), 1e-4f, 1e4f);
                const float vr = (sd / sdo) * (sd / sdo);
                const float t1 = ((mu - muo) / sdo) * ((mu - muo) / sdo);
                kl += 0.5f * (vr + t1 - 1.f - logf(vr));
            }
        }
        const float raw_ratio = expf(logp_a - old_logp[d]);
        const float ratio = clampf(raw_ratio, 0.05f, 20.0f);
        if (ratio_out) ratio_out[i] = ratio;  // summaries (learner.py:886-903)
        const int64_t da = h.dense_adv ? i : d;
        const float advn = (adv[da] - adv_mean) / denom;
        const float clipped = clampf(ratio, h.clip_lo, h.clip_hi);
        const float lu_ = ratio * advn, lc_ = clipped * advn;
        const float pl = fminf(lu_, lc_);
        const float v = values[i * ldv], vo = old_values[ov_row(h, d)], R = targets[da];
        const float vclip = vo + clampf(v - vo, -h.clip_value, h.clip_value);
        const float l1 = (v - R) * (v - R), l2 = (vclip - R) * (vclip - R);
        const float vl = fmaxf(l1, l2);
        if (valid) {
            acc[0] = pl;
            acc[1] = (h.expl_kind == 2) ? symkl : ent;
            acc[2] = kl;
            acc[3] = vl;
            kl_max = kl;
            // ---------------- backward (autograd semantics of min/max ties and clamp boundaries)
            float dpl_dr;
            const bool in_clip = ratio >= h.clip_lo && ratio <= h.clip_hi;
            if (lu_ < lc_) dpl_dr = advn;
            else if (lu_ > lc_) dpl_dr = in_clip ? advn : 0.f;
            else dpl_dr = 0.5f * advn + (in_clip ? 0.5f * advn : 0.f);
            const bool in_hard = raw_ratio >= 0.05f && raw_ratio <= 20.0f;
            const float dL_dlogp = in_hard ? (-inv_n) * dpl_dr * raw_ratio : 0.f;
            if (h.action_kind == 0) {
                const float u = 1.0f / (float)A, lu = logf(u);
                float klpu = 0.f;
                if (h.expl_kind == 2) {
#pragma unroll
                    for (int k = 0; k < MAXA; ++k)
                        if (k < A) { const float lp = (zz[k] - mx) - lse; klpu += expf(lp) * (lp - lu); }
                }
#pragma unroll
                for (int k = 0; k < MAXA; ++k)
                    if (k < A) {
                        const float lp = (zz[k] - mx) - lse, p = expf(lp), q = (zo[k] - mxo) - lseo;
                        float gk = dL_dlogp * ((k == act ? 1.f : 0.f) - p);
                        if (h.expl_kind == 1) gk += h.expl_coeff * inv_n * (p * (lp + ent));
                        if (h.expl_kind == 2)
                            gk += symkl_gate * h.expl_coeff * inv_n * 0.5f * (p * ((lp - lu) - klpu) + p - u);
                        if (h.kl_coeff != 0.f) gk += h.kl_coeff * inv_n * (p * ((lp - q) - kl));
                        gz[k] = gk;
                    }
            } else {
                for (int k = 0; k < D; ++k) {
                    const float mu = z[k], e = expf(z[D + k]);
                    const float sd = clampf(e, 1e-4f, 1e4f);
                    const float dsd = (e >= 1e-4f && e <= 1e4f) ? e : 0.f;
                    const float a = actions[d * D + k];
                    const float var = sd * sd;
                    float gmu = dL_dlogp * ((a - mu) / var);
                    float gsd = dL_dlogp * (((a - mu) * (a - mu)) / (var * sd) - 1.f / sd);
                    if (h.expl_kind == 1) gsd += -h.expl_coeff * inv_n * (1.f / sd);
                    if (h.kl_coeff != 0.f) {
                        const float muo = old_params[d * A + k];
                        const float sdo = clampf(expf(old_params[d * A + D + k]), 1e-4f, 1e4f);
                        gmu += h.kl_coeff * inv_n * ((mu - muo) / (sdo * sdo));
                        gsd += h.kl_coeff * inv_n * (sd / (sdo * sdo) - 1.f / sd);
                    }
                    gz[k] = gmu;
                    gz[D + k] = gsd * dsd;
                }
            }
            const bool in_v = (v - vo) >= -h.clip_value && (v - vo) <= h.clip_value;
            float dvl;
            if (l1 > l2) dvl = 2.f * (v - R);
            else if (l2 > l1) dvl = in_v ? 2.f * (vclip - R) : 0.f;
            else dvl = (v - R) + (in_v ? (vclip - R) : 0.f);
            g_values[i * ldv] = h.value_coeff * inv_n * dvl;
        } else {
            for (int k = 0; k < A; ++k) gz[k] = 0.f;
            g_values[i * ldv] = 0.f;
        }
    }
    // block reduction
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    kl_max = sf_wave_max(kl_max);
    if (lane == 0) lds_max[wave] = kl_max;
    sf_block_sum<4>(acc, lds);  // contains a __syncthreads()
    if (threadIdx.x == 0) {
        atomicAdd(&sums[0], acc[0]);
        atomicAdd(&sums[1], acc[1]);
        atomicAdd(&sums[2], acc[2]);
        atomicAdd(&sums[3], acc[3]);
        const float m = fmaxf(fmaxf(lds_max[0], lds_max[1]), fmaxf(lds_max[2], lds_max[3]));
        if (m > -1.0f) atomic_max_float(&sums[4], m);
    }
}


// Tuple of Discrete heads (TupleActionDistribution, action_distributions.py:197-287): independent categoricals whose
// log-prob / entropy / KL / symmetric-KL add up.  Same loss and backward as k_ppo_loss, per-head softmax statistics;
// heads are read from global memory in place (this kernel is ~0.1 % of a step: clarity over register blocking).
// sym_pass != 0: only accumulate the symmetric-KL sum (pre-pass for its mean gate) into sums[7].
__global__ __launch_bounds__(256) void k_ppo_loss_md(const float *__restrict__ params, int ldp,
                                                     const float *__restrict__ values, int ldv,
                                                     const float *__restrict__ actions,
                                                     const float *__restrict__ old_logp,
                                                     const float *__restrict__ old_params,
                                                     const float *__restrict__ old_values, const float *__restrict__ adv,
                                                     const float *__restrict__ targets,
                                                     const uint8_t *__restrict__ valids,
                                                     const int32_t *__restrict__ index, int64_t offset, int64_t n, int A,
                                                     LossDev h, const double *__restrict__ moments,
                                                     double *__restrict__ sums, float *__restrict__ g_params,
                                                     float *__restrict__ g_values, float *__restrict__ ratio_out,
                                                     int sym_pass) {
    __shared__ double lds[4 * 4];
    __shared__ float lds_max[4];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const double mn = moments[2];
    const double mean64 = moments[0] / mn;
    const double var64 = (moments[1] - moments[0] * mean64) / (mn - 1.0);
    const float adv_mean = (float)mean64;
    const float adv_std = (float)sqrt(var64 > 0.0 ? var64 : (mn > 1.0 ? 0.0 : NAN));
    const float denom = fmaxf(adv_std, 1e-7f);
    const float inv_n = 1.0f / (float)mn;
    const float symkl_gate = (h.expl_kind == 2 && !sym_pass) ? (float)sums[6] : 1.0f;
    const int H = h.num_heads;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    float kl_max = -1.0f;
    if (i < n) {
        const int64_t d = index ? (int64_t)index[i] : offset + i;
        const bool valid = valids[d] != 0;
        const float *z = params + i * ldp, *zo = old_params + d * A;
        float *gz = g_params + i * ldp;
        float logp_a = 0.f, ent = 0.f, kl = 0.f, symkl = 0.f;
        float mx[8], lse[8], mxo[8], lseo[8], ent_h[8], kl_h[8], klpu_h[8];
        int off = 0, aoff = 0;
        const int NA = heads_action_cols(h.head_n, H);
        for (int hd = 0; hd < H; ++hd) {
            const int nh = h.head_n[hd];
            if (nh < 0) {  // Box(D) member: [means | log_std], the formulas of k_ppo_loss's continuous branch
                const int Dh = -nh;
                float e = 0.f, kk = 0.f;
                for (int k = 0; k < Dh; ++k) {
                    const float mu = z[off + k], sd = clampf(expf(z[off + Dh + k]), 1e-4f, 1e4f);
                    const float a = actions[d * NA + aoff + k];
                    logp_a += -((a - mu) * (a - mu)) / (2.f * (sd * sd)) - logf(sd) - 0.91893853320467274178f;
                    e += 0.5f + 0.91893853320467274178f + logf(sd);
                    const float muo = zo[off + k], sdo = clampf(expf(zo[off + Dh + k]), 1e-4f, 1e4f);
                    const float vr = (sd / sdo) * (sd / sdo);
                    const float t1 = ((mu - muo) / sdo) * ((mu - muo) / sdo);
                    kk += 0.5f * (vr + t1 - 1.f - logf(vr));
                }
                ent_h[hd] = e; kl_h[hd] = kk; klpu_h[hd] = 0.f;
                ent += e; kl += kk;
                off += 2 * Dh; aoff += Dh;
                continue;
            }
            float m1 = -INFINITY, m2 = -INFINITY;
            for (int k = 0; k < nh; ++k) { m1 = fmaxf(m1, z[off + k]); m2 = fmaxf(m2, zo[off + k]); }
            float s1 = 0.f, s2 = 0.f;
            for (int k = 0; k < nh; ++k) { s1 += expf(z[off + k] - m1); s2 += expf(zo[off + k] - m2); }
            mx[hd] = m1; lse[hd] = logf(s1); mxo[hd] = m2; lseo[hd] = logf(s2);
            const int act = (int)actions[d * NA + aoff];
            const float u = 1.0f / (float)nh, lu = logf(u);
            float e = 0.f, kk = 0.f, a1 = 0.f, a2 = 0.f;
            for (int k = 0; k < nh; ++k) {
                const float lp = (z[off + k] - m1) - lse[hd], p = expf(lp), q = (zo[off + k] - m2) - lseo[hd];
                if (k == act) logp_a += lp;
                e -= p * lp;
                kk += p * (lp - q);
                a1 += p * (lp - lu);
                a2 += u * (lu - lp);
            }
            ent_h[hd] = e; kl_h[hd] = kk; klpu_h[hd] = a1;
            ent += e; kl += kk; symkl += 0.5f * (a1 + a2);
            off += nh; aoff += 1;
        }
        if (sym_pass) {
            if (valid) acc[0] = symkl;
        } else {
            const float raw_ratio = expf(logp_a - old_logp[d]);
            const float ratio = clampf(raw_ratio, 0.05f, 20.0f);
            if (ratio_out) ratio_out[i] = ratio;
            const int64_t da = h.dense_adv ? i : d;
            const float advn = (adv[da] - adv_mean) / denom;
            const float clipped = clampf(ratio, h.clip_lo, h.clip_hi);
            const float lu_ = ratio * advn, lc_ = clipped * advn;
            const float pl = fminf(lu_, lc_);
            const float v = values[i * ldv], vo = old_values[ov_row(h, d)], R = targets[da];
            const float vclip = vo + clampf(v - vo, -h.clip_value, h.clip_value);
            const float l1 = (v - R) * (v - R), l2 = (vclip - R) * (vclip - R);
            const float vl = fmaxf(l1, l2);
            if (valid) {
                acc[0] = pl;
                acc[1] = (h.expl_kind == 2) ? symkl : ent;
                acc[2] = kl;
                acc[3] = vl;
                kl_max = kl;
                float dpl_dr;
                const bool in_clip = ratio >= h.clip_lo && ratio <= h.clip_hi;
                if (lu_ < lc_) dpl_dr = advn;
                else if (lu_ > lc_) dpl_dr = in_clip ? advn : 0.f;
                else dpl_dr = 0.5f * advn + (in_clip ? 0.5f * advn : 0.f);
                const bool in_hard = raw_ratio >= 0.05f && raw_ratio <= 20.0f;
                const float dL_dlogp = in_hard ? (-inv_n) * dpl_dr * raw_ratio : 0.f;
                off = 0;
                aoff = 0;
                for (int hd = 0; hd < H; ++hd) {
                    const int nh = h.head_n[hd];
                    if (nh < 0) {  // Box(D) member
                        const int Dh = -nh;
                        for (int k = 0; k < Dh; ++k) {
                            const float mu = z[off + k], e = expf(z[off + Dh + k]);
                            const float sd = clampf(e, 1e-4f, 1e4f);
                            const float dsd = (e >= 1e-4f && e <= 1e4f) ? e : 0.f;
                            const float a = actions[d * NA + aoff + k];
                            const float var = sd * sd;
                            float gmu = dL_dlogp * ((a - mu) / var);
                            float gsd = dL_dlogp * (((a - mu) * (a - mu)) / (var * sd) - 1.f / sd);
                            if (h.expl_kind == 1) gsd += -h.expl_coeff * inv_n * (1.f / sd);
                            if (h.kl_coeff != 0.f) {
                                const float muo = zo[off + k], sdo = clampf(expf(zo[off + Dh + k]), 1e-4f, 1e4f);
                                gmu += h.kl_coeff * inv_n * ((mu - muo) / (sdo * sdo));
                                gsd += h.kl_coeff * inv_n * (sd / (sdo * sdo) - 1.f / sd);
                            }
                            gz[off + k] = gmu;
                            gz[off + Dh + k] = gsd * dsd;
                        }
                        off += 2 * Dh; aoff += Dh;
                        continue;
                    }
                    const int act = (int)actions[d * NA + aoff];
                    aoff += 1;
                    const float u = 1.0f / (float)nh, lu = logf(u);
                    for (int k = 0; k < nh; ++k) {
                        const float lp = (z[off + k] - mx[hd]) - lse[hd], p = expf(lp);
                        const float q = (zo[off + k] - mxo[hd]) - lseo[hd];
                        float gk = dL_dlogp * ((k == act ? 1.f : 0.f) - p);
                        if (h.expl_kind == 1) gk += h.expl_coeff * inv_n * (p * (lp + ent_h[hd]));
                        if (h.expl_kind == 2)
                            gk += symkl_gate * h.expl_coeff * inv_n * 0.5f * (p * ((lp - lu) - klpu_h[hd]) + p - u);
                        if (h.kl_coeff != 0.f) gk += h.kl_coeff * inv_n * (p * ((lp - q) - kl_h[hd]));
                        gz[off + k] = gk;
                    }
                    off += nh;
                }
                const bool in_v = (v - vo) >= -h.clip_value && (v - vo) <= h.clip_value;
                float dvl;
                if (l1 > l2) dvl = 2.f * (v - R);
                else if (l2 > l1) dvl = in_v ? 2.f * (vclip - R) : 0.f;
                else dvl = (v - R) + (in_v ? (vclip - R) : 0.f);
                g_values[i * ldv] = h.value_coeff * inv_n * dvl;
            } else {
                for (int k = 0; k < A; ++k) gz[k] = 0.f;
                g_values[i * ldv] = 0.f;
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    kl_max = sf_wave_max(kl_max);
    if (lane == 0) lds_max[wave] = kl_max;
    sf_block_sum<4>(acc, lds);
    if (threadIdx.x == 0) {
        if (sym_pass) { atomicAdd(&sums[7], acc[0]); return; }
        atomicAdd(&sums[0], acc[0]);
        atomicAdd(&sums[1], acc[1]);
        atomicAdd(&sums[2], acc[2]);
        atomicAdd(&sums[3], acc[3]);
        const float m = fmaxf(fmaxf(lds_max[0], lds_max[1]), fmaxf(lds_max[2], lds_max[3]));
        if (m > -1.0f) atomic_max_float(&sums[4], m);
    }
}

// pre-pass for the symmetric-KL exploration loss: mean over valid samples decides clamp(max=30) / isfinite gate
template <int MAXA>
__global__ __launch_bounds__(256) void k_symkl_sum(const float *__restrict__ params, int ldp,
                                                   const uint8_t *__restrict__ valids,
                                                   const int32_t *__restrict__ index, int64_t offset, int64_t n, int A,
                                                   double *__restrict__ out) {
    __shared__ double lds[4];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc[1] = {0.0};
    if (i < n) {
        const int64_t d = index ? (int64_t)index[i] : offset + i;
        if (valids[d]) {
            const float *z = params + i * ldp;
            float zz[MAXA];
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < MAXA; ++k) { zz[k] = k < A ? z[k] : -INFINITY; mx = fmaxf(mx, zz[k]); }
            float se = 0.f;
#pragma unroll
            for (int k = 0; k < MAXA; ++k) if (k < A) se += expf(zz[k] - mx);
            const float lse = logf(se), u = 1.0f / (float)A, lu = logf(u);
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int k = 0; k < MAXA; ++k)
                if (k < A) { const float lp = (zz[k] - mx) - lse; a1 += expf(lp) * (lp - lu); a2 += u * (lu - lp); }
            acc[0] = 0.5f * (a1 + a2);
        }
    }
    sf_block_sum<1>(acc, lds);
    if (threadIdx.x == 0) atomicAdd(out, acc[0]);
}

__global__ void k_symkl_gate(double *__restrict__ sums, const double *__restrict__ moments) {
    if (threadIdx.x || blockIdx.x) return;
    const float m = (float)(sums[7] / moments[2]);
    sums[6] = (isfinite(m) && m <= 30.0f) ? 1.0 : 0.0;
}

static LossDev make_loss_dev(const sf_loss_cfg *c) {
    LossDev h;
    h.clip_hi = (float)(1.0 + (double)c->clip_ratio);
    h.clip_lo = (float)(1.0 / (1.0 + (double)c->clip_ratio));
    h.clip_value = c->clip_value;
    h.value_coeff = c->value_loss_coeff;
    h.expl_coeff = c->exploration_coeff;
    h.kl_coeff = c->kl_coeff;
    h.expl_kind = c->exploration_coeff == 0.f ? 0 : c->exploration_kind;
    h.action_kind = c->action_kind;
    h.dense_adv = c->dense_adv;
    h.num_heads = c->num_heads > 1 ? c->num_heads : 1;
    for (int i = 0; i < 8; ++i) h.head_n[i] = c->num_heads > 1 ? c->head_n[i] : 0;
    h.ov_T = c->old_values_T > 0 ? c->old_values_T : 0;
    return h;
}

extern "C" int sf_ppo_loss(const float *params, int ld_params, const float *values, int ld_values,
                           const float *actions, const float *old_logp, const float *old_params,
                           const float *old_values, const float *adv, const float *targets, const uint8_t *valids,
                           const int32_t *index, int64_t offset, int64_t n, int A, const sf_loss_cfg *h_cfg,
                           const double *moments, double *sums, float *g_params, float *g_values, float *ratio_out,
                           void *stream) {
    SF_REQUIRE(ld_params >= A && ld_values >= 1, "sf_ppo_loss: bad strides");
    SF_REQUIRE(params && values && actions && old_logp && old_params && old_values && adv && targets && valids &&
                   h_cfg && moments && sums && g_params && g_values,
               "sf_ppo_loss: null pointer");
    SF_REQUIRE(n > 0 && A > 0 && A <= 128, "sf_ppo_loss: bad shape n=%lld A=%d", (long long)n, A);
    SF_REQUIRE(h_cfg->action_kind == 0 || (h_cfg->action_kind == 1 && A % 2 == 0), "sf_ppo_loss: bad action_kind");
    SF_REQUIRE(!(h_cfg->exploration_kind == 2 && h_cfg->action_kind != 0 && h_cfg->exploration_coeff != 0.f),
               "sf_ppo_loss: symmetric_kl exploration loss needs a categorical distribution");
    const LossDev h = make_loss_dev(h_cfg);
    int rc = sf_hip_status(hipMemsetAsync(sums, 0, 8 * sizeof(double), STREAM(stream)), "sf_ppo_loss memset");
    if (rc) return rc;
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
#define PL_DISPATCH(KERNEL, ...)                                                        \
    do {                                                                                \
        if (A <= 8) KERNEL<8><<<grid, block, 0, STREAM(stream)>>>(__VA_ARGS__);         \
        else if (A <= 32) KERNEL<32><<<grid, block, 0, STREAM(stream)>>>(__VA_ARGS__);  \
        else KERNEL<128><<<grid, block, 0, STREAM(stream)>>>(__VA_ARGS__);              \
    } while (0)
    if (h.action_kind == 0 && h.num_heads > 1) {
        int tot = 0;
        SF_REQUIRE(h.num_heads <= 8, "sf_ppo_loss: at most 8 action heads");
        bool any_box = false;  // head_n > 0: Discrete(n); head_n < 0: Box(-n) member with 2 * (-n) parameters
        for (int i = 0; i < h.num_heads; ++i) {
            SF_REQUIRE(h.head_n[i] != 0, "sf_ppo_loss: empty action head");
            tot += h.head_n[i] > 0 ? h.head_n[i] : -2 * h.head_n[i];
            any_box = any_box || h.head_n[i] < 0;
        }
        SF_REQUIRE(tot == A, "sf_ppo_loss: head sizes sum to %d, A = %d", tot, A);
        SF_REQUIRE(!(any_box && h.expl_kind == 2),
                   "sf_ppo_loss: symmetric_kl exploration loss needs categorical heads only (the reference's "
                   "ContinuousActionDistribution has no symmetric_kl_with_uniform_prior)");
        if (h.expl_kind == 2) {
            k_ppo_loss_md<<<grid, block, 0, STREAM(stream)>>>(params, ld_params, values, ld_values, actions, old_logp, old_params,
                                                              old_values, adv, targets, valids, index, offset, n, A, h, moments,
                                                              sums, g_params, g_values, nullptr, 1);
            k_symkl_gate<<<dim3(1), dim3(64), 0, STREAM(stream)>>>(sums, moments);
        }
        k_ppo_loss_md<<<grid, block, 0, STREAM(stream)>>>(params, ld_params, values, ld_values, actions, old_logp, old_params,
                                                          old_values, adv, targets, valids, index, offset, n, A, h, moments, sums,
                                                          g_params, g_values, ratio_out, 0);
        return sf_launch_status("sf_ppo_loss");
    }
    if (h.expl_kind == 2) {
        PL_DISPATCH(k_symkl_sum, params, ld_params, valids, index, offset, n, A, sums + 7);
        k_symkl_gate<<<dim3(1), dim3(64), 0, STREAM(stream)>>>(sums, moments);
    }
    PL_DISPATCH(k_ppo_loss, params, ld_params, values, ld_values, actions, old_logp, old_params, old_values, adv, targets, valids, index,
                offset, n, A, h, moments, sums, g_params, g_values, ratio_out);
#undef PL_DISPATCH
    return sf_launch_status("sf_ppo_loss");
}

__global__ void k_loss_scalars(const double *__restrict__ sums, const double *__restrict__ moments, LossDev h,
                               float *__restrict__ out) {
    if (threadIdx.x || blockIdx.x) return;
    const double n = moments[2];
    const double mean64 = moments[0] / n;
    const double var64 = (moments[1] - moments[0] * mean64) / (n - 1.0);
    out[0] = (float)(-(sums[0] / n));
    if (h.expl_kind == 1) out[1] = (float)(-(double)h.expl_coeff * (sums[1] / n));
    else if (h.expl_kind == 2) {
        float m = (float)(sums[1] / n);
        if (!isfinite(m)) m = 0.f;
        out[1] = h.expl_coeff * fminf(m, 30.0f);
    } else out[1] = 0.f;
    out[2] = (float)((double)h.kl_coeff * (sums[2] / n));
    out[3] = (float)((double)h.value_coeff * (sums[3] / n));
    out[4] = (float)(sums[2] / n);
    out[5] = (float)sums[4];
    out[6] = (float)mean64;
    out[7] = (float)sqrt(var64 > 0.0 ? var64 : 0.0);
    out[8] = (float)n;
    out[9] = (float)(sums[1] / n);
}

extern "C" int sf_loss_scalars(const double *sums, const double *moments, const sf_loss_cfg *h_cfg, float *out,
                               void *stream) {
    SF_REQUIRE(sums && moments && h_cfg && out, "sf_loss_scalars: null pointer");
    k_loss_scalars<<<dim3(1), dim3(64), 0, STREAM(stream)>>>(sums, moments, make_loss_dev(h_cfg), out);
    return sf_launch_status("sf_loss_scalars");
}

// =========================================================================================== train summaries
// learner.py:843-923 (_record_summaries) for the last minibatch of a training call: the reference evaluates ~25 small
// torch reductions with one .item() each; here ONE pass over the minibatch's rows accumulates every statistic
// (block reduction -> one atomic per block and quantity), a second small launch takes the maximum of Adam's second
// moments, and the caller reads 24 doubles back.  out layout (device double[24], initialised by the call):
//   [0] n rows  [1] n valid  [2] n same-policy  [3] sum value  [4] sum |1 - ratio| (valid)  [5] n clipped (valid)
//   [6] sum |v - v_old|  [7] sum version_diff (same policy)
//   [8] ratio min (valid)  [9] ratio max (valid)  [10] max |v - v_old|  [11] act min  [12] act max  [13] adv min
//   [14] adv max  [15] max |old action parameter|  [16] version_diff min  [17] version_diff max  [18] max exp_avg_sq
constexpr int SUMM_NS = 8, SUMM_NM = 11, SUMM_OUT = 24;
struct SummArgs {
    const uint8_t *valids;
    const float *ratio, *values, *old_values, *actions, *adv, *policy_version, *action_logits;
    const int32_t *policy_id, *index;
    int64_t offset, n;
    int ld_values, old_values_T, num_actions, A, dense_adv, my_pid;
    float train_step, clip_lo, clip_hi;
};
__device__ __forceinline__ void atomic_minmax_d(double *addr, double v, bool is_max) {
    unsigned long long *a = reinterpret_cast<unsigned long long *>(addr);
    unsigned long long old = *a, assumed;
    do {
        assumed = old;
        const double cur = __longlong_as_double((long long)assumed);
        if (is_max ? cur >= v : cur <= v) break;
        old = atomicCAS(a, assumed, (unsigned long long)__double_as_longlong(v));
    } while (assumed != old);
}
__global__ void k_summ_init(double *out) {
    const int i = threadIdx.x;
    if (i >= SUMM_OUT) return;
    // sums 0; minima +inf; maxima -inf
    const bool is_min = i == 8 || i == 11 || i == 13 || i == 16;
    const bool is_max = i == 9 || i == 10 || i == 12 || i == 14 || i == 15 || i == 17 || i == 18;
    out[i] = is_min ? (double)INFINITY : is_max ? -(double)INFINITY : 0.0;
}
__global__ __launch_bounds__(256) void k_train_summaries(SummArgs p, double *__restrict__ out) {
    __shared__ double lds[4 * SUMM_NS];
    __shared__ float ldm[4 * SUMM_NM];
    double s[SUMM_NS] = {0, 0, 0, 0, 0, 0, 0, 0};
    // minima: r_min, act_min, adv_min, vd_min;  maxima: r_max, dv_max, act_max, adv_max, logit_absmax, vd_max (+1 spare)
    float mn[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    float mx[7] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t d = p.index ? (int64_t)p.index[i] : p.offset + i;
        const bool valid = p.valids[d] != 0, same = p.policy_id[d] == p.my_pid;
        const float v = p.values[i * p.ld_values];
        const int64_t dv_row = p.old_values_T > 0 ? d + d / p.old_values_T : d;
        const float dv = fabsf(v - p.old_values[dv_row]);
        s[0] += 1.0;
        s[1] += valid ? 1.0 : 0.0;
        s[2] += same ? 1.0 : 0.0;
        s[3] += (double)v;
        s[6] += (double)dv;
        mx[1] = fmaxf(mx[1], dv);
        if (valid) {
            const float r = p.ratio[i];
            s[4] += (double)fabsf(1.0f - r);
            s[5] += (r < p.clip_lo ? 1.0 : 0.0) + (r > p.clip_hi ? 1.0 : 0.0);
            mn[0] = fminf(mn[0], r);
            mx[0] = fmaxf(mx[0], r);
        }
        for (int a = 0; a < p.num_actions; ++a) {
            const float x = p.actions[d * p.num_actions + a];
            mn[1] = fminf(mn[1], x);
            mx[2] = fmaxf(mx[2], x);
        }
        const float ad = p.adv[p.dense_adv ? i : d];
        mn[2] = fminf(mn[2], ad);
        mx[3] = fmaxf(mx[3], ad);
        for (int a = 0; a < p.A; ++a) mx[4] = fmaxf(mx[4], fabsf(p.action_logits[d * p.A + a]));
        if (same) {
            const float vd = p.train_step - p.policy_version[d];
            s[7] += (double)vd;
            mn[3] = fminf(mn[3], vd);
            mx[5] = fmaxf(mx[5], vd);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < SUMM_NS; ++k) s[k] = sf_wave_sum(s[k]);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mn[k] = fminf(mn[k], __shfl_down(mn[k], off, 64));
#pragma unroll
    for (int k = 0; k < 7; ++k) mx[k] = sf_wave_max(mx[k]);
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < SUMM_NS; ++k) lds[wave * SUMM_NS + k] = s[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) ldm[wave * SUMM_NM + k] = mn[k];
#pragma unroll
        for (int k = 0; k < 7; ++k) ldm[wave * SUMM_NM + 4 + k] = mx[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + 63) >> 6;
        for (int k = 0; k < SUMM_NS; ++k) {
            double t = 0.0;
            for (int w = 0; w < nw; ++w) t += lds[w * SUMM_NS + k];
            if (t != 0.0) atomicAdd(&out[k], t);
        }
        float m4[4] = {INFINITY, INFINITY, INFINITY, INFINITY}, x7[7];
        for (int k = 0; k < 7; ++k) x7[k] = -INFINITY;
        for (int w = 0; w < nw; ++w) {
            for (int k = 0; k < 4; ++k) m4[k] = fminf(m4[k], ldm[w * SUMM_NM + k]);
            for (int k = 0; k < 7; ++k) x7[k] = fmaxf(x7[k], ldm[w * SUMM_NM + 4 + k]);
        }
        // out slots: 8 r_min, 9 r_max, 10 dv_max, 11 act_min, 12 act_max, 13 adv_min, 14 adv_max, 15 logit, 16 vd_min, 17 vd_max
        atomic_minmax_d(&out[8], (double)m4[0], false);
        atomic_minmax_d(&out[9], (double)x7[0], true);
        atomic_minmax_d(&out[10], (double)x7[1], true);
        atomic_minmax_d(&out[11], (double)m4[1], false);
        atomic_minmax_d(&out[12], (double)x7[2], true);
        atomic_minmax_d(&out[13], (double)m4[2], false);
        atomic_minmax_d(&out[14], (double)x7[3], true);
        atomic_minmax_d(&out[15], (double)x7[4], true);
        atomic_minmax_d(&out[16], (double)m4[3], false);
        atomic_minmax_d(&out[17], (double)x7[5], true);
    }
}
__global__ __launch_bounds__(256) void k_max_f32(const float *__restrict__ x, int64_t n, double *__restrict__ out) {
    __shared__ float ldm[4];
    float m = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        m = fmaxf(m, x[i]);
    m = sf_wave_max(m);
    if ((threadIdx.x & 63) == 0) ldm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)((blockDim.x + 63) >> 6); ++w) m = fmaxf(m, ldm[w]);
        atomic_minmax_d(out, (double)m, true);
    }
}

extern "C" int sf_train_summaries(const uint8_t *valids, const float *ratio, const float *values, int ld_values,
                                  const float *old_values, int old_values_T, const float *actions, int num_actions,
                                  const float *adv, int dense_adv, const int32_t *policy_id, const float *policy_version,
                                  const float *action_logits, int A, const int32_t *index, int64_t offset, int64_t n,
                                  int my_policy_id, int train_step, float clip_ratio, const float *exp_avg_sq, int64_t P,
                                  double *out, void *stream) {
    SF_REQUIRE(valids && ratio && values && old_values && actions && adv && policy_id && policy_version && action_logits &&
                   out && n > 0 && num_actions > 0 && A > 0 && ld_values >= 1,
               "sf_train_summaries: bad args");
    SummArgs p{valids, ratio, values, old_values, actions, adv, policy_version, action_logits, policy_id, index, offset, n,
               ld_values, old_values_T, num_actions, A, dense_adv, my_policy_id, (float)train_step,
               (float)(1.0 / (1.0 + (double)clip_ratio)), (float)(1.0 + (double)clip_ratio)};
    k_summ_init<<<dim3(1), dim3(64), 0, STREAM(stream)>>>(out);
    const int64_t blocks = (n + 255) / 256;
    k_train_summaries<<<dim3((unsigned)(blocks < 1024 ? blocks : 1024)), dim3(256), 0, STREAM(stream)>>>(p, out);
    if (exp_avg_sq && P > 0) {
        const int64_t b2 = (P + 1023) / 1024;
        k_max_f32<<<dim3((unsigned)(b2 < 1024 ? b2 : 1024)), dim3(256), 0, STREAM(stream)>>>(exp_avg_sq, P, out + 18);
    }
    return sf_launch_status("sf_train_summaries");
}

// =========================================================================================== device row copies
// dst[r][0..row_bytes) = src[r][0..row_bytes) for r < rows, rows `pitch` bytes apart on either side: the slab's column
// copies (next rollout's obs[:, 0] <- obs[:, T], rnn_states likewise, bootstrap value -> values[:, T]).  16-byte units
// when every address allows it, 4-byte units otherwise, bytes as the last resort.
template <typename V>
__global__ __launch_bounds__(256) void k_copy_rows(char *__restrict__ dst, int64_t dst_pitch, const char *__restrict__ src,
                                                   int64_t src_pitch, int64_t units_per_row, int64_t rows) {
    const int64_t total = units_per_row * rows;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / units_per_row, u = i - r * units_per_row;
        *reinterpret_cast<V *>(dst + r * dst_pitch + u * (int64_t)sizeof(V)) =
            *reinterpret_cast<const V *>(src + r * src_pitch + u * (int64_t)sizeof(V));
    }
}
extern "C" int sf_copy_rows(void *dst, int64_t dst_pitch, const void *src, int64_t src_pitch, int64_t row_bytes,
                            int64_t rows, void *stream) {
    SF_REQUIRE(dst && src && row_bytes > 0 && rows > 0 && (rows == 1 || (llabs(dst_pitch) >= row_bytes && llabs(src_pitch) >= row_bytes)),
               "sf_copy_rows: bad args (row_bytes=%lld rows=%lld)", (long long)row_bytes, (long long)rows);
    const uintptr_t all = (uintptr_t)dst | (uintptr_t)src | (uintptr_t)dst_pitch | (uintptr_t)src_pitch | (uintptr_t)row_bytes;
    const int unit = (all & 15) == 0 ? 16 : (all & 3) == 0 ? 4 : 1;
    const int64_t upr = row_bytes / unit, total = upr * rows;
    const dim3 grid((unsigned)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536)), block(256);
    char *d = reinterpret_cast<char *>(dst);
    const char *s = reinterpret_cast<const char *>(src);
    if (unit == 16) k_copy_rows<uint4><<<grid, block, 0, STREAM(stream)>>>(d, dst_pitch, s, src_pitch, upr, rows);
    else if (unit == 4) k_copy_rows<uint32_t><<<grid, block, 0, STREAM(stream)>>>(d, dst_pitch, s, src_pitch, upr, rows);
    else k_copy_rows<uint8_t><<<grid, block, 0, STREAM(stream)>>>(d, dst_pitch, s, src_pitch, upr, rows);
    return sf_launch_status("sf_copy_rows");
}

// =========================================================================================== host-env ingest
extern "C" int sf_h2d_rows(void *dst, int64_t dst_pitch, const void *src, int64_t src_pitch, int64_t row_bytes,
                           int64_t rows, void *stream) {
    SF_REQUIRE(dst && src && row_bytes > 0 && rows > 0 && dst_pitch >= row_bytes && src_pitch >= row_bytes,
               "sf_h2d_rows: row_bytes=%lld rows=%lld pitches %lld/%lld", (long long)row_bytes, (long long)rows,
               (long long)dst_pitch, (long long)src_pitch);
    const hipError_t e = hipMemcpy2DAsync(dst, (size_t)dst_pitch, src, (size_t)src_pitch, (size_t)row_bytes,
                                          (size_t)rows, hipMemcpyHostToDevice, STREAM(stream));
    SF_REQUIRE(e == hipSuccess, "sf_h2d_rows: hipMemcpy2DAsync: %s", hipGetErrorString(e));
    return 0;
}

// =========================================================================================== K14 minibatch indices
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}
// bijection on [0, 2^(2*half)) : 4-round balanced Feistel
__device__ __forceinline__ uint32_t feistel(uint32_t x, int half, uint32_t key) {
    const uint32_t mask = (1u << half) - 1u;
    uint32_t l = x >> half, r = x & mask;
#pragma unroll
    for (int round = 0; round < 4; ++round) {
        const uint32_t f = mix32(r ^ (key + 0x9E3779B9u * (uint32_t)(round + 1))) & mask;
        const uint32_t nl = r, nr = l ^ f;
        l = nl; r = nr;
    }
    return (l << half) | r;
}

// chunk_starts != NULL: the permutation comes from the host (the reference's seeded np.random.permutation of the
// recurrence-aligned chunk starts, learner.py:507-510), given as dataset indices; only the expansion runs here.
__global__ __launch_bounds__(256) void k_minibatch_indices(int32_t *__restrict__ out, int64_t n_chunks, int rec,
                                                           int shuffle, int half, uint32_t key,
                                                           const int32_t *__restrict__ chunk_starts) {
    const int64_t wave_first = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~(int64_t)63;  // first chunk of wave
    const int lane = threadIdx.x & 63;
    const int64_t j = wave_first + lane;
    uint32_t start = 0;
    if (j < n_chunks) {
        uint32_t x = (uint32_t)j;
        if (chunk_starts) {
            x = (uint32_t)chunk_starts[j] / (uint32_t)rec;
        } else if (shuffle) {
            do { x = feistel(x, half, key); } while ((int64_t)x >= n_chunks);  // cycle walking keeps it a bijection
        }
        start = x;
    }
    // expansion: the wave's 64 chunk starts are exchanged with wave shuffles so that the 64*rec output words are
    // written as fully coalesced runs (lane l writes word q*64+l of the wave's output span).
    const int64_t out_base = wave_first * rec;
    for (int q = 0; q < rec; ++q) {
        const int w = q * 64 + lane;
        const int src_lane = w / rec, r = w - src_lane * rec;
        const uint32_t s = __shfl(start, src_lane, 64);
        if (wave_first + src_lane < n_chunks) out[out_base + w] = (int32_t)(s * (uint32_t)rec + (uint32_t)r);
    }
}

extern "C" int sf_minibatch_indices(int32_t *out, int64_t experience_size, int recurrence, int shuffle, uint32_t seed,
                                    uint32_t epoch, void *stream) {
    SF_REQUIRE(out && experience_size > 0 && recurrence > 0 && experience_size % recurrence == 0,
               "sf_minibatch_indices: experience_size=%lld recurrence=%d", (long long)experience_size, recurrence);
    SF_REQUIRE(experience_size < (1LL << 31), "sf_minibatch_indices: experience too large for int32 indices");
    const int64_t n_chunks = experience_size / recurrence;
    int bits = 1;
    while ((1LL << bits) < n_chunks) ++bits;
    const int half = (bits + 1) / 2;
    const uint32_t key = seed * 0x9E3779B1u + epoch * 0x85EBCA77u + 0x165667B1u;
    k_minibatch_indices<<<dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, STREAM(stream)>>>(
        out, n_chunks, recurrence, shuffle, half, key, nullptr);
    return sf_launch_status("sf_minibatch_indices");
}

extern "C" int sf_minibatch_expand(const int32_t *chunk_starts, int32_t *out, int64_t experience_size, int recurrence,
                                   void *stream) {
    SF_REQUIRE(chunk_starts && out && experience_size > 0 && recurrence > 0 && experience_size % recurrence == 0,
               "sf_minibatch_expand: experience_size=%lld recurrence=%d", (long long)experience_size, recurrence);
    SF_REQUIRE(experience_size < (1LL << 31), "sf_minibatch_expand: experience too large for int32 indices");
    const int64_t n_chunks = experience_size / recurrence;
    k_minibatch_indices<<<dim3((unsigned)((n_chunks + 255) / 256)), dim3(256), 0, STREAM(stream)>>>(
        out, n_chunks, recurrence, 0, 0, 0u, chunk_starts);
    return sf_launch_status("sf_minibatch_expand");
}

// =========================================================================================== K18/K19 clip + Adam
__global__ __launch_bounds__(256) void k_grad_sumsq(const float *__restrict__ g, int64_t P,
                                                    double *__restrict__ sumsq) {
    __shared__ double lds[4];
    double acc[1] = {0.0};
    const int64_t nvec = P / 4;
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = tid; i < nvec; i += nt) {
        const float4 v = g4[i];
        acc[0] += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
    }
    for (int64_t i = nvec * 4 + tid; i < P; i += nt) acc[0] += (double)g[i] * g[i];
    sf_block_sum<1>(acc, lds);
    if (threadIdx.x == 0) atomicAdd(sumsq, acc[0]);
}

extern "C" int sf_grad_sumsq(const float *g, int64_t P, double *sumsq, void *stream) {
    SF_REQUIRE(g && sumsq && P > 0, "sf_grad_sumsq: bad args");
    SF_REQUIRE(((uintptr_t)g & 15) == 0, "sf_grad_sumsq: gradient buffer must be 16-byte aligned");
    int rc = sf_hip_status(hipMemsetAsync(sumsq, 0, sizeof(double), STREAM(stream)), "sf_grad_sumsq memset");
    if (rc) return rc;
    const int64_t blocks = (P / 4 + 255) / 256;
    k_grad_sumsq<<<dim3((unsigned)(blocks < 1024 ? (blocks ? blocks : 1) : 1024)), dim3(256), 0, STREAM(stream)>>>(
        g, P, sumsq);
    return sf_launch_status("sf_grad_sumsq");
}

struct AdamC {
    float lr_step, bc2_sqrt, w1, b2, w2, eps, max_norm, grad_scale;
};

__device__ __forceinline__ void adam1(float &p, float g, float &m, float &v, const AdamC &c, float coef) {
    g = g * coef;
    m = m + (g - m) * c.w1;
    v = v * c.b2 + (g * g) * c.w2;
    const float denom = sqrtf(v) / c.bc2_sqrt + c.eps;
    p = p - c.lr_step * (m / denom);
}

__global__ __launch_bounds__(256) void k_adam(float *__restrict__ p, const float *__restrict__ g,
                                              float *__restrict__ m, float *__restrict__ v, int64_t P, AdamC c,
                                              const double *__restrict__ sumsq, const uint32_t *__restrict__ skip,
                                              const float *__restrict__ lr_dev) {
    if (skip && *skip) return;  // an aborted fused recurrent pass produced this gradient: leave weights and moments alone
    if (lr_dev) c.lr_step *= *lr_dev;  // sf_adam_step_dlr: the learning rate lives on the device (KL-adaptive schedule)
    float coef = c.grad_scale;
    if (sumsq && c.max_norm > 0.f) {
        // clip_grad_norm_: total_norm of the (already grad_scale'd) gradient
        const float total = (float)sqrt(*sumsq) * fabsf(c.grad_scale);
        const float cc = c.max_norm / (total + 1e-6f);
        coef = c.grad_scale * fminf(cc, 1.0f);
    }
    const int64_t nvec = P / 4;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
    float4 *p4 = reinterpret_cast<float4 *>(p), *m4 = reinterpret_cast<float4 *>(m), *v4 = reinterpret_cast<float4 *>(v);
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    for (int64_t i = tid; i < nvec; i += nt) {
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const float4 gg = g4[i];
        adam1(pp.x, gg.x, mm.x, vv.x, c, coef);
        adam1(pp.y, gg.y, mm.y, vv.y, c, coef);
        adam1(pp.z, gg.z, mm.z, vv.z, c, coef);
        adam1(pp.w, gg.w, mm.w, vv.w, c, coef);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    for (int64_t i = nvec * 4 + tid; i < P; i += nt) adam1(p[i], g[i], m[i], v[i], c, coef);
}

static int adam_step_impl(float *p, const float *g, float *m, float *v, int64_t P, int step, float lr, float beta1,
                          float beta2, float eps, float max_grad_norm, const double *sumsq, float grad_scale,
                          const uint32_t *skip_flag, const float *lr_dev, void *stream, const char *who) {
    SF_REQUIRE(p && g && m && v && P > 0 && step >= 1, "%s: bad args", who);
    SF_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
               "%s: buffers must be 16-byte aligned", who);
    AdamC c;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    c.lr_step = (float)((double)lr / bc1);
    c.bc2_sqrt = (float)sqrt(bc2);
    c.w1 = (float)(1.0 - (double)beta1);
    c.b2 = beta2;
    c.w2 = (float)(1.0 - (double)beta2);
    c.eps = eps;
    c.max_norm = max_grad_norm;
    c.grad_scale = grad_scale;
    const int64_t blocks = (P / 4 + 255) / 256;
    k_adam<<<dim3((unsigned)(blocks < 2048 ? (blocks ? blocks : 1) : 2048)), dim3(256), 0, STREAM(stream)>>>(
        p, g, m, v, P, c, sumsq, skip_flag, lr_dev);
    return sf_launch_status(who);
}

extern "C" int sf_adam_step(float *p, const float *g, float *m, float *v, int64_t P, int step, float lr, float beta1,
                            float beta2, float eps, float max_grad_norm, const double *sumsq, float grad_scale,
                            const uint32_t *skip_flag, void *stream) {
    return adam_step_impl(p, g, m, v, P, step, lr, beta1, beta2, eps, max_grad_norm, sumsq, grad_scale, skip_flag, nullptr,
                          stream, "sf_adam_step");
}
// the same step with the learning rate read from DEVICE memory: lr = *lr_dev * lr_scale (lr_scale: the valid-sample
// fraction of learner.py:788-794).  With sf_lr_kl_adaptive this keeps the per-minibatch KL-adaptive schedule
// (learner.py:46-85, lr_schedule=kl_adaptive_minibatch) off the host: no read-back between SGD steps.
extern "C" int sf_adam_step_dlr(float *p, const float *g, float *m, float *v, int64_t P, int step, const float *lr_dev,
                                float lr_scale, float beta1, float beta2, float eps, float max_grad_norm,
                                const double *sumsq, float grad_scale, const uint32_t *skip_flag, void *stream) {
    SF_REQUIRE(lr_dev, "sf_adam_step_dlr: null lr_dev");
    return adam_step_impl(p, g, m, v, P, step, lr_scale, beta1, beta2, eps, max_grad_norm, sumsq, grad_scale, skip_flag,
                          lr_dev, stream, "sf_adam_step_dlr");
}

// Measurement helper (bench.py's roofline.clock_ghz): ONE wave spins for `spin_cycles` shader cycles and reports how many
// ticks of the constant 100 MHz wall clock (s_memrealtime) went by meanwhile: shader clock [GHz] = 0.1 * out[0] / out[1].
// Launched on a side stream during the timed region it reads the clock the chip actually runs at under that load (the
// DVFS-governed clock is what the 2.4 GHz peak has to be scaled by; sysfs / smi values proved unreliable on the boxes).
__global__ void k_clock_probe(unsigned long long *__restrict__ out, int spin_cycles) {
    if (threadIdx.x || blockIdx.x) return;
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    unsigned long long c1 = c0;
    while ((long long)(c1 - c0) < (long long)spin_cycles) c1 = clock64();
    const unsigned long long w1 = wall_clock64();
    out[0] = c1 - c0;
    out[1] = w1 - w0;
}
extern "C" int sf_clock_probe(unsigned long long *out, int spin_cycles, void *stream) {
    SF_REQUIRE(out && spin_cycles > 0, "sf_clock_probe: bad args");
    k_clock_probe<<<dim3(1), dim3(64), 0, STREAM(stream)>>>(out, spin_cycles);
    return sf_launch_status("sf_clock_probe");
}

// learner.py:46-85 (KlAdaptiveScheduler.update): lr /= 1.5 (floor lr_min) when the KL of the step exceeds 2 x threshold,
// lr *= 1.5 (cap lr_max) when it is below threshold / 2.  kl: device float (the minibatch's mean KL, sf_loss_scalars[4]);
// lr_dev: device float, updated in place; lr_out (may be NULL): a copy of the new value for the epoch's read-back.
__global__ void k_lr_kl_adaptive(const float *__restrict__ kl, float *__restrict__ lr_dev, float threshold, float lr_min,
                                 float lr_max, float *__restrict__ lr_out) {
    if (threadIdx.x || blockIdx.x) return;
    const double k = (double)*kl;
    double lr = (double)*lr_dev;
    if (k > 2.0 * (double)threshold) lr = fmax(lr / 1.5, (double)lr_min);
    if (k < 0.5 * (double)threshold) lr = fmin(lr * 1.5, (double)lr_max);
    *lr_dev = (float)lr;
    if (lr_out) *lr_out = (float)lr;
}
extern "C" int sf_lr_kl_adaptive(const float *kl, float *lr_dev, float threshold, float lr_min, float lr_max,
                                 float *lr_out, void *stream) {
    SF_REQUIRE(kl && lr_dev, "sf_lr_kl_adaptive: null pointer");
    k_lr_kl_adaptive<<<dim3(1), dim3(64), 0, STREAM(stream)>>>(kl, lr_dev, threshold, lr_min, lr_max, lr_out);
    return sf_launch_status("sf_lr_kl_adaptive");
}

// =========================================================================================== Lamb (cfg.optimizer=lamb)
// sample_factory/algo/utils/optimizers.py:14-189 as the Learner configures it (learner.py:228-243: lr, betas, eps;
// weight_decay 1e-4, min_trust 0.01, bias correction, no look-ahead), list-params path: Adam direction + weight decay,
// then a per-TENSOR trust ratio min(||w||, 10) / ||step|| clamped to [min_trust, 1/min_trust].  The flat buffer carries
// a segment id per element (one id per reference tensor; 255 = padding), so two streaming passes suffice:
//   pass 1: moments, direction u -> scratch, per-segment sum w^2 / sum u^2 (LDS bins -> one atomic per bin and block)
//   pass 2: w -= lr * trust[seg] * u
struct LambC {
    float inv_bc1, inv_bc2_sqrt, w1, b1, b2, w2, eps, wd, max_norm, grad_scale, lr, min_trust;
};
constexpr int LAMB_MAX_SEG = 64;

__global__ __launch_bounds__(256) void k_lamb_dir(const float *__restrict__ p, const float *__restrict__ g,
                                                  float *__restrict__ m, float *__restrict__ v,
                                                  float *__restrict__ upd, const uint8_t *__restrict__ seg, int64_t P,
                                                  LambC c, const double *__restrict__ sumsq,
                                                  double *__restrict__ seg_sums, const uint32_t *__restrict__ skip) {
    if (skip && *skip) return;
    __shared__ double bins[2 * LAMB_MAX_SEG];
    for (int i = threadIdx.x; i < 2 * LAMB_MAX_SEG; i += blockDim.x) bins[i] = 0.0;
    __syncthreads();
    float coef = c.grad_scale;
    if (sumsq && c.max_norm > 0.f) {
        const float total = (float)sqrt(*sumsq) * fabsf(c.grad_scale);
        coef = c.grad_scale * fminf(c.max_norm / (total + 1e-6f), 1.0f);
    }
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = tid; i < P; i += nt) {
        const int sg = seg[i];
        if (sg >= LAMB_MAX_SEG) continue;  // padding
        const float gg = g[i] * coef, w = p[i];
        const float mm = m[i] * c.b1 + c.w1 * gg;          // exp_avg.mul_(b1).add_(grad, alpha=1-b1)
        const float vv = v[i] * c.b2 + c.w2 * (gg * gg);   // exp_avg_sq.mul_(b2).addcmul_(grad, grad, value=1-b2)
        m[i] = mm; v[i] = vv;
        const float mh = mm * c.inv_bc1, vh = sqrtf(vv) * c.inv_bc2_sqrt;
        float u = mh / (vh + c.eps);
        if (c.wd > 0.f) u = u + c.wd * w;
        upd[i] = u;
        atomicAdd(&bins[2 * sg], (double)w * (double)w);
        atomicAdd(&bins[2 * sg + 1], (double)u * (double)u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * LAMB_MAX_SEG; i += blockDim.x)
        if (bins[i] != 0.0) atomicAdd(&seg_sums[i], bins[i]);
}

__global__ __launch_bounds__(256) void k_lamb_apply(float *__restrict__ p, const float *__restrict__ upd,
                                                    const uint8_t *__restrict__ seg, int64_t P, LambC c,
                                                    const double *__restrict__ seg_sums,
                                                    const uint32_t *__restrict__ skip) {
    if (skip && *skip) return;
    __shared__ float trust[LAMB_MAX_SEG];
    for (int i = threadIdx.x; i < LAMB_MAX_SEG; i += blockDim.x) {
        const float wn = (float)sqrt(seg_sums[2 * i]), sn = (float)sqrt(seg_sums[2 * i + 1]);
        float tr = 1.0f;
        if (c.min_trust != 1.0f && wn != 0.f && sn != 0.f) {
            tr = fminf(wn, 10.0f) / sn;
            tr = fminf(fmaxf(tr, c.min_trust), 1.0f / c.min_trust);
        }
        trust[i] = tr;
    }
    __syncthreads();
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = tid; i < P; i += nt) {
        const int sg = seg[i];
        if (sg < LAMB_MAX_SEG) p[i] = p[i] + (-c.lr * trust[sg]) * upd[i];  // p.add_(adam_step, alpha=-lr*trust)
    }
}

extern "C" int sf_lamb_step(float *p, const float *g, float *m, float *v, float *scratch, const uint8_t *seg_id,
                            double *seg_sums, int64_t P, int num_segments, int step, float lr, float beta1, float beta2,
                            float eps, float weight_decay, float min_trust, float max_grad_norm, const double *sumsq,
                            float grad_scale, const uint32_t *skip_flag, void *stream) {
    SF_REQUIRE(p && g && m && v && scratch && seg_id && seg_sums && P > 0 && step >= 1, "sf_lamb_step: bad args");
    SF_REQUIRE(num_segments >= 1 && num_segments <= LAMB_MAX_SEG, "sf_lamb_step: 1..%d segments", LAMB_MAX_SEG);
    LambC c;
    c.inv_bc1 = (float)(1.0 / (1.0 - pow((double)beta1, (double)step)));
    c.inv_bc2_sqrt = (float)(1.0 / sqrt(1.0 - pow((double)beta2, (double)step)));
    c.b1 = beta1; c.w1 = (float)(1.0 - (double)beta1);
    c.b2 = beta2; c.w2 = (float)(1.0 - (double)beta2);
    c.eps = eps; c.wd = weight_decay; c.max_norm = max_grad_norm; c.grad_scale = grad_scale; c.lr = lr;
    c.min_trust = min_trust;
    int rc = sf_hip_status(hipMemsetAsync(seg_sums, 0, 2 * LAMB_MAX_SEG * sizeof(double), STREAM(stream)), "sf_lamb_step");
    if (rc) return rc;
    const int64_t blocks = (P + 255) / 256;
    const dim3 grid((unsigned)(blocks < 1024 ? blocks : 1024));
    k_lamb_dir<<<grid, dim3(256), 0, STREAM(stream)>>>(p, g, m, v, scratch, seg_id, P, c, sumsq, seg_sums, skip_flag);
    k_lamb_apply<<<grid, dim3(256), 0, STREAM(stream)>>>(p, scratch, seg_id, P, c, seg_sums, skip_flag);
    return sf_launch_status("sf_lamb_step");
}

// =========================================================================================== K4/K5 sample + write
template <int MAXA>
__global__ __launch_bounds__(256) void k_sample_write(const float *__restrict__ logits, int ldl,
                                                      const float *__restrict__ values, int ldv, int B, int A, int T, int t,
                                                      uint32_t seed, uint32_t step, uint32_t row0, float version,
                                                      int deterministic, int action_kind,
                                                      float *__restrict__ t_actions,
                                                      float *__restrict__ t_logits, float *__restrict__ t_logp,
                                                      float *__restrict__ t_values, float *__restrict__ t_version,
                                                      int32_t *__restrict__ env_actions) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float *z = logits + (int64_t)b * ldl;
    if (action_kind == 1) {  // Box(D): params = [means | log_std]; a = mu + sd*eps, eps from Box-Muller on Philox
        const int D = A / 2;
        const int64_t it = (int64_t)b * T + t;
        float lp = 0.f;
        for (int k = 0; k < D; ++k) {
            uint32_t w[4];
            sf_philox4x32_10(step, (uint32_t)(k >> 1), 3u, 0u, seed, row0 + (uint32_t)b, w);
            const uint32_t w1 = (k & 1) ? w[2] : w[0], w2 = (k & 1) ? w[3] : w[1];
            const float u1 = ((float)(w1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float u2 = (float)(w2 >> 8) * (1.0f / 16777216.0f);
            const float eps = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
            const float mu = z[k];
            const float sd = clampf(expf(z[D + k]), 1e-4f, 1e4f);
            const float a = deterministic ? mu : mu + sd * eps;  // argmax_actions: the mean (action_distributions.py:78-79)
            t_actions[it * D + k] = a;
            lp += -((a - mu) * (a - mu)) / (2.f * (sd * sd)) - logf(sd) - 0.91893853320467274178f;
        }
        for (int k = 0; k < A; ++k) t_logits[it * A + k] = z[k];
        t_logp[it] = lp;
        t_version[it] = version;
        t_values[(int64_t)b * (T + 1) + t] = values[(int64_t)b * ldv];
        return;
    }
    float zz[MAXA];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < MAXA; ++k) { zz[k] = k < A ? z[k] : -INFINITY; mx = fmaxf(mx, zz[k]); }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < MAXA; ++k) if (k < A) se += expf(zz[k] - mx);
    const float lse = logf(se);
    int a = A - 1;
    if (deterministic) {
        bool found = false;
#pragma unroll
        for (int k = 0; k < MAXA; ++k) if (k < A && !found && zz[k] == mx) { a = k; found = true; }
    } else {
        uint32_t w[4];
        sf_philox4x32_10(step, 0u, 2u, 0u, seed, row0 + (uint32_t)b, w);
        const float u = (float)(w[0] >> 8) * (1.0f / 16777216.0f);
        float acc = 0.f;
        bool found = false;
#pragma unroll
        for (int k = 0; k < MAXA; ++k)
            if (k < A && !found) {
                acc += expf((zz[k] - mx) - lse);
                if (u < acc) { a = k; found = true; }
            }
    }
    float lp = 0.f;
#pragma unroll
    for (int k = 0; k < MAXA; ++k) if (k == a) lp = (zz[k] - mx) - lse;
    const int64_t it = (int64_t)b * T + t;
    t_actions[it] = (float)a;
    t_logp[it] = lp;
    t_version[it] = version;
    t_values[(int64_t)b * (T + 1) + t] = values[(int64_t)b * ldv];
#pragma unroll
    for (int k = 0; k < MAXA; ++k) if (k < A) t_logits[it * A + k] = zz[k];
    if (env_actions) env_actions[b] = a;
}

extern "C" int sf_sample_write_step(const float *logits, int ld_logits, const float *values, int ld_values, int B,
                                    int A, int T, int t, uint32_t seed, uint32_t step, uint32_t row0, float policy_version,
                                    int deterministic, int action_kind, float *traj_actions, float *traj_logits, float *traj_logp,
                                    float *traj_values, float *traj_policy_version, int32_t *env_actions,
                                    void *stream) {
    SF_REQUIRE(logits && values && traj_actions && traj_logits && traj_logp && traj_values && traj_policy_version,
               "sf_sample_write_step: null pointer");
    SF_REQUIRE(action_kind == 1 || env_actions, "sf_sample_write_step: discrete actions need the int32 env_actions");
    SF_REQUIRE(action_kind == 0 || (action_kind == 1 && A % 2 == 0), "sf_sample_write_step: bad action_kind");
    SF_REQUIRE(B > 0 && A > 0 && A <= 128 && T > 0 && t >= 0 && t < T, "sf_sample_write_step: bad shape B=%d A=%d T=%d t=%d",
               B, A, T, t);
    const dim3 grid((unsigned)((B + 255) / 256)), block(256);
#define SW_LAUNCH(M)                                                                                               \
    k_sample_write<M><<<grid, block, 0, STREAM(stream)>>>(logits, ld_logits, values, ld_values, B, A, T, t, seed, step, row0,            \
                                                          policy_version, deterministic, action_kind, traj_actions, traj_logits, \
                                                          traj_logp, traj_values, traj_policy_version, env_actions)
    if (A <= 8) SW_LAUNCH(8);
    else if (A <= 32) SW_LAUNCH(32);
    else SW_LAUNCH(128);
#undef SW_LAUNCH
    return sf_launch_status("sf_sample_write_step");
}


// Discrete(A) with an action mask (obs["action_mask"], inference_worker.py:324-331; action_distributions.py:84-96,
// 110-142): logits of masked-out actions get -1e9, probs = softmax * mask / (sum + 1e-13), an all-zero row falls back to
// uniform (the reference substitutes 1e-6 everywhere before torch.multinomial), log-prob from the masked log-softmax.
// The RAW logits go to the trajectory (the learner re-evaluates the unmasked distribution, as the reference does).
__global__ __launch_bounds__(256) void k_sample_write_masked(const float *__restrict__ logits, int ldl,
                                                             const float *__restrict__ values, int ldv,
                                                             const uint8_t *__restrict__ mask, int64_t ldm, int B, int A,
                                                             int T, int t, uint32_t seed, uint32_t step, uint32_t row0,
                                                             float version, int deterministic,
                                                             float *__restrict__ t_actions, float *__restrict__ t_logits,
                                                             float *__restrict__ t_logp, float *__restrict__ t_values,
                                                             float *__restrict__ t_version,
                                                             int32_t *__restrict__ env_actions) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float *z = logits + (int64_t)b * ldl;
    const uint8_t *mk = mask + (int64_t)b * ldm;
    float mx = -INFINITY;
    for (int k = 0; k < A; ++k) mx = fmaxf(mx, z[k] + (mk[k] ? 0.f : -1e9f));
    float se = 0.f;
    for (int k = 0; k < A; ++k) se += expf((z[k] + (mk[k] ? 0.f : -1e9f)) - mx);
    const float lse = logf(se);
    float psum = 0.f;
    for (int k = 0; k < A; ++k) psum += mk[k] ? expf(((z[k]) - mx) - lse) : 0.f;
    const bool all_zero = psum == 0.f;
    const float inv = 1.0f / (psum + 1e-13f);
    int a = A - 1;
    if (deterministic) {
        float best = -1.f;
        for (int k = 0; k < A; ++k) {
            const float p = all_zero ? 1e-6f : (mk[k] ? expf((z[k] - mx) - lse) * inv : 0.f);
            if (p > best) { best = p; a = k; }
        }
    } else {
        uint32_t w[4];
        sf_philox4x32_10(step, 0u, 2u, 0u, seed, row0 + (uint32_t)b, w);
        const float u = (float)(w[0] >> 8) * (1.0f / 16777216.0f);
        float acc = 0.f;
        const float tot = all_zero ? (float)A * 1e-6f : psum * inv;  // multinomial normalises its weights
        for (int k = 0; k < A; ++k) {
            const float p = all_zero ? 1e-6f : (mk[k] ? expf((z[k] - mx) - lse) * inv : 0.f);
            acc += p / tot;
            if (u < acc) { a = k; break; }
        }
        if (!all_zero && !mk[a]) {  // rounding left u beyond the last allowed action: take the last allowed one
            for (int k = A - 1; k >= 0; --k) if (mk[k]) { a = k; break; }
        }
    }
    const int64_t it = (int64_t)b * T + t;
    t_actions[it] = (float)a;
    t_logp[it] = ((z[a] + (mk[a] ? 0.f : -1e9f)) - mx) - lse;
    t_version[it] = version;
    t_values[(int64_t)b * (T + 1) + t] = values[(int64_t)b * ldv];
    for (int k = 0; k < A; ++k) t_logits[it * A + k] = z[k];
    if (env_actions) env_actions[b] = a;
}

extern "C" int sf_sample_write_step_masked(const float *logits, int ld_logits, const float *values, int ld_values,
                                           const uint8_t *action_mask, int64_t ld_mask, int B, int A, int T, int t,
                                           uint32_t seed, uint32_t step, uint32_t row0, float policy_version,
                                           int deterministic, float *traj_actions, float *traj_logits, float *traj_logp,
                                           float *traj_values, float *traj_policy_version, int32_t *env_actions,
                                           void *stream) {
    SF_REQUIRE(logits && values && action_mask && traj_actions && traj_logits && traj_logp && traj_values &&
                   traj_policy_version, "sf_sample_write_step_masked: null pointer");
    SF_REQUIRE(B > 0 && A > 0 && T > 0 && t >= 0 && t < T && ld_logits >= A && ld_values >= 1 && ld_mask >= A,
               "sf_sample_write_step_masked: bad shape B=%d A=%d T=%d t=%d", B, A, T, t);
    k_sample_write_masked<<<dim3((unsigned)((B + 255) / 256)), dim3(256), 0, STREAM(stream)>>>(
        logits, ld_logits, values, ld_values, action_mask, ld_mask, B, A, T, t, seed, step, row0, policy_version,
        deterministic, traj_actions, traj_logits, traj_logp, traj_values, traj_policy_version, env_actions);
    return sf_launch_status("sf_sample_write_step_masked");
}

// Tuple of Discrete heads: every head is sampled from its own Philox uniform (counter lane 1 = head index)
__global__ __launch_bounds__(256) void k_sample_write_tuple(const float *__restrict__ logits, int ldl,
                                                            const float *__restrict__ values, int ldv, int B, int A, int T,
                                                            int t, uint32_t seed, uint32_t step, uint32_t row0,
                                                            float version, int deterministic, LossDev hd,
                                                            float *__restrict__ t_actions, float *__restrict__ t_logits,
                                                            float *__restrict__ t_logp, float *__restrict__ t_values,
                                                            float *__restrict__ t_version,
                                                            int32_t *__restrict__ env_actions) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const float *z = logits + (int64_t)b * ldl;
    const int64_t it = (int64_t)b * T + t;
    const int H = hd.num_heads;
    const int NA = heads_action_cols(hd.head_n, H);
    float lp_sum = 0.f;
    int off = 0, aoff = 0;
    for (int h = 0; h < H; ++h) {
        const int nh = hd.head_n[h];
        if (nh < 0) {  // Box(D) member: a = mu + sd * eps, eps from Box-Muller on Philox counter (step, k / 2, 3, h)
            const int D = -nh;
            for (int k = 0; k < D; ++k) {
                uint32_t w[4];
                sf_philox4x32_10(step, (uint32_t)(k >> 1), 3u, (uint32_t)h, seed, row0 + (uint32_t)b, w);
                const uint32_t w1 = (k & 1) ? w[2] : w[0], w2 = (k & 1) ? w[3] : w[1];
                const float u1 = ((float)(w1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
                const float u2 = (float)(w2 >> 8) * (1.0f / 16777216.0f);
                const float eps = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
                const float mu = z[off + k];
                const float sd = clampf(expf(z[off + D + k]), 1e-4f, 1e4f);
                const float a = deterministic ? mu : mu + sd * eps;
                t_actions[it * NA + aoff + k] = a;
                lp_sum += -((a - mu) * (a - mu)) / (2.f * (sd * sd)) - logf(sd) - 0.91893853320467274178f;
            }
            off += 2 * D; aoff += D;
            continue;
        }
        float mx = -INFINITY;
        for (int k = 0; k < nh; ++k) mx = fmaxf(mx, z[off + k]);
        float se = 0.f;
        for (int k = 0; k < nh; ++k) se += expf(z[off + k] - mx);
        const float lse = logf(se);
        int a = nh - 1;
        if (deterministic) {
            for (int k = nh - 1; k >= 0; --k) if (z[off + k] == mx) a = k;  // first maximum, as torch.argmax
        } else {
            uint32_t w[4];
            sf_philox4x32_10(step, (uint32_t)h, 2u, 0u, seed, row0 + (uint32_t)b, w);
            const float u = (float)(w[0] >> 8) * (1.0f / 16777216.0f);
            float acc = 0.f;
            for (int k = 0; k < nh; ++k) {
                acc += expf((z[off + k] - mx) - lse);
                if (u < acc) { a = k; break; }
            }
        }
        lp_sum += (z[off + a] - mx) - lse;
        t_actions[it * NA + aoff] = (float)a;
        if (env_actions) env_actions[(int64_t)b * H + h] = a;  // all-Discrete tuples only (the launcher checks)
        off += nh; aoff += 1;
    }
    for (int k = 0; k < A; ++k) t_logits[it * A + k] = z[k];
    t_logp[it] = lp_sum;
    t_version[it] = version;
    t_values[(int64_t)b * (T + 1) + t] = values[(int64_t)b * ldv];
}

extern "C" int sf_sample_write_step_tuple(const float *logits, int ld_logits, const float *values, int ld_values, int B,
                                          int num_heads, const int32_t *head_n, int T, int t, uint32_t seed,
                                          uint32_t step, uint32_t row0, float policy_version, int deterministic,
                                          float *traj_actions, float *traj_logits, float *traj_logp, float *traj_values,
                                          float *traj_policy_version, int32_t *env_actions, void *stream) {
    SF_REQUIRE(logits && values && head_n && traj_actions && traj_logits && traj_logp && traj_values &&
                   traj_policy_version, "sf_sample_write_step_tuple: null pointer");
    SF_REQUIRE(num_heads >= 1 && num_heads <= 8 && B > 0 && T > 0 && t >= 0 && t < T,
               "sf_sample_write_step_tuple: bad shape heads=%d B=%d T=%d t=%d", num_heads, B, T, t);
    LossDev hd = {};
    hd.num_heads = num_heads;
    int A = 0;
    for (int i = 0; i < num_heads; ++i) {  // head_n > 0: Discrete(n); head_n < 0: Box(-n) member, 2 * (-n) parameters
        SF_REQUIRE(head_n[i] != 0, "sf_sample_write_step_tuple: empty action head");
        SF_REQUIRE(head_n[i] > 0 || !env_actions,
                   "sf_sample_write_step_tuple: a tuple with a Box member has no int32 env_actions (the env reads the "
                   "trajectory's f32 action row)");
        hd.head_n[i] = head_n[i];
        A += head_n[i] > 0 ? head_n[i] : -2 * head_n[i];
    }
    SF_REQUIRE(ld_logits >= A && ld_values >= 1, "sf_sample_write_step_tuple: bad strides");
    k_sample_write_tuple<<<dim3((unsigned)((B + 255) / 256)), dim3(256), 0, STREAM(stream)>>>(
        logits, ld_logits, values, ld_values, B, A, T, t, seed, step, row0, policy_version, deterministic, hd,
        traj_actions, traj_logits, traj_logp, traj_values, traj_policy_version, env_actions);
    return sf_launch_status("sf_sample_write_step_tuple");
}

// =========================================================================================== K1/K6 env step -> traj
__global__ __launch_bounds__(256) void k_traj_write_env(const float *__restrict__ rewards,
                                                        const uint8_t *__restrict__ terminated,
                                                        const uint8_t *__restrict__ truncated, int B, int T, int t,
                                                        float scale, float clip, int pid,
                                                        float *__restrict__ t_rewards, uint8_t *__restrict__ t_dones,
                                                        uint8_t *__restrict__ t_timeouts, int32_t *__restrict__ t_pid,
                                                        float *__restrict__ ep_ret, int32_t *__restrict__ ep_len,
                                                        double *__restrict__ ep_stats) {
    __shared__ double lds[4 * 3];
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    double acc[3] = {0.0, 0.0, 0.0};
    if (b < B) {
        const float raw = rewards[b];
        const bool term = terminated[b] != 0, trunc = truncated ? truncated[b] != 0 : false;
        const bool done = term || trunc;
        const int64_t it = (int64_t)b * T + t;
        t_rewards[it] = clampf(raw * scale, -clip, clip);  // batched_sampling.py:208-213
        t_dones[it] = (uint8_t)done;
        t_timeouts[it] = (uint8_t)trunc;
        t_pid[it] = pid;
        if (ep_ret) {
            const float r = ep_ret[b] + raw;  // episode stats use the raw reward (batched_sampling.py:216-219)
            const int l = ep_len[b] + 1;
            if (done) { acc[0] = r; acc[1] = (double)l; acc[2] = 1.0; ep_ret[b] = 0.f; ep_len[b] = 0; }
            else { ep_ret[b] = r; ep_len[b] = l; }
        }
    }
    if (ep_stats) {
        sf_block_sum<3>(acc, lds);
        if (threadIdx.x == 0 && acc[2] > 0.0) {
            atomicAdd(&ep_stats[0], acc[0]);
            atomicAdd(&ep_stats[1], acc[1]);
            atomicAdd(&ep_stats[2], acc[2]);
        }
    }
}

extern "C" int sf_traj_write_env_step(const float *rewards, const uint8_t *terminated, const uint8_t *truncated, int B,
                                      int T, int t, float reward_scale, float reward_clip, int policy_id,
                                      float *traj_rewards, uint8_t *traj_dones, uint8_t *traj_time_outs,
                                      int32_t *traj_policy_id, float *ep_return, int32_t *ep_len, double *ep_stats,
                                      void *stream) {
    SF_REQUIRE(rewards && terminated && traj_rewards && traj_dones && traj_time_outs && traj_policy_id,
               "sf_traj_write_env_step: null pointer");
    SF_REQUIRE(B > 0 && T > 0 && t >= 0 && t < T, "sf_traj_write_env_step: bad shape");
    SF_REQUIRE((ep_return == nullptr) == (ep_len == nullptr), "sf_traj_write_env_step: ep_return/ep_len mismatch");
    k_traj_write_env<<<dim3((unsigned)((B + 255) / 256)), dim3(256), 0, STREAM(stream)>>>(
        rewards, terminated, truncated, B, T, t, reward_scale, reward_clip, policy_id, traj_rewards, traj_dones,
        traj_time_outs, traj_policy_id, ep_return, ep_len, ep_stats);
    return sf_launch_status("sf_traj_write_env_step");
}

// =========================================================================================== synthetic env
__global__ __launch_bounds__(256) void k_synth_obs(uint8_t *__restrict__ obs, int64_t env_stride, int B, int env0,
                                                   int64_t blocks_per_env, uint32_t seed, uint32_t step) {
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (int64_t)B * blocks_per_env) return;
    const int64_t e = gid / blocks_per_env, blk = gid - e * blocks_per_env;
    uint32_t w[4];
    sf_philox4x32_10(step, (uint32_t)blk, 0u, 0u, seed, (uint32_t)(env0 + e), w);
    *reinterpret_cast<uint4 *>(obs + e * env_stride + blk * 16) = make_uint4(w[0], w[1], w[2], w[3]);
}

extern "C" int sf_synth_obs(uint8_t *obs, int64_t env_stride, int B, int env0, int64_t obs_bytes, uint32_t seed,
                            uint32_t step, void *stream) {
    SF_REQUIRE(obs && B > 0 && obs_bytes > 0 && obs_bytes % 16 == 0 && env_stride % 16 == 0 &&
                   ((uintptr_t)obs & 15) == 0,
               "sf_synth_obs: obs_bytes/env_stride/base must be multiples of 16");
    const int64_t total = (int64_t)B * (obs_bytes / 16);
    k_synth_obs<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, STREAM(stream)>>>(obs, env_stride, B, env0,
                                                                                         obs_bytes / 16, seed, step);
    return sf_launch_status("sf_synth_obs");
}

__global__ __launch_bounds__(256) void k_synth_step(const int32_t *__restrict__ actions, int B, int env0,
                                                    int num_actions, uint32_t seed, uint32_t step,
                                                    float *__restrict__ rewards, uint8_t *__restrict__ terminated) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint32_t env = (uint32_t)(env0 + b);
    rewards[b] = (actions[b] == (int32_t)((step + env) % (uint32_t)num_actions)) ? 1.0f : 0.0f;
    uint32_t w[4];
    sf_philox4x32_10(step, 0u, 1u, 0u, seed, env, w);
    terminated[b] = (uint8_t)(w[0] < (1u << 22));
}

extern "C" int sf_synth_step(const int32_t *actions, int B, int env0, int num_actions, uint32_t seed, uint32_t step,
                             float *rewards, uint8_t *terminated, void *stream) {
    SF_REQUIRE(actions && rewards && terminated && B > 0 && num_actions > 0, "sf_synth_step: bad args");
    k_synth_step<<<dim3((unsigned)((B + 255) / 256)), dim3(256), 0, STREAM(stream)>>>(actions, B, env0, num_actions,
                                                                                      seed, step, rewards, terminated);
    return sf_launch_status("sf_synth_step");
}

// ---- Ant-shaped continuous synthetic env (SURVEY.md §8d C5 stand-in), one lane per env:
// obs' = terminated ? noise : 0.9 * obs + 0.1 * noise, reward = -mean(a^2) + 0.1 * obs[0], terminated ~ Bernoulli(1/256);
// noise ~ N(0, 1) from Philox4x32-10 keyed by (seed, env), counter (step, block) via Box-Muller.  `state` is the env's
// own copy of the observation; `obs_out` (row stride out_stride floats) is slot t+1 of the trajectory slab.
__device__ __forceinline__ void synth_normals(uint32_t seed, uint32_t env, uint32_t step, uint32_t blk, float (&z)[4]) {
    uint32_t w[4];
    sf_philox4x32_10(step, blk, 2u, 0u, seed, env, w);
    const float u0 = ((float)(w[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = (float)(w[1] >> 8) * (1.0f / 16777216.0f);
    const float u2 = ((float)(w[2] >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = (float)(w[3] >> 8) * (1.0f / 16777216.0f);
    const float r0 = sqrtf(-2.0f * logf(u0)), r1 = sqrtf(-2.0f * logf(u2));
    z[0] = r0 * cosf(6.2831853f * u1); z[1] = r0 * sinf(6.2831853f * u1);
    z[2] = r1 * cosf(6.2831853f * u3); z[3] = r1 * sinf(6.2831853f * u3);
}

// one lane per (env, block of 4 observation dims): 7 lanes per env at D = 27
__global__ __launch_bounds__(256) void k_synth_vec_step(float *__restrict__ state, const float *__restrict__ actions,
                                                        int64_t act_stride, float *__restrict__ obs_out, int64_t out_stride,
                                                        int B, int D, int A, int env0, uint32_t seed, uint32_t step,
                                                        int reset, float *__restrict__ rewards,
                                                        uint8_t *__restrict__ terminated) {
    const int nblk = (D + 3) >> 2;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= B * nblk) return;
    const int b = gid / nblk, blk = gid - b * nblk, d0 = blk * 4;
    const uint32_t env = (uint32_t)(env0 + b);
    float *st = state + (int64_t)b * D, *out = obs_out + (int64_t)b * out_stride;
    bool term = reset != 0;
    if (!reset) {
        uint32_t w[4];
        sf_philox4x32_10(step, 0u, 1u, 0u, seed, env, w);  // (every lane of the env draws the same word)
        term = w[0] < (1u << 24);  // 1/256
        if (blk == 0) {
            float sq = 0.0f;
            for (int a = 0; a < A; ++a) {
                const float v = actions[(int64_t)b * act_stride + a];
                sq += v * v;
            }
            rewards[b] = -(sq / (float)A) + 0.1f * st[0];
            terminated[b] = (uint8_t)term;
        }
    }
    float z[4];
    synth_normals(seed, env, step, (uint32_t)blk, z);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (d0 + k < D) {
            const float v = term ? z[k] : 0.9f * st[d0 + k] + 0.1f * z[k];
            st[d0 + k] = v;
            out[d0 + k] = v;
        }
    }
}

extern "C" int sf_synth_vec_step(float *state, const float *actions, int64_t act_stride, float *obs_out, int64_t out_stride,
                                 int B, int D, int A, int env0, uint32_t seed, uint32_t step, int reset, float *rewards,
                                 uint8_t *terminated, void *stream) {
    SF_REQUIRE(state && obs_out && B > 0 && D > 0 && (reset || (actions && rewards && terminated && A > 0)),
               "sf_synth_vec_step: bad args");
    const int64_t lanes = (int64_t)B * ((D + 3) / 4);
    k_synth_vec_step<<<dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, STREAM(stream)>>>(
        state, actions, act_stride, obs_out, out_stride, B, D, A, env0, seed, step, reset, rewards, terminated);
    return sf_launch_status("sf_synth_vec_step");
}

// =========================================================================================== observation normaliser
// utils/normalize.py:24-70 (ObservationNormalizer) + running_mean_std.py:22-136 (RunningMeanStd(Dict)InPlace with
// full-shape statistics): x' = (float(x) - obs_subtract_mean) * (1/obs_scale); per-element running mean/var/count
// (f64, Chan merge); normalised = clamp((x' - mu) * (1/sqrt(var + 1e-5)), +-5).
// Sample addressing is the same as the network kernels': logical sample i -> row (index ? index[i] : offset + i),
// optionally mapped dataset->slab (e*T+t -> e*(T+1)+t), times `stride` elements.
__device__ __forceinline__ int64_t on_row(const int32_t *__restrict__ index, int64_t offset, int traj_T, int64_t i) {
    int64_t d = index ? (int64_t)index[i] : offset + i;
    if (traj_T > 0) d += d / traj_T;
    return d;
}

template <bool U8>
__global__ __launch_bounds__(256) void k_obsnorm_moments(const void *__restrict__ in, int64_t stride,
                                                         const int32_t *__restrict__ index, int64_t offset, int traj_T,
                                                         int64_t n, int D, float sub_mean, float inv_scale,
                                                         double *__restrict__ sum, double *__restrict__ sumsq) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const int64_t per = (n + gridDim.y - 1) / gridDim.y;
    const int64_t beg = (int64_t)blockIdx.y * per, end = (beg + per < n) ? beg + per : n;
    double s = 0.0, ss = 0.0;
    for (int64_t i = beg; i < end; ++i) {
        const int64_t row = on_row(index, offset, traj_T, i);
        const float raw = U8 ? (float)reinterpret_cast<const uint8_t *>(in)[row * stride + d]
                             : reinterpret_cast<const float *>(in)[row * stride + d];
        const double x = (double)((raw - sub_mean) * inv_scale);
        s += x;
        ss += x * x;
    }
    atomicAdd(&sum[d], s);
    atomicAdd(&sumsq[d], ss);
}

extern "C" int sf_obsnorm_moments(const void *in, int in_u8, int64_t stride, const int32_t *index, int64_t offset,
                                  int traj_T, int64_t n, int D, float sub_mean, float inv_scale, double *sum,
                                  double *sumsq, void *stream) {
    SF_REQUIRE(in && sum && sumsq && n > 0 && D > 0, "sf_obsnorm_moments: bad args");
    int rc = sf_hip_status(hipMemsetAsync(sum, 0, sizeof(double) * D, STREAM(stream)), "sf_obsnorm_moments memset");
    if (!rc) rc = sf_hip_status(hipMemsetAsync(sumsq, 0, sizeof(double) * D, STREAM(stream)), "sf_obsnorm_moments memset");
    if (rc) return rc;
    const unsigned bx = (unsigned)((D + 255) / 256);
    unsigned by = (unsigned)(4096 / bx);
    if (by < 1) by = 1;
    if ((int64_t)by > n) by = (unsigned)n;
    if (in_u8) k_obsnorm_moments<true><<<dim3(bx, by), dim3(256), 0, STREAM(stream)>>>(in, stride, index, offset, traj_T, n, D, sub_mean, inv_scale, sum, sumsq);
    else k_obsnorm_moments<false><<<dim3(bx, by), dim3(256), 0, STREAM(stream)>>>(in, stride, index, offset, traj_T, n, D, sub_mean, inv_scale, sum, sumsq);
    return sf_launch_status("sf_obsnorm_moments");
}

// stats: mean[D], var[D] (f64), count[1] (f64); batch moments {sum, sumsq}[D] over n samples -> merged stats (in
// place; count advanced by the launch with blockIdx 0) and the derived f32 tables mu[D], rstd[D] the apply kernel uses.
__global__ __launch_bounds__(256) void k_obsnorm_update(double *__restrict__ mean, double *__restrict__ var,
                                                        const double *__restrict__ count_in,
                                                        double *__restrict__ count_out, const double *__restrict__ sum,
                                                        const double *__restrict__ sumsq, double n, int D,
                                                        float *__restrict__ mu_tab, float *__restrict__ rstd_tab) {
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    const double count = count_in[0];
    if (d < D) {
        double m = mean[d], v = var[d];
        if (n > 0.0) {
            const double bm64 = sum[d] / n;
            const double bv64 = (sumsq[d] - sum[d] * bm64) / (n - 1.0);
            const double bm = (double)(float)bm64, bv = (double)(float)bv64;  // x.mean(0), x.var(0) are f32 tensors
            const double delta = bm - m, tot = count + n;
            const double M2 = v * count + bv * n + (delta * delta) * count * n / tot;
            m = m + delta * n / tot;
            v = M2 / tot;
            mean[d] = m;
            var[d] = v;
        }
        mu_tab[d] = (float)m;
        rstd_tab[d] = 1.0f / sqrtf((float)v + 1e-5f);
    }
    if (d == 0) count_out[0] = count + n;
}

extern "C" int sf_obsnorm_update(double *mean, double *var, const double *count_in, double *count_out,
                                 const double *sum, const double *sumsq, int64_t n, int D, float *mu_tab,
                                 float *rstd_tab, void *stream) {
    SF_REQUIRE(mean && var && count_in && count_out && mu_tab && rstd_tab && D > 0 && n >= 0 && (n == 0 || (sum && sumsq)),
               "sf_obsnorm_update: bad args");
    SF_REQUIRE(count_in != count_out, "sf_obsnorm_update: count_in and count_out must be different buffers");
    k_obsnorm_update<<<dim3((unsigned)((D + 255) / 256)), dim3(256), 0, STREAM(stream)>>>(
        mean, var, count_in, count_out, sum, sumsq, (double)n, D, mu_tab, rstd_tab);
    return sf_launch_status("sf_obsnorm_update");
}

// out[i][pos] = clamp(((x - sub_mean)*inv_scale - mu[d]) * rstd[d], +-5); images (C,H,W given, C>0): d = c*HW + p is
// written channels-last at pos = p*C + c (the layout the conv kernels read); vectors (C == 0): pos = d.
template <bool U8>
__global__ __launch_bounds__(256) void k_obsnorm_apply(const void *__restrict__ in, int64_t stride,
                                                       const int32_t *__restrict__ index, int64_t offset, int traj_T,
                                                       int64_t n, int D, int C, int HW, float sub_mean,
                                                       float inv_scale, const float *__restrict__ mu,
                                                       const float *__restrict__ rstd, float *__restrict__ out) {
    const int64_t total = n * D;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = g / D;
        const int pos = (int)(g - i * D);
        int d = pos;
        if (C > 0) { const int p = pos / C, c = pos - p * C; d = c * HW + p; }
        const int64_t row = on_row(index, offset, traj_T, i);
        const float raw = U8 ? (float)reinterpret_cast<const uint8_t *>(in)[row * stride + d]
                             : reinterpret_cast<const float *>(in)[row * stride + d];
        const float x = (raw - sub_mean) * inv_scale;
        out[g] = clampf((x - mu[d]) * rstd[d], -5.0f, 5.0f);
    }
}

extern "C" int sf_obsnorm_apply(const void *in, int in_u8, int64_t stride, const int32_t *index, int64_t offset,
                                int traj_T, int64_t n, int D, int C, int HW, float sub_mean, float inv_scale,
                                const float *mu, const float *rstd, float *out, void *stream) {
    SF_REQUIRE(in && mu && rstd && out && n > 0 && D > 0, "sf_obsnorm_apply: bad args");
    SF_REQUIRE(C == 0 || C * HW == D, "sf_obsnorm_apply: C*HW != D");
    const int64_t blocks = (n * D + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 65536 ? blocks : 65536);
    if (in_u8) k_obsnorm_apply<true><<<dim3(grid), dim3(256), 0, STREAM(stream)>>>(in, stride, index, offset, traj_T, n, D, C, HW, sub_mean, inv_scale, mu, rstd, out);
    else k_obsnorm_apply<false><<<dim3(grid), dim3(256), 0, STREAM(stream)>>>(in, stride, index, offset, traj_T, n, D, C, HW, sub_mean, inv_scale, mu, rstd, out);
    return sf_launch_status("sf_obsnorm_apply");
}

// =========================================================================================== fused small MLP encoder
// Inference on VECTOR observations (model/encoder.py:72-87 MlpEncoder behind utils/normalize.py:24-70): input
// normalisation -> Linear(D, H1) + act -> Linear(H1, H2) + act in ONE launch.  At rollout sizes (2048 envs x 27 floats)
// each of these layers is a 13 us MFMA-kernel launch doing 7 MFLOP; here a 256-thread block takes 16 samples, keeps both
// weight matrices in LDS and each thread produces 4 columns of one row with fmaf chains in ascending k — the same
// arithmetic (bit for bit) as the exact-f32 MFMA kernels, which the training pass keeps using (it needs the
// intermediate activations).
__device__ __forceinline__ float mlp_act(float x, int kind) {
    if (kind == 1) return fmaxf(x, 0.f);
    if (kind == 2) return tanhf(x);
    if (kind == 3) return x > 0.f ? x : expm1f(x);
    return x;
}

constexpr int MLP2_ROWS = 16;  // samples per block (2048 envs: 128 blocks); a thread owns 1 row x 4 columns

__global__ __launch_bounds__(256) void k_mlp2_fwd(const float *__restrict__ x, int64_t x_stride, int64_t n, int D,
                                                  float sub_mean, float inv_scale, const float *__restrict__ mu,
                                                  const float *__restrict__ rstd, const float *__restrict__ w1,
                                                  const float *__restrict__ b1, int H1, const float *__restrict__ w2,
                                                  const float *__restrict__ b2, int H2, int act,
                                                  float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float *sw1 = sm, *sw2 = sw1 + D * H1, *sx = sw2 + H1 * H2, *sh = sx + MLP2_ROWS * D;  // sh: [ROWS][H1]
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * MLP2_ROWS;
    // every global load of the prologue is issued before the first LDS store (a load -> store loop waits for each load
    // in turn: 30 serialised round trips were 14 of the first version's 18 us)
    {
        constexpr int NV = 8;  // float4s per thread and matrix: up to 8192 weights each (checked by the launcher)
        float4 v1[NV], v2[NV];
        const int n1 = D * H1 / 4, n2 = H1 * H2 / 4;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int q1 = tid + 256 * i, q2 = tid + 256 * i;
            v1[i] = reinterpret_cast<const float4 *>(w1)[q1 < n1 ? q1 : 0];
            v2[i] = reinterpret_cast<const float4 *>(w2)[q2 < n2 ? q2 : 0];
        }
        constexpr int NX = 4;  // observation words per thread: up to 16 rows x 64 values
        float xv[NX];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int q = tid + 256 * i;
            const int qq = q < MLP2_ROWS * D ? q : 0;
            const int r = qq / D, d = qq - r * D;
            const int64_t row = r0 + r < n ? r0 + r : n - 1;
            float v = (x[row * x_stride + d] - sub_mean) * inv_scale;
            if (mu) v = fminf(fmaxf((v - mu[d]) * rstd[d], -5.0f), 5.0f);
            xv[i] = v;
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int q = tid + 256 * i;
            if (q < n1) reinterpret_cast<float4 *>(sw1)[q] = v1[i];
            if (q < n2) reinterpret_cast<float4 *>(sw2)[q] = v2[i];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int q = tid + 256 * i;
            if (q < MLP2_ROWS * D) sx[q] = xv[i];
        }
    }
    __syncthreads();
    const int ra = tid >> 4, cg = tid & 15;
    auto layer = [&](const float *in, int K, const float *w, const float *bias, int N, float *dst, int64_t dst_ld,
                     bool to_global) {
        for (int c0 = cg * 4; c0 < N; c0 += 64) {  // 4 consecutive columns per pass
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
            for (int k = 0; k < K; ++k) {  // ascending k, one fmaf chain per output element (= the MFMA kernels' sums)
                const float a = in[ra * K + k];
                const float4 wv = *reinterpret_cast<const float4 *>(w + k * N + c0);
                acc[0] = fmaf(a, wv.x, acc[0]);
                acc[1] = fmaf(a, wv.y, acc[1]);
                acc[2] = fmaf(a, wv.z, acc[2]);
                acc[3] = fmaf(a, wv.w, acc[3]);
            }
            if (!to_global || r0 + ra < n) {
                float4 o;
                o.x = mlp_act(acc[0] + bias[c0], act);
                o.y = mlp_act(acc[1] + bias[c0 + 1], act);
                o.z = mlp_act(acc[2] + bias[c0 + 2], act);
                o.w = mlp_act(acc[3] + bias[c0 + 3], act);
                *reinterpret_cast<float4 *>(dst + (to_global ? r0 + ra : ra) * dst_ld + c0) = o;
            }
        }
    };
    layer(sx, D, sw1, b1, H1, sh, H1, false);
    __syncthreads();
    layer(sh, H1, sw2, b2, H2, out, H2, true);
}

extern "C" int sf_mlp2_fwd(const float *x, int64_t x_stride, int64_t n, int D, float sub_mean, float inv_scale,
                           const float *mu, const float *rstd, const float *w1, const float *b1, int H1, const float *w2,
                           const float *b2, int H2, int act, float *out, void *stream) {
    SF_REQUIRE(x && w1 && b1 && w2 && b2 && out && n > 0 && D > 0, "sf_mlp2_fwd: bad args");
    SF_REQUIRE(H1 > 0 && H2 > 0 && H1 % 8 == 0 && H2 % 8 == 0 && (mu == nullptr) == (rstd == nullptr),
               "sf_mlp2_fwd: layer widths must be multiples of 8 (H1=%d H2=%d)", H1, H2);
    const size_t lds = sizeof(float) * ((size_t)D * H1 + (size_t)H1 * H2 + MLP2_ROWS * (size_t)(D + H1));
    SF_REQUIRE(lds <= 64 * 1024 && D <= 64 && D * H1 <= 8192 && H1 * H2 <= 8192 && ((uintptr_t)w1 & 15) == 0 && ((uintptr_t)w2 & 15) == 0 &&
                   ((uintptr_t)out & 15) == 0,
               "sf_mlp2_fwd: D=%d H1=%d H2=%d exceed the fused kernel (use the layer kernels)", D, H1, H2);
    k_mlp2_fwd<<<dim3((unsigned)((n + MLP2_ROWS - 1) / MLP2_ROWS)), dim3(256), lds, STREAM(stream)>>>(x, x_stride, n, D, sub_mean, inv_scale, mu,
                                                                                   rstd, w1, b1, H1, w2, b2, H2, act, out);
    return sf_launch_status("sf_mlp2_fwd");
}

// =========================================================================================== recurrent core cells
// model/core.py:19-64 (ModelCoreRNN = torch.nn.GRU / nn.LSTM, one layer) as explicit cell kernels; the projections
// (x W_ih^T + b_ih, h W_hh^T + b_hh) are the MFMA linear kernels, these are the elementwise halves.  BPTT runs as a
// time loop over recurrence-length chunks with the state zeroed after a done/invalid step — the loop form that the
// reference's own test (tests/algo/test_rnn.py) proves equal to its PackedSequence path (rnn_utils.py:114-158).
// kind 0 = GRU (gates r,z,n), 1 = LSTM (gates i,f,g,o).  All matrices row-major; h_prev/c_prev may be strided rows.
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// gates_out [C, 4H]: GRU stores {r, z, n, hn = (W_hn h + b_hn)}; LSTM stores {i, f, g, o}.
// h_out/c_out: the new state (core output); h_next/c_next = state * keep[c] (keep = 1 - done_or_invalid; NULL = 1).
__global__ __launch_bounds__(256) void k_rnn_cell_fwd(int kind, const float *__restrict__ gx,
                                                      const float *__restrict__ gh, const float *__restrict__ h_prev,
                                                      int64_t ld_h, const float *__restrict__ c_prev, int64_t ld_c,
                                                      const float *__restrict__ keep, int C, int H,
                                                      float *__restrict__ gates_out, float *__restrict__ h_out,
                                                      float *__restrict__ c_out, float *__restrict__ h_next,
                                                      float *__restrict__ c_next) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)C * H) return;
    const int64_t c = i / H;
    const int j = (int)(i - c * H);
    const float k = keep ? keep[c] : 1.0f;
    if (kind == 0) {
        const float hp = h_prev[c * ld_h + j];
        float r, z, hn, xn;
        if (gh) {
            const float *x = gx + c * 3 * H, *g = gh + c * 3 * H;
            r = sigmoidf_(x[j] + g[j]);
            z = sigmoidf_(x[H + j] + g[H + j]);
            hn = g[2 * H + j];
            xn = x[2 * H + j];
        } else {  // gx [C, 4H] = {r_pre, z_pre (x and h parts summed), x W_in^T + b_in, h W_hn^T + b_hn} (sf_linear_fwd_dual)
            const float *x = gx + c * 4 * H;
            r = sigmoidf_(x[j]);
            z = sigmoidf_(x[H + j]);
            xn = x[2 * H + j];
            hn = x[3 * H + j];
        }
        const float n = tanhf(xn + r * hn);
        const float h = (1.0f - z) * n + z * hp;
        if (gates_out) {
            float *go = gates_out + c * 4 * H;
            go[j] = r; go[H + j] = z; go[2 * H + j] = n; go[3 * H + j] = hn;
        }
        h_out[i] = h;
        if (h_next) h_next[i] = h * k;
    } else {
        const float *x = gx + c * 4 * H;
        const float cp = c_prev[c * ld_c + j];
        float ig, fg, gg, og;
        if (gh) {
            const float *g = gh + c * 4 * H;
            ig = sigmoidf_(x[j] + g[j]);
            fg = sigmoidf_(x[H + j] + g[H + j]);
            gg = tanhf(x[2 * H + j] + g[2 * H + j]);
            og = sigmoidf_(x[3 * H + j] + g[3 * H + j]);
        } else {  // gx already holds x W_ih^T + b_ih + h W_hh^T + b_hh (sf_linear_fwd_dual)
            ig = sigmoidf_(x[j]);
            fg = sigmoidf_(x[H + j]);
            gg = tanhf(x[2 * H + j]);
            og = sigmoidf_(x[3 * H + j]);
        }
        const float cn = fg * cp + ig * gg;
        const float h = og * tanhf(cn);
        if (gates_out) {
            float *go = gates_out + c * 4 * H;
            go[j] = ig; go[H + j] = fg; go[2 * H + j] = gg; go[3 * H + j] = og;
        }
        h_out[i] = h;
        c_out[i] = cn;
        if (h_next) h_next[i] = h * k;
        if (c_next) c_next[i] = cn * k;
    }
}

extern "C" int sf_rnn_cell_fwd(int kind, const float *gx, const float *gh, const float *h_prev, int64_t ld_h,
                               const float *c_prev, int64_t ld_c, const float *keep, int C, int H, float *gates_out,
                               float *h_out, float *c_out, float *h_next, float *c_next, void *stream) {
    SF_REQUIRE((kind == 0 || kind == 1) && gx && h_prev && h_out && C > 0 && H > 0, "sf_rnn_cell_fwd: bad args");
    SF_REQUIRE(kind == 0 || (c_prev && c_out), "sf_rnn_cell_fwd: LSTM needs c_prev and c_out");
    const int64_t n = (int64_t)C * H;
    k_rnn_cell_fwd<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, STREAM(stream)>>>(
        kind, gx, gh, h_prev, ld_h, c_prev, ld_c, keep, C, H, gates_out, h_out, c_out, h_next, c_next);
    return sf_launch_status("sf_rnn_cell_fwd");
}

// backward of one step.  dh [C,H] = gradient wrt this step's h_out (output gradient + carry from t+1, already
// masked); dc_in = carry into c_out (LSTM; NULL = 0).  Outputs: dgx [C,G*H] (gradient wrt x W_ih^T + b_ih),
// dgh [C,G*H] (wrt h W_hh^T + b_hh; LSTM: may alias dgx or be NULL since both are equal), dh_direct [C,H] (the part
// of dL/dh_prev that does not go through W_hh; GRU: dh*z, LSTM: 0 -> not written), dc_prev [C,H] (LSTM: dc*f).
__global__ __launch_bounds__(256) void k_rnn_cell_bwd(int kind, const float *__restrict__ dh,
                                                      const float *__restrict__ dc_in,
                                                      const float *__restrict__ gates, const float *__restrict__ h_prev,
                                                      int64_t ld_h, const float *__restrict__ c_prev, int64_t ld_c,
                                                      const float *__restrict__ c_out, int C, int H,
                                                      float *__restrict__ dgx, float *__restrict__ dgh,
                                                      float *__restrict__ dh_direct, float *__restrict__ dc_prev) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)C * H) return;
    const int64_t c = i / H;
    const int j = (int)(i - c * H);
    const float *go = gates + c * 4 * H;
    const float d = dh[i];
    if (kind == 0) {
        const float r = go[j], z = go[H + j], n = go[2 * H + j], hn = go[3 * H + j];
        const float hp = h_prev[c * ld_h + j];
        const float dn_pre = (d * (1.0f - z)) * (1.0f - n * n);
        const float dz_pre = (d * (hp - n)) * (z * (1.0f - z));
        const float dr_pre = (dn_pre * hn) * (r * (1.0f - r));
        float *x = dgx + c * 3 * H, *g = dgh + c * 3 * H;
        x[j] = dr_pre; x[H + j] = dz_pre; x[2 * H + j] = dn_pre;
        g[j] = dr_pre; g[H + j] = dz_pre; g[2 * H + j] = dn_pre * r;
        dh_direct[i] = d * z;
    } else {
        const float ig = go[j], fg = go[H + j], gg = go[2 * H + j], og = go[3 * H + j];
        const float tc = tanhf(c_out[i]);
        const float dc = d * og * (1.0f - tc * tc) + (dc_in ? dc_in[i] : 0.0f);
        float *x = dgx + c * 4 * H;
        const float di = (dc * gg) * (ig * (1.0f - ig)), df = (dc * c_prev[c * ld_c + j]) * (fg * (1.0f - fg));
        const float dg = (dc * ig) * (1.0f - gg * gg), dob = (d * tc) * (og * (1.0f - og));
        x[j] = di; x[H + j] = df; x[2 * H + j] = dg; x[3 * H + j] = dob;
        if (dgh && dgh != dgx) {
            float *g = dgh + c * 4 * H;
            g[j] = di; g[H + j] = df; g[2 * H + j] = dg; g[3 * H + j] = dob;
        }
        dc_prev[i] = dc * fg;
    }
}

extern "C" int sf_rnn_cell_bwd(int kind, const float *dh, const float *dc_in, const float *gates, const float *h_prev,
                               int64_t ld_h, const float *c_prev, int64_t ld_c, const float *c_out, int C, int H,
                               float *dgx, float *dgh, float *dh_direct, float *dc_prev, void *stream) {
    SF_REQUIRE((kind == 0 || kind == 1) && dh && gates && dgx && C > 0 && H > 0, "sf_rnn_cell_bwd: bad args");
    SF_REQUIRE(kind == 1 || (h_prev && dgh && dh_direct), "sf_rnn_cell_bwd: GRU needs h_prev, dgh, dh_direct");
    SF_REQUIRE(kind == 0 || (c_prev && c_out && dc_prev), "sf_rnn_cell_bwd: LSTM needs c_prev, c_out, dc_prev");
    const int64_t n = (int64_t)C * H;
    k_rnn_cell_bwd<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, STREAM(stream)>>>(
        kind, dh, dc_in, gates, h_prev, ld_h, c_prev, ld_c, c_out, C, H, dgx, dgh, dh_direct, dc_prev);
    return sf_launch_status("sf_rnn_cell_bwd");
}

// rollout: state input of step t+1 = new state * (1 - done) (batched_sampling.py:332-335); out row b = [h | c] (c: LSTM)
__global__ __launch_bounds__(256) void k_rnn_store_state(const float *__restrict__ h, const float *__restrict__ c,
                                                         const uint8_t *__restrict__ dones, int64_t done_stride,
                                                         float *__restrict__ out, int64_t out_stride, int64_t B, int H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int W = c ? 2 * H : H;
    if (i >= B * W) return;
    const int64_t b = i / W;
    const int j = (int)(i - b * W);
    const float v = j < H ? h[b * H + j] : c[b * H + j - H];
    out[b * out_stride + j] = dones[b * done_stride] ? 0.0f : v;
}

extern "C" int sf_rnn_store_state(const float *h, const float *c, const uint8_t *dones, int64_t done_stride, float *out,
                                  int64_t out_stride, int64_t B, int H, void *stream) {
    SF_REQUIRE(h && dones && out && B > 0 && H > 0, "sf_rnn_store_state: bad args");
    const int64_t n = B * (c ? 2 * H : H);
    k_rnn_store_state<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, STREAM(stream)>>>(h, c, dones, done_stride, out,
                                                                                             out_stride, B, H);
    return sf_launch_status("sf_rnn_store_state");
}

// training pass of a recurrent model over a minibatch of Cn chunks of R consecutive dataset rows (learner.py:557-569,
// rnn_utils.py:59-105): chunk c starts at dataset row r0 = index ? index[c*R] : offset + c*R.  One launch instead of the
// torch op sequence {arange, gathers, ~, |, transpose, float, index_select}:
//   keep_tm[t][c] = !(dones[r0 + t] | !valids[r0 + t])      (state is zeroed AFTER a done / invalid step)
//   h0[c][:]      = rnn_states[r0][:]                       (the state the chunk's first step was collected with)
__global__ __launch_bounds__(256) void k_rnn_chunk_setup(const uint8_t *__restrict__ dones,
                                                         const uint8_t *__restrict__ valids,
                                                         const float *__restrict__ rnn_states,
                                                         const int32_t *__restrict__ index, int64_t offset, int Cn,
                                                         int R, int S, int traj_T, float *__restrict__ keep_tm,
                                                         float *__restrict__ h0) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t nk = (int64_t)Cn * R;
    if (i < nk) {  // i = t * Cn + c
        const int t = (int)(i / Cn), c = (int)(i - (int64_t)t * Cn);
        const int64_t row = index ? (int64_t)index[(int64_t)c * R + t] : offset + (int64_t)c * R + t;
        keep_tm[i] = (dones[row] || !valids[row]) ? 0.0f : 1.0f;
    } else if (i < nk + (int64_t)Cn * S) {
        const int64_t j = i - nk;
        const int c = (int)(j / S), e = (int)(j - (int64_t)c * S);
        int64_t row = index ? (int64_t)index[(int64_t)c * R] : offset + (int64_t)c * R;
        if (traj_T > 0) row = (row / traj_T) * (traj_T + 1) + row % traj_T;  // dataset row e*T+t -> slab row e*(T+1)+t
        h0[j] = rnn_states[row * S + e];
    }
}

extern "C" int sf_rnn_chunk_setup(const uint8_t *dones, const uint8_t *valids, const float *rnn_states,
                                  const int32_t *index, int64_t offset, int Cn, int R, int S, int traj_T, float *keep_tm,
                                  float *h0, void *stream) {
    SF_REQUIRE(dones && valids && rnn_states && keep_tm && h0 && Cn > 0 && R > 0 && S > 0, "sf_rnn_chunk_setup: bad args");
    const int64_t tot = (int64_t)Cn * R + (int64_t)Cn * S;
    k_rnn_chunk_setup<<<dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, STREAM(stream)>>>(
        dones, valids, rnn_states, index, offset, Cn, R, S, traj_T, keep_tm, h0);
    return sf_launch_status("sf_rnn_chunk_setup");
}

// y[c, :] = (a[c, :] + b[c, :]) * keep[c]    (carry of dL/dh across a step boundary; b or keep may be NULL)
__global__ __launch_bounds__(256) void k_rows_add_scale(const float *__restrict__ a, const float *__restrict__ b,
                                                        const float *__restrict__ keep, int64_t C, int H,
                                                        float *__restrict__ y) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * H) return;
    const float v = a[i] + (b ? b[i] : 0.0f);
    y[i] = keep ? v * keep[i / H] : v;
}

extern "C" int sf_rows_add_scale(const float *a, const float *b, const float *keep, int64_t C, int H, float *y,
                                 void *stream) {
    SF_REQUIRE(a && y && C > 0 && H > 0, "sf_rows_add_scale: bad args");
    k_rows_add_scale<<<dim3((unsigned)((C * H + 255) / 256)), dim3(256), 0, STREAM(stream)>>>(a, b, keep, C, H, y);
    return sf_launch_status("sf_rows_add_scale");
}


// ---- ActionParameterizationContinuousNonAdaptiveStddev with continuous_tanh_scale > 0 (model/action_parameterization.py
// :62-66): action means y = tanh(x / s) * s, in place on `ncols` columns of the heads matrix; backward g *= 1 - (y/s)^2.
__global__ __launch_bounds__(256) void k_tanh_scale(float *__restrict__ x, float *__restrict__ gy, const float *__restrict__ y,
                                                    int ld, int64_t n, int col0, int ncols, float s, float inv_s) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n * ncols; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / ncols;
        const int64_t o = r * ld + col0 + (i - r * ncols);
        if (x) x[o] = tanhf(x[o] * inv_s) * s;
        else { const float t = y[o] * inv_s; gy[o] *= 1.f - t * t; }
    }
}
extern "C" int sf_tanh_scale_fwd(float *x, int ld, int64_t n, int col0, int ncols, float scale, void *stream) {
    SF_REQUIRE(x && n > 0 && ncols > 0 && col0 >= 0 && col0 + ncols <= ld && scale > 0.f, "sf_tanh_scale_fwd: bad args");
    const int64_t tot = n * ncols;
    k_tanh_scale<<<dim3((unsigned)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096)), dim3(256), 0, STREAM(stream)>>>(
        x, nullptr, nullptr, ld, n, col0, ncols, scale, 1.f / scale);
    return sf_launch_status("sf_tanh_scale_fwd");
}
extern "C" int sf_tanh_scale_bwd(float *g, const float *y, int ld, int64_t n, int col0, int ncols, float scale,
                                 void *stream) {
    SF_REQUIRE(g && y && n > 0 && ncols > 0 && col0 >= 0 && col0 + ncols <= ld && scale > 0.f, "sf_tanh_scale_bwd: bad args");
    const int64_t tot = n * ncols;
    k_tanh_scale<<<dim3((unsigned)((tot + 255) / 256 < 4096 ? (tot + 255) / 256 : 4096)), dim3(256), 0, STREAM(stream)>>>(
        nullptr, g, y, ld, n, col0, ncols, scale, 1.f / scale);
    return sf_launch_status("sf_tanh_scale_bwd");
}

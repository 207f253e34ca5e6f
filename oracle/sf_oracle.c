/*
 * sf_oracle.c — CPU restatement of the reference's APPO hot-path arithmetic.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under sample_factory_amd/ (the product) may link, import or call this
 * file.  It exists so that tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg have an independent
 * checker for the HIP kernels that travels to the GPU box (where /root/reference does not exist).
 *
 * Every function cites the reference (Sample Factory v2.1.3, paths relative to sample_factory/) it restates.
 * The restatement is pinned against the reference ITSELF: oracle/gen_golden.py imports the reference in the build
 * container, runs its real functions (gae_advantages, Learner._prepare_batch, Learner._calculate_losses,
 * Learner.train, RunningMeanStdInPlace, CategoricalActionDistribution, ...) on seeded inputs and commits the
 * input/output vectors to tests/golden/; tests/test_oracle_golden.py checks this file against those vectors.
 *
 * Compile: gcc -O2 -ffp-contract=off -shared -fPIC   (contraction off: the reference's torch CPU kernels round
 * after every elementwise op; we keep the same op order in fp32.)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define SFO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------------------------
 * GAE — algo/utils/rl_utils.py:78-94 (gae_advantages) and :52-73 (calculate_discounted_sum_torch).
 * Boundary layout is env-major: rewards/dones [E,T], values/valids [E,T+1] (learner.py:992-999).
 * deltas = (r - v[t]) * valid[t] + (1 - done[t]) * (gamma * v[t+1] * valid[t+1])
 * cum    = delta + (gamma*lambda * valid[t] + (1 - valid[t])) * cum * (1 - done[t])      (reverse over t)
 * ---------------------------------------------------------------------------------------------------------- */
SFO_API void sfo_gae(const float *rewards, const uint8_t *dones, const float *values, const uint8_t *valids,
                     int E, int T, double gamma, double lambda, float *adv) {
    const float g = (float)gamma;
    const float gl = (float)(gamma * lambda); /* python computes γ*λ in double, torch casts the scalar to f32 */
    for (int e = 0; e < E; ++e) {
        const float *r = rewards + (size_t)e * T;
        const uint8_t *d = dones + (size_t)e * T;
        const float *v = values + (size_t)e * (T + 1);
        const uint8_t *va = valids + (size_t)e * (T + 1);
        float cum = 0.0f;
        for (int t = T - 1; t >= 0; --t) {
            const float valid_t = va[t] ? 1.0f : 0.0f, valid_n = va[t + 1] ? 1.0f : 0.0f;
            const float done = d[t] ? 1.0f : 0.0f;
            const float a = (r[t] - v[t]) * valid_t;
            const float b = (1.0f - done) * ((g * v[t + 1]) * valid_n);
            const float delta = a + b;
            const float disc = gl * valid_t + (1.0f - valid_t);
            cum = delta + (disc * cum) * (1.0f - done);
            adv[(size_t)e * T + t] = cum;
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------
 * RunningMeanStdInPlace — algo/utils/running_mean_std.py:22-110.  Scalar-shaped statistics (input_shape (1,),
 * as used by returns_normalizer, model/actor_critic.py:33-37).  stats = {mean, var, count} in f64.
 * ---------------------------------------------------------------------------------------------------------- */
SFO_API void sfo_rms_update(double *stats, const float *x, long n) {
    /* x.mean(), x.var() (unbiased) are fp32 tensors in the reference; we accumulate in f64 and round to f32. */
    double s = 0.0;
    for (long i = 0; i < n; ++i) s += x[i];
    const double m = s / (double)n;
    double ss = 0.0;
    for (long i = 0; i < n; ++i) { const double dlt = x[i] - m; ss += dlt * dlt; }
    const double batch_mean = (double)(float)m;
    const double batch_var = (double)(float)(n > 1 ? ss / (double)(n - 1) : NAN);
    /* _update_mean_var_count_from_moments, running_mean_std.py:51-62 */
    const double mean = stats[0], var = stats[1], count = stats[2];
    const double delta = batch_mean - mean;
    const double tot = count + (double)n;
    const double new_mean = mean + delta * (double)n / tot;
    const double m_a = var * count, m_b = batch_var * (double)n;
    const double M2 = m_a + m_b + (delta * delta) * count * (double)n / tot;
    stats[0] = new_mean; stats[1] = M2 / tot; stats[2] = tot;
}

/* normalize / denormalize in place — running_mean_std.py:96-110 (eps 1e-5, clip 5) */
SFO_API void sfo_rms_apply(const double *stats, float *x, long n, int denormalize) {
    const float mu = (float)stats[0];
    const float sigma = sqrtf((float)stats[1] + 1e-5f);
    const float clip = 5.0f;
    if (denormalize) {
        for (long i = 0; i < n; ++i) {
            float y = x[i]; y = y < -clip ? -clip : (y > clip ? clip : y);
            x[i] = y * sigma + mu;
        }
    } else {
        const float inv = 1.0f / sigma;
        for (long i = 0; i < n; ++i) {
            float y = (x[i] - mu) * inv;
            x[i] = y < -clip ? -clip : (y > clip ? clip : y);
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------
 * Learner._prepare_batch, scalar part — algo/learning/learner.py:943-1034.
 * In:  rewards[E,T] (mutated: value bootstrap :990), dones, time_outs [E,T] u8, values[E,T+1] (column T already
 *      holds the bootstrap value written at :967), policy_id[E,T] i32, policy_version[E,T] f32,
 *      actions[E*T*num_actions], log_prob_actions[E*T] (sanitised at :1029-1032),
 *      rms[3] = returns normaliser stats (updated when normalize_returns).
 * Out: valids[E,T+1] u8, advantages[E,T], returns[E,T] (normalised if normalize_returns); returns #invalid.
 * ---------------------------------------------------------------------------------------------------------- */
SFO_API long sfo_prepare_batch(float *rewards, const uint8_t *dones, const uint8_t *time_outs, const float *values,
                               const int32_t *policy_id, const float *policy_version, float *actions, int num_actions,
                               float *log_prob_actions, int E, int T, int my_policy_id, int train_step,
                               int max_policy_lag, int normalize_returns, int value_bootstrap, double gamma,
                               double lambda, double *rms, uint8_t *valids, float *adv, float *returns) {
    const size_t N = (size_t)E * T;
    for (int e = 0; e < E; ++e) {
        for (int t = 0; t < T; ++t) {
            const size_t i = (size_t)e * T + t;
            const int same = policy_id[i] == my_policy_id;
            const int fresh = ((float)train_step - policy_version[i]) < (float)max_policy_lag;
            valids[(size_t)e * (T + 1) + t] = (uint8_t)(same && fresh);
        }
        valids[(size_t)e * (T + 1) + T] = valids[(size_t)e * (T + 1) + T - 1];
    }
    float *dv = (float *)malloc(sizeof(float) * (size_t)E * (T + 1));
    memcpy(dv, values, sizeof(float) * (size_t)E * (T + 1));
    if (normalize_returns) sfo_rms_apply(rms, dv, (long)E * (T + 1), 1);
    if (value_bootstrap) {
        const float g = (float)gamma;
        for (int e = 0; e < E; ++e)
            for (int t = 0; t < T; ++t) {
                const size_t i = (size_t)e * T + t;
                const float to = time_outs[i] ? 1.0f : 0.0f, dn = dones[i] ? 1.0f : 0.0f;
                rewards[i] = rewards[i] + ((g * dv[(size_t)e * (T + 1) + t]) * to) * dn;
            }
    }
    sfo_gae(rewards, dones, dv, valids, E, T, gamma, lambda, adv);
    for (int e = 0; e < E; ++e)
        for (int t = 0; t < T; ++t) {
            const size_t i = (size_t)e * T + t;
            const float vf = valids[(size_t)e * (T + 1) + t] ? 1.0f : 0.0f;
            returns[i] = adv[i] + vf * dv[(size_t)e * (T + 1) + t];
        }
    free(dv);
    if (normalize_returns) {
        sfo_rms_update(rms, returns, (long)N);
        sfo_rms_apply(rms, returns, (long)N, 0);
    }
    long num_invalid = 0;
    for (int e = 0; e < E; ++e)
        for (int t = 0; t < T; ++t) {
            const size_t i = (size_t)e * T + t;
            if (!valids[(size_t)e * (T + 1) + t]) {
                ++num_invalid;
                for (int a = 0; a < num_actions; ++a) actions[i * num_actions + a] = 0.0f;
                log_prob_actions[i] = -1.0f;
            }
        }
    return num_invalid;
}

/* ------------------------------------------------------------------------------------------------------------
 * V-trace — learner.py:601-640.  Flat minibatch [N], trajectories of length `recurrence` at stride 1 inside the
 * flat index (index = traj*recurrence + i).  ratio is the clamped ratio (:594), values the CURRENT critic output.
 * ---------------------------------------------------------------------------------------------------------- */
SFO_API void sfo_vtrace(const float *ratio, const float *values, const float *rewards, const float *dones, long N,
                        int recurrence, double gamma, double rho_hat, double c_hat, float *vs, float *adv) {
    const float g = (float)gamma, rh = (float)rho_hat, ch = (float)c_hat;
    const long ntraj = N / recurrence;
    for (long j = 0; j < ntraj; ++j) {
        const long base = j * recurrence;
        float next_values = (values[base + recurrence - 1] - rewards[base + recurrence - 1]) / g;
        float next_vs = next_values;
        for (int i = recurrence - 1; i >= 0; --i) {
            const long k = base + i;
            const float rho = ratio[k] < rh ? ratio[k] : rh;
            const float c = ratio[k] < ch ? ratio[k] : ch;
            const float not_done = 1.0f - dones[k];
            const float ndg = not_done * g;
            const float cv = values[k];
            const float delta_s = rho * ((rewards[k] + ndg * next_values) - cv);
            adv[k] = rho * ((rewards[k] + ndg * next_vs) - cv);
            next_vs = (cv + delta_s) + (ndg * c) * (next_vs - next_values);
            vs[k] = next_vs;
            next_values = cv;
        }
    }
}

/* ------------------------------------------------------------------------------------------------------------
 * PPO loss head, forward + analytic backward — learner.py:586-669 (_calculate_losses tail), :431-486
 * (_policy_loss/_value_loss/_kl_loss/_entropy_exploration_loss/_symmetric_kl_exploration_loss),
 * action_distributions.py:99-194 (Categorical) and :290-323 (Continuous = Independent(Normal)).
 *
 * action_kind 0: Discrete(A): params = logits[N,A], actions[N] (f32 holding the index)
 * action_kind 1: Box(D), A = 2*D: params = [means | log_std], actions[N,D]
 * exploration_kind 0: none (coeff==0), 1: entropy, 2: symmetric_kl (categorical only)
 * `adv` is the UN-normalised advantage; normalisation (per minibatch, unbiased std over valids, floor 1e-7,
 * learner.py:646-647) happens here.  valids u8; all means over valid samples (masked_select then mean; identity
 * when there are no invalids, torch_utils.py:50-55).
 * out_scalars: [0]=policy_loss [1]=exploration_loss [2]=kl_loss [3]=value_loss [4]=kl_old mean [5]=kl_old max
 *              [6]=adv_mean [7]=adv_std [8]=n_valid [9]=entropy mean
 * Gradients of (policy+exploration+kl+value) wrt params [N,A] and values [N] (autograd semantics of torch.min /
 * torch.max ties and clamp boundaries).
 * ---------------------------------------------------------------------------------------------------------- */
static inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }

SFO_API void sfo_ppo_loss(const float *params, const float *values, const float *actions, const float *old_logp,
                          const float *old_params, const float *old_values, const float *adv_in,
                          const float *targets, const uint8_t *valids, long N, int A, int action_kind,
                          double clip_ratio, double clip_value, double value_loss_coeff, double exploration_coeff,
                          int exploration_kind, double kl_coeff, const double *ext_moments, float *out_scalars,
                          float *g_params, float *g_values, const int *head_n, int num_heads) {
    /* head_n/num_heads: Tuple space (action_distributions.py:197-287) = independent heads whose log-prob / entropy / KL /
     * symmetric-KL add up.  head_n[h] > 0: Discrete(head_n[h]) — head_n[h] logits, one action column; head_n[h] < 0:
     * Box(D = -head_n[h]) — 2 D parameters [means | log_std], D action columns (TupleActionDistribution builds every member
     * with get_action_distribution, :222-225, so any mix is legal there).  actions holds sum(columns) floats per sample.
     * NULL / <= 1: one Discrete(A). */
    int one_head[1];
    one_head[0] = A;
    if (!head_n || num_heads <= 1) { head_n = one_head; num_heads = 1; }
    const int H = num_heads;
    int NA = 0;
    for (int hd = 0; hd < H; ++hd) NA += head_n[hd] > 0 ? 1 : -head_n[hd];
    float ent_h[8], kl_h[8], klpu_h[8];
    int acts[8];
    const float clip_hi = (float)(1.0 + clip_ratio);
    const float clip_lo = (float)(1.0 / (1.0 + clip_ratio));
    const float cv = (float)clip_value;
    /* adv normalisation: torch.std_mean over valid entries */
    double s = 0.0; long n = 0;
    for (long i = 0; i < N; ++i) if (valids[i]) { s += adv_in[i]; ++n; }
    double mean = n ? s / (double)n : NAN;
    double ss = 0.0;
    for (long i = 0; i < N; ++i) if (valids[i]) { const double d = adv_in[i] - mean; ss += d * d; }
    double var = n > 1 ? ss / (double)(n - 1) : NAN;
    if (ext_moments) { /* data-parallel shard: GLOBAL {sum, sumsq, n} supplied by the caller */
        n = (long)ext_moments[2];
        mean = ext_moments[0] / ext_moments[2];
        var = (ext_moments[1] - ext_moments[0] * mean) / (ext_moments[2] - 1.0);
    }
    const float adv_mean = (float)mean;
    const float adv_std = (float)sqrt(var);
    const float denom = adv_std < 1e-7f ? 1e-7f : adv_std;
    const float inv_n = 1.0f / (float)n;
    double sum_pl = 0, sum_ent = 0, sum_kl = 0, sum_vl = 0, sum_symkl = 0; float max_kl = -INFINITY;
    const int D = A / 2;
    float *p = (float *)malloc(sizeof(float) * A), *lp = (float *)malloc(sizeof(float) * A),
          *q = (float *)malloc(sizeof(float) * A);
    /* two passes: first the means (symmetric-KL needs the mean for its clamp/isfinite gate), then gradients */
    for (int pass = 0; pass < 2; ++pass) {
        float symkl_gate = 1.0f;
        if (pass == 1 && exploration_kind == 2) {
            const float m = (float)(sum_symkl / (double)n);
            symkl_gate = (isfinite(m) && m <= 30.0f) ? 1.0f : 0.0f; /* clamp(max=30) / zeros() kill the grad */
        }
        for (long i = 0; i < N; ++i) {
            const float *z = params + i * A;
            const int valid = valids[i] != 0;
            float logp_a, ent = 0.f, kl = 0.f, symkl = 0.f;
            float *gz = g_params + i * A;
            if (pass == 1) for (int k = 0; k < A; ++k) gz[k] = 0.f;
            if (action_kind == 0) {
                const float *zo = old_params + i * A;
                logp_a = 0.f;
                int off = 0, aoff = 0;
                for (int hd = 0; hd < H; ++hd) {
                    const int nh = head_n[hd];
                    const float *zh = z + off, *zoh = zo + off;
                    if (nh < 0) { /* Box member: the Continuous formulas of the action_kind 1 branch below */
                        const int Dh = -nh;
                        float e = 0.f, kk = 0.f;
                        for (int k = 0; k < Dh; ++k) {
                            const float mu = zh[k], sd = clampf(expf(zh[Dh + k]), 1e-4f, 1e4f);
                            const float a = actions[i * NA + aoff + k];
                            logp_a += -((a - mu) * (a - mu)) / (2.f * (sd * sd)) - logf(sd) - 0.91893853320467274178f;
                            e += 0.5f + 0.91893853320467274178f + logf(sd);
                            const float muo = zoh[k], sdo = clampf(expf(zoh[Dh + k]), 1e-4f, 1e4f);
                            const float vr = (sd / sdo) * (sd / sdo);
                            const float t1 = ((mu - muo) / sdo) * ((mu - muo) / sdo);
                            kk += 0.5f * (vr + t1 - 1.f - logf(vr));
                        }
                        ent_h[hd] = e; kl_h[hd] = kk; klpu_h[hd] = 0.f; ent += e; kl += kk;
                        off += 2 * Dh; aoff += Dh;
                        continue;
                    }
                    float mx = zh[0]; for (int k = 1; k < nh; ++k) mx = zh[k] > mx ? zh[k] : mx;
                    float se = 0.f; for (int k = 0; k < nh; ++k) se += expf(zh[k] - mx);
                    const float lse = logf(se);
                    for (int k = 0; k < nh; ++k) { lp[off + k] = (zh[k] - mx) - lse; p[off + k] = expf(lp[off + k]); }
                    float mxo = zoh[0]; for (int k = 1; k < nh; ++k) mxo = zoh[k] > mxo ? zoh[k] : mxo;
                    float seo = 0.f; for (int k = 0; k < nh; ++k) seo += expf(zoh[k] - mxo);
                    const float lseo = logf(seo);
                    for (int k = 0; k < nh; ++k) q[off + k] = (zoh[k] - mxo) - lseo;
                    acts[hd] = (int)actions[i * NA + aoff];
                    logp_a += lp[off + acts[hd]];
                    float e = 0.f, kk = 0.f;
                    for (int k = 0; k < nh; ++k) { e -= p[off + k] * lp[off + k]; kk += p[off + k] * (lp[off + k] - q[off + k]); }
                    ent_h[hd] = e; kl_h[hd] = kk; ent += e; kl += kk;
                    const float u = 1.0f / (float)nh, lu = logf(u);
                    float a1 = 0.f, a2 = 0.f;
                    for (int k = 0; k < nh; ++k) { a1 += p[off + k] * (lp[off + k] - lu); a2 += u * (lu - lp[off + k]); }
                    klpu_h[hd] = a1;
                    if (exploration_kind == 2) symkl += 0.5f * (a1 + a2);
                    off += nh; aoff += 1;
                }
            } else {
                /* Normal(mu, clamp(exp(log_std), 1e-4, 1e4)); Independent sums over D */
                const float *zo = old_params + i * A;
                logp_a = 0.f;
                for (int k = 0; k < D; ++k) {
                    const float mu = z[k], ls = z[D + k];
                    const float sd = clampf(expf(ls), 1e-4f, 1e4f);
                    const float a = actions[i * D + k];
                    const float var = sd * sd;
                    logp_a += -((a - mu) * (a - mu)) / (2.f * var) - logf(sd) - 0.91893853320467274178f;
                    ent += 0.5f + 0.91893853320467274178f + logf(sd);
                    const float muo = zo[k], sdo = clampf(expf(zo[D + k]), 1e-4f, 1e4f);
                    /* torch.distributions.kl._kl_normal_normal(p=new, q=old) */
                    const float vr = (sd / sdo) * (sd / sdo);
                    const float t1 = ((mu - muo) / sdo) * ((mu - muo) / sdo);
                    kl += 0.5f * (vr + t1 - 1.f - logf(vr));
                }
            }
            const float raw_ratio = expf(logp_a - old_logp[i]);
            const float ratio = clampf(raw_ratio, 0.05f, 20.0f);
            const float advn = (adv_in[i] - adv_mean) / denom;
            const float clipped = clampf(ratio, clip_lo, clip_hi);
            const float lu_ = ratio * advn, lc_ = clipped * advn;
            const float pl = lu_ < lc_ ? lu_ : lc_;
            const float v = values[i], vo = old_values[i], R = targets[i];
            const float vclip = vo + clampf(v - vo, -cv, cv);
            const float l1 = (v - R) * (v - R), l2 = (vclip - R) * (vclip - R);
            const float vl = l1 > l2 ? l1 : l2;
            if (pass == 0) {
                if (valid) {
                    sum_pl += pl; sum_ent += ent; sum_kl += kl; sum_vl += vl; sum_symkl += symkl;
                    if (kl > max_kl) max_kl = kl;
                }
                continue;
            }
            if (!valid) { g_values[i] = 0.f; continue; }
            /* ---- backward ---- */
            /* policy: L = -(1/n) min(r A, clip(r) A) */
            float dpl_dr;
            const int in_clip = ratio >= clip_lo && ratio <= clip_hi;
            if (lu_ < lc_) dpl_dr = advn;
            else if (lu_ > lc_) dpl_dr = in_clip ? advn : 0.f;
            else dpl_dr = 0.5f * advn + (in_clip ? 0.5f * advn : 0.f);
            const int in_hard = raw_ratio >= 0.05f && raw_ratio <= 20.0f;
            const float dL_dlogp = in_hard ? (-inv_n) * dpl_dr * raw_ratio : 0.f;
            if (action_kind == 0) {
                int off = 0, aoff = 0;
                for (int hd = 0; hd < H; ++hd) {
                    const int nh = head_n[hd];
                    if (nh < 0) { /* Box member */
                        const int Dh = -nh;
                        const float *zh = z + off, *zoh = old_params + i * A + off;
                        for (int k = 0; k < Dh; ++k) {
                            const float mu = zh[k], e = expf(zh[Dh + k]);
                            const float sd = clampf(e, 1e-4f, 1e4f);
                            const float dsd_dls = (e >= 1e-4f && e <= 1e4f) ? e : 0.f;
                            const float a = actions[i * NA + aoff + k];
                            const float var = sd * sd;
                            float gmu = dL_dlogp * ((a - mu) / var);
                            float gsd = dL_dlogp * (((a - mu) * (a - mu)) / (var * sd) - 1.f / sd);
                            if (exploration_kind == 1) gsd += -(float)exploration_coeff * inv_n * (1.f / sd);
                            if (kl_coeff != 0.0) {
                                const float muo = zoh[k], sdo = clampf(expf(zoh[Dh + k]), 1e-4f, 1e4f);
                                gmu += (float)kl_coeff * inv_n * ((mu - muo) / (sdo * sdo));
                                gsd += (float)kl_coeff * inv_n * (sd / (sdo * sdo) - 1.f / sd);
                            }
                            gz[off + k] = gmu;
                            gz[off + Dh + k] = gsd * dsd_dls;
                        }
                        off += 2 * Dh; aoff += Dh;
                        continue;
                    }
                    aoff += 1;
                    const float u = 1.0f / (float)nh, lu = logf(u);
                    for (int k = off; k < off + nh; ++k) {
                        float gk = dL_dlogp * ((k - off == acts[hd] ? 1.f : 0.f) - p[k]);
                        if (exploration_kind == 1) gk += (float)exploration_coeff * inv_n * (p[k] * (lp[k] + ent_h[hd]));
                        if (exploration_kind == 2)
                            gk += symkl_gate * (float)exploration_coeff * inv_n * 0.5f *
                                  (p[k] * ((lp[k] - lu) - klpu_h[hd]) + p[k] - u);
                        if (kl_coeff != 0.0) gk += (float)kl_coeff * inv_n * (p[k] * ((lp[k] - q[k]) - kl_h[hd]));
                        gz[k] = gk;
                    }
                    off += nh;
                }
            } else {
                const float *zo = old_params + i * A;
                for (int k = 0; k < D; ++k) {
                    const float mu = z[k], ls = z[D + k];
                    const float e = expf(ls);
                    const float sd = clampf(e, 1e-4f, 1e4f);
                    const float dsd_dls = (e >= 1e-4f && e <= 1e4f) ? e : 0.f;
                    const float a = actions[i * D + k];
                    const float var = sd * sd;
                    /* d logp / d mu, d logp / d sd */
                    float gmu = dL_dlogp * ((a - mu) / var);
                    float gsd = dL_dlogp * (((a - mu) * (a - mu)) / (var * sd) - 1.f / sd);
                    if (exploration_kind == 1) gsd += -(float)exploration_coeff * inv_n * (1.f / sd);
                    if (kl_coeff != 0.0) {
                        const float muo = zo[k], sdo = clampf(expf(zo[D + k]), 1e-4f, 1e4f);
                        gmu += (float)kl_coeff * inv_n * ((mu - muo) / (sdo * sdo));
                        gsd += (float)kl_coeff * inv_n * (sd / (sdo * sdo) - 1.f / sd);
                    }
                    gz[k] = gmu;
                    gz[D + k] = gsd * dsd_dls;
                }
            }
            /* value: L = (c/n) max(l1, l2) */
            const int in_v = (v - vo) >= -cv && (v - vo) <= cv;
            float dvl;
            if (l1 > l2) dvl = 2.f * (v - R);
            else if (l2 > l1) dvl = in_v ? 2.f * (vclip - R) : 0.f;
            else dvl = (v - R) + (in_v ? (vclip - R) : 0.f);
            g_values[i] = (float)value_loss_coeff * inv_n * dvl;
        }
    }
    free(p); free(lp); free(q);
    out_scalars[0] = (float)(-(sum_pl / (double)n));
    if (exploration_kind == 1) out_scalars[1] = (float)(-exploration_coeff * (sum_ent / (double)n));
    else if (exploration_kind == 2) {
        float m = (float)(sum_symkl / (double)n);
        if (!isfinite(m)) m = 0.f;
        if (m > 30.f) m = 30.f;
        out_scalars[1] = (float)exploration_coeff * m;
    } else out_scalars[1] = 0.f;
    out_scalars[2] = (float)(kl_coeff * (sum_kl / (double)n));
    out_scalars[3] = (float)(value_loss_coeff * (sum_vl / (double)n));
    out_scalars[4] = (float)(sum_kl / (double)n);
    out_scalars[5] = max_kl;
    out_scalars[6] = adv_mean;
    out_scalars[7] = adv_std;
    out_scalars[8] = (float)n;
    out_scalars[9] = (float)(sum_ent / (double)n);
}

/* ------------------------------------------------------------------------------------------------------------
 * Global-norm gradient clip — torch.nn.utils.clip_grad_norm_ as called at learner.py:782-784:
 * total_norm = ||g||_2; coef = min(1, max_norm / (total_norm + 1e-6)); g *= coef.  Returns total_norm.
 * ---------------------------------------------------------------------------------------------------------- */
SFO_API float sfo_clip_grad_norm(float *g, long P, double max_norm) {
    double ss = 0.0;
    for (long i = 0; i < P; ++i) ss += (double)g[i] * (double)g[i];
    const float total = (float)sqrt(ss);
    if (max_norm > 0.0) {
        float coef = (float)max_norm / (total + 1e-6f);
        if (coef > 1.0f) coef = 1.0f;
        for (long i = 0; i < P; ++i) g[i] = g[i] * coef;
    }
    return total;
}

/* ------------------------------------------------------------------------------------------------------------
 * torch.optim.Adam as configured at learner.py:228-243 (no weight decay, no amsgrad, eps = cfg.adam_eps):
 *   m = m + (g - m) * (1 - b1);  v = v*b2 + (1-b2)*g*g
 *   step_size = lr / (1 - b1^t);  denom = sqrt(v) / sqrt(1 - b2^t) + eps;  p = p - step_size * m / denom
 * (torch/optim/adam.py _single_tensor_adam; scalars are python doubles cast to f32 at each tensor op.)
 * ---------------------------------------------------------------------------------------------------------- */
SFO_API void sfo_adam_step(float *p, const float *g, float *m, float *v, long P, int step, double lr, double b1,
                           double b2, double eps) {
    const double bc1 = 1.0 - pow(b1, (double)step);
    const double bc2 = 1.0 - pow(b2, (double)step);
    const float step_size = (float)(lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const float w1 = (float)(1.0 - b1), fb2 = (float)b2, w2 = (float)(1.0 - b2), feps = (float)eps;
    for (long i = 0; i < P; ++i) {
        m[i] = m[i] + (g[i] - m[i]) * w1;
        v[i] = v[i] * fb2 + (g[i] * g[i]) * w2;
        const float denom = sqrtf(v[i]) / bc2_sqrt + feps;
        p[i] = p[i] - step_size * (m[i] / denom);
    }
}

/* ------------------------------------------------------------------------------------------------------------
 * Categorical helpers — action_distributions.py:110-148: log_softmax, log_prob(action) via gather, entropy.
 * ---------------------------------------------------------------------------------------------------------- */
SFO_API void sfo_categorical(const float *logits, const float *actions, long N, int A, float *log_probs,
                             float *logp_a, float *entropy) {
    for (long i = 0; i < N; ++i) {
        const float *z = logits + i * A;
        float mx = z[0]; for (int k = 1; k < A; ++k) mx = z[k] > mx ? z[k] : mx;
        float se = 0.f; for (int k = 0; k < A; ++k) se += expf(z[k] - mx);
        const float lse = logf(se);
        float ent = 0.f;
        for (int k = 0; k < A; ++k) {
            const float l = (z[k] - mx) - lse;
            if (log_probs) log_probs[i * A + k] = l;
            ent -= expf(l) * l;
        }
        if (logp_a) logp_a[i] = ((z[(int)actions[i]] - mx) - lse);
        if (entropy) entropy[i] = ent;
    }
}

/* ------------------------------------------------------------------------------------------------------------
 * Synthetic vector env (ours — SURVEY.md §8d "C2 synthetic inputs"; there is no reference file for it, the
 * reference's pattern for device-resident envs is sf_examples/brax/train_brax.py:160-204).  Counter-based
 * Philox4x32-10 so CPU and GPU produce identical bytes.
 *   key  = (seed, env_id);  ctr = (step, block, stream, 0)
 *   stream 0: obs — block b yields 16 bytes = pixels [16b, 16b+16) of the flattened [C,H,W] u8 frame
 *   stream 1, block 0: word0 < 2^32/1024  -> terminated
 *   reward = (action == (step + env_id) % num_actions) ? 1 : 0 ; truncated = false
 * ---------------------------------------------------------------------------------------------------------- */
static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                 uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

SFO_API void sfo_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                        uint32_t *out) { philox4x32_10(c0, c1, c2, c3, k0, k1, out); }

/* obs for envs [env0, env0+n) at `step` into obs[n, obs_bytes] (obs_bytes % 16 == 0) */
SFO_API void sfo_synth_obs(uint8_t *obs, int n, int env0, long obs_bytes, uint32_t seed, uint32_t step) {
    for (int e = 0; e < n; ++e)
        for (long b = 0; b < obs_bytes / 16; ++b) {
            uint32_t w[4];
            philox4x32_10(step, (uint32_t)b, 0u, 0u, seed, (uint32_t)(env0 + e), w);
            memcpy(obs + (size_t)e * obs_bytes + b * 16, w, 16); /* little-endian byte order */
        }
}

/* step the scalars: actions i32 [n] -> rewards f32, terminated u8 */
SFO_API void sfo_synth_step(const int32_t *actions, int n, int env0, int num_actions, uint32_t seed, uint32_t step,
                            float *rewards, uint8_t *terminated) {
    for (int e = 0; e < n; ++e) {
        const uint32_t env = (uint32_t)(env0 + e);
        rewards[e] = (actions[e] == (int32_t)((step + env) % (uint32_t)num_actions)) ? 1.0f : 0.0f;
        uint32_t w[4];
        philox4x32_10(step, 0u, 1u, 0u, seed, env, w);
        terminated[e] = (uint8_t)(w[0] < (1u << 22)); /* 2^32 / 1024 */
    }
}

/* Inverse-CDF categorical sampling from a uniform drawn off the same Philox stream family (stream 2), as done by
 * the HIP sampler (sf_sample_categorical).  The reference uses torch.multinomial (action_distributions.py:136-142)
 * whose RNG stream cannot be reproduced; parity on sampling is distributional, this function pins the
 * HIP sampler bit-for-bit instead. */
/* Lamb, list-params path of algo/utils/optimizers.py:99-135 with the Learner's configuration (weight decay, trust
 * ratio per tensor = per segment id; seg 255 = padding).  The reference's per-parameter `step` starts at 1. */
SFO_API void sfo_lamb_step(float *p, const float *g, float *m, float *v, const uint8_t *seg, long P, int nseg,
                           int step, double lr, double b1, double b2, double eps, double wd, double min_trust) {
    const float fb1 = (float)b1, fb2 = (float)b2, w1 = (float)(1.0 - b1), w2 = (float)(1.0 - b2);
    const float ib1 = (float)(1.0 / (1.0 - pow(b1, (double)step)));
    const float ib2 = (float)(1.0 / sqrt(1.0 - pow(b2, (double)step)));
    float *u = (float *)malloc(sizeof(float) * P);
    double *wn = (double *)calloc(2 * (nseg > 0 ? nseg : 1), sizeof(double));
    for (long i = 0; i < P; ++i) {
        if (seg[i] >= nseg) continue;
        m[i] = m[i] * fb1 + w1 * g[i];
        v[i] = v[i] * fb2 + w2 * (g[i] * g[i]);
        const float mh = m[i] * ib1, vh = sqrtf(v[i]) * ib2;
        float s = mh / (vh + (float)eps);
        if (wd > 0.0) s = s + (float)wd * p[i];
        u[i] = s;
        wn[2 * seg[i]] += (double)p[i] * p[i];
        wn[2 * seg[i] + 1] += (double)s * s;
    }
    for (long i = 0; i < P; ++i) {
        if (seg[i] >= nseg) continue;
        const float a = (float)sqrt(wn[2 * seg[i]]), b = (float)sqrt(wn[2 * seg[i] + 1]);
        float tr = 1.0f;
        if (min_trust != 1.0 && a != 0.f && b != 0.f) {
            tr = (a < 10.0f ? a : 10.0f) / b;
            tr = tr < (float)min_trust ? (float)min_trust : (tr > (float)(1.0 / min_trust) ? (float)(1.0 / min_trust) : tr);
        }
        p[i] = p[i] + (-(float)lr * tr) * u[i];
    }
    free(u); free(wn);
}

SFO_API void sfo_sample_categorical(const float *logits, long N, int A, uint32_t seed, uint32_t step,
                                    uint32_t row0, float *actions, float *logp) {
    for (long i = 0; i < N; ++i) {
        const float *z = logits + i * A;
        float mx = z[0]; for (int k = 1; k < A; ++k) mx = z[k] > mx ? z[k] : mx;
        float se = 0.f; for (int k = 0; k < A; ++k) se += expf(z[k] - mx);
        const float lse = logf(se);
        uint32_t w[4];
        philox4x32_10(step, 0u, 2u, 0u, seed, row0 + (uint32_t)i, w);
        const float u = (float)(w[0] >> 8) * (1.0f / 16777216.0f); /* 24-bit uniform in [0,1) */
        float acc = 0.f; int a = A - 1;
        for (int k = 0; k < A; ++k) {
            acc += expf((z[k] - mx) - lse);
            if (u < acc) { a = k; break; }
        }
        actions[i] = (float)a;
        logp[i] = (z[a] - mx) - lse;
    }
}

/* Discrete(A) with an action mask, as sf_sample_write_step_masked (action_distributions.py:84-96,110-142) */
SFO_API void sfo_sample_masked(const float *logits, const uint8_t *mask, long N, int A, uint32_t seed, uint32_t step,
                               uint32_t row0, float *actions, float *logp) {
    for (long i = 0; i < N; ++i) {
        const float *z = logits + i * A; const uint8_t *mk = mask + i * A;
        float mx = -INFINITY;
        for (int k = 0; k < A; ++k) { const float v = z[k] + (mk[k] ? 0.f : -1e9f); mx = v > mx ? v : mx; }
        float se = 0.f;
        for (int k = 0; k < A; ++k) se += expf((z[k] + (mk[k] ? 0.f : -1e9f)) - mx);
        const float lse = logf(se);
        float psum = 0.f;
        for (int k = 0; k < A; ++k) psum += mk[k] ? expf((z[k] - mx) - lse) : 0.f;
        const int all_zero = psum == 0.f;
        const float inv = 1.0f / (psum + 1e-13f);
        const float tot = all_zero ? (float)A * 1e-6f : psum * inv;
        uint32_t w[4];
        philox4x32_10(step, 0u, 2u, 0u, seed, row0 + (uint32_t)i, w);
        const float u = (float)(w[0] >> 8) * (1.0f / 16777216.0f);
        float acc = 0.f; int a = A - 1;
        for (int k = 0; k < A; ++k) {
            const float p = all_zero ? 1e-6f : (mk[k] ? expf((z[k] - mx) - lse) * inv : 0.f);
            acc += p / tot;
            if (u < acc) { a = k; break; }
        }
        if (!all_zero && !mk[a]) for (int k = A - 1; k >= 0; --k) if (mk[k]) { a = k; break; }
        actions[i] = (float)a;
        logp[i] = ((z[a] + (mk[a] ? 0.f : -1e9f)) - mx) - lse;
    }
}

/* Tuple heads as sampled by sf_sample_write_step_tuple: a Discrete head h draws from Philox counter (step, h, 2, 0); a Box(D)
 * head (head_n[h] = -D, parameters [means | log_std]) draws the normals of its dims 2j, 2j+1 from counter (step, j, 3, h) by
 * Box-Muller as sfo_sample_normal does; actions [N, sum(columns)], logp = sum of the heads' log-probs
 * (TupleActionDistribution._calc_log_probs). */
SFO_API void sfo_sample_tuple(const float *logits, long N, const int *head_n, int H, uint32_t seed, uint32_t step,
                              uint32_t row0, float *actions, float *logp) {
    int A = 0, NA = 0;
    for (int h = 0; h < H; ++h) { A += head_n[h] > 0 ? head_n[h] : -2 * head_n[h]; NA += head_n[h] > 0 ? 1 : -head_n[h]; }
    for (long i = 0; i < N; ++i) {
        const float *z = logits + i * A;
        float lps = 0.f;
        int off = 0, aoff = 0;
        for (int h = 0; h < H; ++h) {
            const int nh = head_n[h];
            if (nh < 0) {
                const int D = -nh;
                for (int k = 0; k < D; ++k) {
                    uint32_t w[4];
                    philox4x32_10(step, (uint32_t)(k >> 1), 3u, (uint32_t)h, seed, row0 + (uint32_t)i, w);
                    const uint32_t w1 = (k & 1) ? w[2] : w[0], w2 = (k & 1) ? w[3] : w[1];
                    const float u1 = ((float)(w1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
                    const float u2 = (float)(w2 >> 8) * (1.0f / 16777216.0f);
                    const float eps = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
                    const float mu = z[off + k];
                    float sd = expf(z[off + D + k]);
                    sd = sd < 1e-4f ? 1e-4f : (sd > 1e4f ? 1e4f : sd);
                    const float a = mu + sd * eps;
                    actions[i * NA + aoff + k] = a;
                    lps += -((a - mu) * (a - mu)) / (2.f * (sd * sd)) - logf(sd) - 0.91893853320467274178f;
                }
                off += 2 * D; aoff += D;
                continue;
            }
            float mx = z[off]; for (int k = 1; k < nh; ++k) mx = z[off + k] > mx ? z[off + k] : mx;
            float se = 0.f; for (int k = 0; k < nh; ++k) se += expf(z[off + k] - mx);
            const float lse = logf(se);
            uint32_t w[4];
            philox4x32_10(step, (uint32_t)h, 2u, 0u, seed, row0 + (uint32_t)i, w);
            const float u = (float)(w[0] >> 8) * (1.0f / 16777216.0f);
            float acc = 0.f; int a = nh - 1;
            for (int k = 0; k < nh; ++k) { acc += expf((z[off + k] - mx) - lse); if (u < acc) { a = k; break; } }
            actions[i * NA + aoff] = (float)a;
            lps += (z[off + a] - mx) - lse;
            off += nh; aoff += 1;
        }
        logp[i] = lps;
    }
}

/* Continuous (Box) action sampling as done by sf_sample_write_step for action_kind 1: a = mu + sd * eps with
 * sd = clamp(exp(log_std), 1e-4, 1e4) (action_distributions.py:290-310) and eps ~ N(0,1) from Box-Muller on two
 * 24-bit Philox uniforms (stream 3, one Philox call yields the normals of dims 2j, 2j+1); log-prob = sum over dims
 * of Normal.log_prob (torch.distributions.Independent).  The reference samples with torch's RNG (distributional
 * parity only); this pins the HIP sampler bit-for-bit up to libm. */
SFO_API void sfo_sample_normal(const float *params, long N, int D, uint32_t seed, uint32_t step, uint32_t row0,
                               float *actions, float *logp) {
    for (long i = 0; i < N; ++i) {
        const float *z = params + i * 2 * D;
        float lp = 0.f;
        for (int k = 0; k < D; ++k) {
            uint32_t w[4];
            philox4x32_10(step, (uint32_t)(k / 2), 3u, 0u, seed, row0 + (uint32_t)i, w);
            const float u1 = ((float)(w[(k & 1) * 2] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            const float u2 = (float)(w[(k & 1) * 2 + 1] >> 8) * (1.0f / 16777216.0f);
            const float eps = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
            const float mu = z[k];
            float sd = expf(z[D + k]);
            sd = sd < 1e-4f ? 1e-4f : (sd > 1e4f ? 1e4f : sd);
            const float a = mu + sd * eps;
            actions[i * D + k] = a;
            lp += -((a - mu) * (a - mu)) / (2.f * (sd * sd)) - logf(sd) - 0.91893853320467274178f;
        }
        logp[i] = lp;
    }
}

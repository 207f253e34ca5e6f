/*
 * sf_hip.h — C ABI of libsf_hip.so: the MI355X (gfx950) APPO hot path behind Sample Factory's plugin surface.
 *
 * The reference (Sample Factory v2.1.3) has NO native boundary: its hot path is Python calling stock PyTorch ops
 * (SURVEY.md §2.3).  This header therefore declares one entry point per reference *op sequence* on the path; each
 * comment cites the reference lines (paths relative to sample_factory/) that the call replaces.  INTEGRATION.md
 * shows the ctypes stub a Sample Factory maintainer would add at each of those call sites.
 *
 * Conventions
 *  - plain C types only; every pointer is a DEVICE pointer (HBM) unless its name starts with `h_`;
 *  - `stream` is a hipStream_t passed as void* (0 = the null stream); calls only enqueue work, they never
 *    synchronise the device, never allocate, never throw;
 *  - return 0 on success, a negative code on failure; sf_last_error() returns the message of the last failure
 *    on the calling thread;
 *  - boundary layout is the reference's: env-major [E, T(+1), ...] trajectory tensors, flat index e*T + t
 *    (learner.py:1009-1012); policy outputs stored as f32 (shared_buffers.py:100-103); bool tensors are 1 byte.
 */
#ifndef SF_HIP_H
#define SF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SF_OK 0
#define SF_ERR_ARG (-1)
#define SF_ERR_LAUNCH (-2)
#define SF_ERR_UNSUPPORTED (-3)

const char *sf_last_error(void);
int sf_abi_version(void);

/* ---- K13: validity mask ---------------------------------------------------------------------------------
 * learner.py:949-955 (valids = policy_id==pid & train_step-policy_version < max_policy_lag; last column copies
 * the previous one) and :1021-1032 (count invalids; actions=0, log_prob_actions=-1 at invalid rows).
 * valids [E,T+1] u8 out; valids_flat (may be NULL): the same mask as the flat [E*T] dataset array the minibatch consumers
 * index (the reference's `valids[:, :-1]` flatten, learner.py:1009-1012, without its copy); num_invalid: device int32
 * (overwritten). actions [E*T*num_actions], logp [E*T] in/out. */
int sf_valid_mask(const int32_t *policy_id, const float *policy_version, uint8_t *valids, uint8_t *valids_flat,
                  float *actions, int num_actions, float *log_prob_actions, int E, int T, int my_policy_id,
                  int train_step, int max_policy_lag, int32_t *num_invalid, void *stream);

/* ---- K10+K11: value de-normalisation, value bootstrap, GAE, returns ---------------------------------------
 * learner.py:969-1003 + rl_utils.py:52-94 (gae_advantages / calculate_discounted_sum_torch).
 * values [E,T+1] (column T = bootstrap value, learner.py:964-967), valids [E,T+1] u8, dones/time_outs [E,T] u8.
 * rms_stats: device double[3] {mean,var,count} of the returns normaliser, or NULL when normalize_returns=False.
 * rewards is updated in place when value_bootstrap != 0 (learner.py:990).  advantages, returns: [E,T] out;
 * returns are NOT yet normalised (that is sf_rms_*).  One wavefront scans 64 envs; tiles are staged through LDS. */
int sf_gae_returns(float *rewards, const uint8_t *dones, const uint8_t *time_outs, const float *values,
                   const uint8_t *valids, const double *rms_stats, int E, int T, float gamma, float gae_lambda,
                   int value_bootstrap, float *advantages, float *returns, void *stream);

/* ---- K12: RunningMeanStdInPlace (scalar statistics) -------------------------------------------------------
 * running_mean_std.py:51-110.  sf_moments accumulates {sum, sumsq, count} (double[3], zeroed by the call) over x[n]
 * restricted to valid entries: element i of the minibatch is dataset row j = index[i] (index != NULL) or offset + i;
 * valids NULL = all rows count, else valids[j]; the value is x[j], or x[i] when dense_x != 0 (x holds the minibatch's own
 * n values, e.g. the V-trace advantages, while valids is still the dataset's array).
 * Between the two calls a data-parallel learner all-reduces the 3 doubles.
 * sf_rms_update merges the batch moments into stats (Chan merge, :51-62, unbiased batch variance) -> stats_out.
 * sf_rms_apply normalises ((x-mean)/sqrt(var+1e-5) clamp +-5) or de-normalises (clamp, *sigma, +mean) in place. */
int sf_moments(const float *x, const uint8_t *valids, const int32_t *index, int64_t offset, int64_t n, int dense_x,
               double *moments, void *stream);
int sf_rms_update(const double *stats_in, const double *moments, double *stats_out, void *stream);
int sf_rms_apply(float *x, int64_t n, const double *stats, int denormalize, void *stream);

/* ---- K2/K8 (normalize_input=True): observation running mean/std ------------------------------------------------
 * utils/normalize.py:24-70 (ObservationNormalizer), running_mean_std.py:22-136 (RunningMeanStdDictInPlace, full-shape
 * statistics).  x' = (float(x) - obs_subtract_mean) * (1/obs_scale).  Sample addressing as in the network kernels
 * (index | offset, traj_T slab mapping, `stride` elements between rows).  D = elements per observation.
 * sf_obsnorm_moments: per-element {sum, sumsq}[D] of x' over n samples (f64, zeroed by the call).
 * sf_obsnorm_update:  Chan merge into mean/var[D] (in place) and count (count_in -> count_out), refreshes the f32
 *                     tables mu[D], rstd[D] = 1/sqrt(var+1e-5) (n == 0: tables only, e.g. after loading a checkpoint).
 * sf_obsnorm_apply:   out f32 [n, D] = clamp((x' - mu) * rstd, +-5); for images (C > 0) written channels-last
 *                     ([n, H*W, C]) for the NHWC conv kernels.  The north-star preset (normalize_input=False) never
 *                     runs these: there the u8 frames are consumed in place by sf_conv_fwd. */
int sf_obsnorm_moments(const void *in, int in_u8, int64_t stride, const int32_t *index, int64_t offset, int traj_T,
                       int64_t n, int D, float sub_mean, float inv_scale, double *sum, double *sumsq, void *stream);
int sf_obsnorm_update(double *mean, double *var, const double *count_in, double *count_out, const double *sum,
                      const double *sumsq, int64_t n, int D, float *mu_tab, float *rstd_tab, void *stream);
int sf_obsnorm_apply(const void *in, int in_u8, int64_t stride, const int32_t *index, int64_t offset, int traj_T,
                     int64_t n, int D, int C, int HW, float sub_mean, float inv_scale, const float *mu,
                     const float *rstd, float *out, void *stream);

/* ---- K3/K15 recurrent core: GRU / LSTM cells -------------------------------------------------------------------
 * model/core.py:19-64 (ModelCoreRNN over torch.nn.GRU / nn.LSTM, one layer); learner.py:557-581 +
 * rnn_utils.py:114-158 (BPTT over recurrence-length chunks; PackedSequence there, a masked time loop here — the
 * equivalence is the reference's own tests/algo/test_rnn.py).  kind 0 = GRU (gate order r,z,n), 1 = LSTM (i,f,g,o),
 * torch's parameter layout.  gx = x W_ih^T + b_ih, gh = h W_hh^T + b_hh come from sf_conv_fwd (1x1) launches; gh == NULL:
 * gx is the output of ONE sf_linear_fwd_dual launch — LSTM: the complete pre-activation gx + gh [C,4H]; GRU (gru_H = H):
 * [C,4H] = {r and z pre-activations with both parts summed, x W_in^T + b_in, h W_hn^T + b_hn}.
 * fwd: gates_out [C,4H] (GRU {r,z,n,W_hn h + b_hn}; LSTM {i,f,g,o}; NULL at inference), h_out/c_out = new state,
 *      h_next/c_next = new state * keep[c] (keep = 1 - done_or_invalid, learner.py:561; NULL = keep everything).
 * bwd: dh = dL/dh_out of this step (output gradient + masked carry), dc_in = carry into c_out (LSTM); writes dgx, dgh
 *      [C,G*H] (LSTM: dgh may be NULL/alias dgx), dh_direct (GRU: the path of dL/dh_prev that bypasses W_hh),
 *      dc_prev (LSTM).  The W_hh path of dL/dh_prev is sf_conv_dgrad(dgh, W_hh). */
int sf_rnn_cell_fwd(int kind, const float *gx, const float *gh, const float *h_prev, int64_t ld_h, const float *c_prev,
                    int64_t ld_c, const float *keep, int C, int H, float *gates_out, float *h_out, float *c_out,
                    float *h_next, float *c_next, void *stream);
int sf_rnn_cell_bwd(int kind, const float *dh, const float *dc_in, const float *gates, const float *h_prev,
                    int64_t ld_h, const float *c_prev, int64_t ld_c, const float *c_out, int C, int H, float *dgx,
                    float *dgh, float *dh_direct, float *dc_prev, void *stream);
/* y[c,:] = (a[c,:] + b[c,:]) * keep[c] — gradient carry across a step boundary (b, keep optional). */
int sf_rows_add_scale(const float *a, const float *b, const float *keep, int64_t C, int H, float *y, void *stream);

/* ---- K17: V-trace -----------------------------------------------------------------------------------------
 * learner.py:601-640 (the reference runs this loop on the CPU).  Flat minibatch of n samples made of
 * n/recurrence trajectories; sample i of the minibatch is dataset row (index ? index[i] : offset+i).
 * params [n,A] (row stride ld_params) / values [n] (stride ld_values): CURRENT policy outputs (minibatch order; the
 * strides let both be columns of the fused heads GEMM output [n, 1+A]); actions/old_logp/rewards/dones: dataset
 * arrays; dones u8.  Outputs vs, adv [n] in minibatch order.  action_kind as in sf_ppo_loss. */
int sf_vtrace(const float *params, int ld_params, const float *values, int ld_values, const float *actions,
              const float *old_logp, const float *rewards, const uint8_t *dones, const int32_t *index, int64_t offset,
              int64_t n, int A, int action_kind, int recurrence, float gamma, float rho_hat, float c_hat, float *vs,
              float *adv, const int32_t *head_n /* host; Tuple members as in sf_loss_cfg.head_n, or NULL */, int num_heads,
              void *stream);

/* ---- K16: PPO loss head, forward + backward ----------------------------------------------------------------
 * learner.py:586-669 (ratio, clamp [0.05,20], per-minibatch advantage normalisation, clipped surrogate, entropy /
 * symmetric-KL exploration loss, KL(new||old), clipped value loss; all means over valid samples) and the autograd
 * backward of their sum wrt the action-distribution parameters and the value.
 * action_kind 0 = Discrete(A) (action_distributions.py:99-194), 1 = Box(A/2) (:290-323).
 * exploration_kind 0 none, 1 entropy, 2 symmetric_kl.
 * moments: device double[3] {sum, sumsq, n_valid} of the UN-normalised advantage over the (global) minibatch,
 * produced by sf_moments (+ all-reduce).  adv/targets: dataset arrays unless dense_adv != 0 (v-trace: minibatch
 * order).  sums: device double[8], zeroed by the call, receives {policy, entropy-or-symkl, kl, value} sums over
 * valid samples, [4] = max KL, [5] = n_valid; sf_loss_scalars turns them into the reference's loss scalars.
 * params/values are read with row strides ld_params/ld_values (elements); g_params [n,A], g_values [n] are written
 * with the SAME strides (so they can be columns of one [n, 1+A] heads-gradient matrix), minibatch order. */
typedef struct {
    float clip_ratio;        /* cfg.ppo_clip_ratio  (clip_high = 1+c, clip_low = 1/(1+c), learner.py:544-546) */
    float clip_value;        /* cfg.ppo_clip_value */
    float value_loss_coeff;  /* cfg.value_loss_coeff */
    float exploration_coeff; /* cfg.exploration_loss_coeff */
    float kl_coeff;          /* cfg.kl_loss_coeff */
    int32_t exploration_kind;
    int32_t action_kind;
    int32_t dense_adv;
    /* Tuple spaces (action_distributions.py:197-287: independent members built by get_action_distribution, :222-225;
     * log-prob, entropy, KL and symmetric-KL are sums over the members): num_heads > 1 and, per member h,
     * head_n[h] = n > 0 for Discrete(n) (n logits, one action column) or head_n[h] = -D < 0 for Box(D) (2 D parameters
     * [means | log_std], D action columns; symmetric-KL exploration is refused for it, as the reference's
     * ContinuousActionDistribution has no symmetric_kl_with_uniform_prior).  Parameters sum to A; `actions` holds the
     * members' columns side by side per sample.  num_heads <= 1: one Discrete(A). */
    int32_t num_heads;
    int32_t head_n[8];
    /* > 0: `old_values` is the trajectory slab's values array [E, old_values_T + 1] read IN PLACE — dataset row
     * e*T + t lives at e*(T+1) + t (the reference drops the last column by `[:, :-1]` + flatten, learner.py:1009-1012:
     * a copy); 0: old_values is the flat [N] array. */
    int32_t old_values_T;
} sf_loss_cfg;

int sf_ppo_loss(const float *params, int ld_params, const float *values, int ld_values, const float *actions,
                const float *old_logp, const float *old_params, const float *old_values, const float *adv,
                const float *targets, const uint8_t *valids, const int32_t *index, int64_t offset, int64_t n, int A,
                const sf_loss_cfg *h_cfg, const double *moments, double *sums, float *g_params, float *g_values,
                float *ratio_out /* [n] clamped pi/pi_old per sample for the summaries (learner.py:886-903), or NULL */,
                void *stream);
/* out[0..3] = policy, exploration, kl, value losses; [4] kl mean; [5] kl max; [6] adv mean; [7] adv std;
 * [8] n_valid; [9] entropy (or symkl) mean — device float[16]. */
int sf_loss_scalars(const double *sums, const double *moments, const sf_loss_cfg *h_cfg, float *out, void *stream);
/* learner.py:843-923 (`_record_summaries`: ~25 torch reductions with an `.item()` each) for one minibatch in ONE pass:
 * rows i < n are dataset rows index[i] | offset + i; ratio [n] (sf_ppo_loss's ratio_out), values (new, stride ld_values),
 * old_values (flat [N], or the slab's [E, old_values_T + 1]), actions [N, num_actions], adv ([n] if dense_adv else [N]),
 * policy_id / policy_version / valids [N], action_logits [N, A]; exp_avg_sq [P] (may be NULL).  out: device double[24]:
 * [0] rows [1] valid [2] same-policy [3] sum value [4] sum |1 - ratio| over valid [5] clipped ratios (valid, outside
 * [1/(1+clip_ratio), 1+clip_ratio]) [6] sum |v - v_old| [7] sum (train_step - policy_version) over same-policy rows;
 * [8]/[9] ratio min / max (valid) [10] max |v - v_old| [11]/[12] action min / max [13]/[14] adv min / max
 * [15] max |old action parameter| [16]/[17] version_diff min / max [18] max exp_avg_sq (minima +inf / maxima -inf when
 * no row qualified). */
int sf_train_summaries(const uint8_t *valids, const float *ratio, const float *values, int ld_values,
                       const float *old_values, int old_values_T, const float *actions, int num_actions, const float *adv,
                       int dense_adv, const int32_t *policy_id, const float *policy_version, const float *action_logits,
                       int A, const int32_t *index, int64_t offset, int64_t n, int my_policy_id, int train_step,
                       float clip_ratio, const float *exp_avg_sq, int64_t P, double *out, void *stream);

/* ---- K14: minibatch index sets ------------------------------------------------------------------------------
 * learner.py:498-526.  Writes experience_size int32 indices: a pseudo-random permutation (stateless 4-round
 * Feistel network with cycle walking, keyed by seed/epoch) of the recurrence-aligned chunk starts, each expanded to
 * `recurrence` consecutive indices; minibatch k is out[k*batch .. (k+1)*batch).  shuffle==0 writes the identity
 * (contiguous slices, the reference default). */
int sf_minibatch_indices(int32_t *out, int64_t experience_size, int recurrence, int shuffle, uint32_t seed,
                         uint32_t epoch, void *stream);
/* learner.py:507-519 with the REFERENCE's permutation: chunk_starts (device int32[experience_size / recurrence]) is the
 * host's seeded np.random.permutation(np.arange(0, experience_size, recurrence)), uploaded as is; out receives the
 * expanded index runs [s, s+1, .., s+recurrence-1] per chunk, i.e. exactly np.concatenate(...) of :512-515. */
int sf_minibatch_expand(const int32_t *chunk_starts, int32_t *out, int64_t experience_size, int recurrence,
                        void *stream);

/* ---- K18/K19: global-norm clip + Adam ------------------------------------------------------------------------
 * learner.py:782-797: torch.nn.utils.clip_grad_norm_(max_grad_norm) then torch.optim.Adam.step (eps=cfg.adam_eps,
 * no weight decay/amsgrad; learner.py:228-243).  All parameters live in one flat fp32 buffer of P elements.
 * sf_grad_sumsq: sumsq (device double[1], zeroed by the call) = sum g^2.  sf_adam_step reads it: coef =
 * min(1, max_norm/(sqrt(sumsq)+1e-6)) (skipped when max_grad_norm <= 0 or sumsq == NULL), g scaled by
 * grad_scale*coef, then the Adam update in torch's op order.  `step` is the 1-based Adam step count.
 * skip_flag (device uint32, may be NULL): when the word is non-zero as the kernel starts, the launch is a no-op —
 * weights and moments stay untouched (the sticky abort word of the fused recurrent passes, see sf_lstm_seq_fwd). */
int sf_grad_sumsq(const float *g, int64_t P, double *sumsq, void *stream);
int sf_adam_step(float *p, const float *g, float *m, float *v, int64_t P, int step, float lr, float beta1,
                 float beta2, float eps, float max_grad_norm, const double *sumsq, float grad_scale,
                 const uint32_t *skip_flag, void *stream);
/* sf_adam_step with the learning rate in DEVICE memory (lr = *lr_dev * lr_scale; lr_scale = the valid-sample fraction of
 * learner.py:788-794), and the per-minibatch KL-adaptive schedule of learner.py:46-85 (lr_schedule=kl_adaptive_minibatch:
 * lr /= 1.5 above 2 x threshold, *= 1.5 below threshold / 2, within [lr_min, lr_max]) as a one-thread launch on the
 * minibatch's mean KL (sf_loss_scalars out[4]): the schedule no longer costs a host read-back per SGD step.
 * lr_out (may be NULL) receives a copy of the new rate (read back once per epoch with the loss scalars). */
int sf_adam_step_dlr(float *p, const float *g, float *m, float *v, int64_t P, int step, const float *lr_dev,
                     float lr_scale, float beta1, float beta2, float eps, float max_grad_norm, const double *sumsq,
                     float grad_scale, const uint32_t *skip_flag, void *stream);
int sf_lr_kl_adaptive(const float *kl, float *lr_dev, float threshold, float lr_min, float lr_max, float *lr_out,
                      void *stream);
/* measurement helper: one wave spins for spin_cycles shader cycles; out[0] = shader cycles, out[1] = ticks of the constant
 * 100 MHz wall clock elapsed meanwhile (device uint64[2]) -> shader clock [GHz] = 0.1 * out[0] / out[1].  bench.py launches
 * it on a side stream during the timed region (roofline.clock_ghz). */
int sf_clock_probe(unsigned long long *out, int spin_cycles, void *stream);

/* Lamb (cfg.optimizer = "lamb"; algo/utils/optimizers.py:14-189 as configured by learner.py:228-243: Adam direction
 * with bias correction + weight_decay * w, then per-TENSOR trust ratio min(||w||, 10)/||step|| clamped to
 * [min_trust, 1/min_trust]; the reference's `step` starts at 1).  seg_id[P] (u8): index of the reference tensor each
 * flat element belongs to, 255 = padding (skipped); scratch[P] f32; seg_sums: device double[128] (zeroed by the call).
 * Gradient clipping and skip_flag as in sf_adam_step. */
int sf_lamb_step(float *p, const float *g, float *m, float *v, float *scratch, const uint8_t *seg_id, double *seg_sums,
                 int64_t P, int num_segments, int step, float lr, float beta1, float beta2, float eps,
                 float weight_decay, float min_trust, float max_grad_norm, const double *sumsq, float grad_scale,
                 const uint32_t *skip_flag, void *stream);

/* ---- K4/K5: action sampling + policy outputs -> trajectory step ------------------------------------------------
 * action_distributions.py:110-148 (softmax, multinomial, log_softmax, gather), actor_critic.py:112-117,
 * inference_worker.py:235-269,330-339 and batched_sampling.py:308-311 (policy outputs copied into traj[:, t]).
 * logits [B,A] (row stride ld_logits), values [B] (stride ld_values): network outputs.  Samples by inverse CDF from a Philox4x32-10 uniform
 * (key = (seed, row0+b), counter = (step, 0, 2, 0)) and writes, for env b, element (b*stride + t) of the
 * trajectory tensors: actions (f32), log_prob_actions, values (row stride T+1), policy_version, action_logits
 * (A floats) and the int32 action for the env.  `deterministic` != 0 takes argmax (enjoy.py:177-182).
 * action_kind 1 = Box(A/2): params are [means | log_std] (action_distributions.py:290-310); the action
 * mu + clamp(exp(log_std),1e-4,1e4)*eps (eps: Box-Muller on Philox stream 3) is written as A/2 floats into
 * traj_actions[b*T+t] (the env reads it from there; env_actions may be NULL); deterministic takes the mean. */
int sf_sample_write_step(const float *logits, int ld_logits, const float *values, int ld_values, int B, int A, int T,
                         int t, uint32_t seed, uint32_t step, uint32_t row0, float policy_version, int deterministic,
                         int action_kind, float *traj_actions,
                         float *traj_logits, float *traj_logp, float *traj_values, float *traj_policy_version,
                         int32_t *env_actions, void *stream);

/* Discrete(A) with obs["action_mask"] (u8 [B, A], row stride ld_mask; inference_worker.py:324-331): sampling and
 * log-prob follow masked_softmax / masked_log_softmax (action_distributions.py:84-96); raw logits are recorded. */
int sf_sample_write_step_masked(const float *logits, int ld_logits, const float *values, int ld_values,
                                const uint8_t *action_mask, int64_t ld_mask, int B, int A, int T, int t, uint32_t seed,
                                uint32_t step, uint32_t row0, float policy_version, int deterministic,
                                float *traj_actions, float *traj_logits, float *traj_logp, float *traj_values,
                                float *traj_policy_version, int32_t *env_actions, void *stream);

/* Tuple variant of sf_sample_write_step (TupleActionDistribution.sample_actions_log_probs,
 * action_distributions.py:241-245): member h (head_n[h] as in sf_loss_cfg, host array of num_heads <= 8 entries) is
 * sampled by inverse CDF from its own Philox uniform (counter (step, h, 2, 0)) if Discrete, as mu + sd * eps with
 * Box-Muller normals from counter (step, dim / 2, 3, h) if Box(D) (deterministic: arg-max / the mean); traj_actions
 * gets the members' columns side by side, traj_logp the SUM of the members' log-probs.  env_actions [B, num_heads]
 * int32 for an all-Discrete tuple; NULL (required) when a member is a Box — the env then reads traj_actions[:, t]. */
int sf_sample_write_step_tuple(const float *logits, int ld_logits, const float *values, int ld_values, int B,
                               int num_heads, const int32_t *head_n, int T, int t, uint32_t seed, uint32_t step,
                               uint32_t row0, float policy_version, int deterministic, float *traj_actions,
                               float *traj_logits, float *traj_logp, float *traj_values, float *traj_policy_version,
                               int32_t *env_actions, void *stream);

/* ---- K1/K6: env outputs -> trajectory step ---------------------------------------------------------------------
 * batched_sampling.py:208-213 (reward*scale, clamp +-clip), :319-335 (rewards/dones/time_outs/policy_id into
 * traj[:, t]) and :215-287 (episode statistics, kept on device: ep_return/ep_len per env, and on `done` the
 * finished episode is added to ep_stats {sum_return, sum_len, count} (device double[3])). */
int sf_traj_write_env_step(const float *rewards, const uint8_t *terminated, const uint8_t *truncated, int B, int T,
                           int t, float reward_scale, float reward_clip, int policy_id, float *traj_rewards,
                           uint8_t *traj_dones, uint8_t *traj_time_outs, int32_t *traj_policy_id, float *ep_return,
                           int32_t *ep_len, double *ep_stats, void *stream);

/* Ant-shaped continuous stand-in env (config 5): f32 observations [B, D], Box actions [B, A] (row stride act_stride
 * floats, e.g. traj.actions[:, t]).  One launch per env step writes the next observation into `state` (the env's own
 * [B, D] copy) AND into obs_out (row stride out_stride floats = slot t+1 of the slab); reset != 0: fresh observations
 * only.  obs' = terminated ? noise : 0.9*obs + 0.1*noise, reward = -mean(a^2) + 0.1*obs[0], terminated ~ Bernoulli(1/256). */
int sf_synth_vec_step(float *state, const float *actions, int64_t act_stride, float *obs_out, int64_t out_stride, int B,
                      int D, int A, int env0, uint32_t seed, uint32_t step, int reset, float *rewards,
                      uint8_t *terminated, void *stream);

/* ---- host-env ingest ---------------------------------------------------------------------
 * batched_sampling.py:62-82 / rl_utils.py:38: observations of a CPU vector env (envpool: one [B, ...] host array per
 * step) go into slot t of the device slab.  Rows of `row_bytes` bytes, `rows` of them, from (pinned) host memory with
 * pitch src_pitch to device memory with pitch dst_pitch (= (T+1) * row_bytes for slab[:, t]): ONE pitched DMA
 * (hipMemcpy2DAsync) on `stream` — no contiguous device staging copy, no kernel.  Returns without synchronising. */
int sf_h2d_rows(void *dst, int64_t dst_pitch, const void *src, int64_t src_pitch, int64_t row_bytes, int64_t rows,
                void *stream);
/* device -> device row copy with pitches (one launch): the column copies of the slab protocol — the next rollout's
 * obs[:, 0] / rnn_states[:, 0] <- this rollout's [:, T] (batched_sampling.py:289-296, TensorDict `copy_` per leaf in the
 * reference), the bootstrap value -> values[:, T] (learner.py:962-969). */
int sf_copy_rows(void *dst, int64_t dst_pitch, const void *src, int64_t src_pitch, int64_t row_bytes, int64_t rows,
                 void *stream);

/* ---- synthetic vector env (SURVEY.md §8d "C2 synthetic inputs") ------------------------------------------------
 * Device-resident stand-in for a GPU env (the reference's pattern: sf_examples/brax/train_brax.py:160-204).
 * sf_synth_obs writes frame `step` of envs [env0, env0+B) as u8 [obs_bytes] each, env b at obs + b*env_stride —
 * i.e. straight into slot t of the trajectory slab (zero-copy K1).  sf_synth_step: reward = (action ==
 * (step+env)%num_actions), terminated ~ Bernoulli(1/1024) from the Philox stream.  Bit-reproducible on CPU. */
int sf_synth_obs(uint8_t *obs, int64_t env_stride, int B, int env0, int64_t obs_bytes, uint32_t seed, uint32_t step,
                 void *stream);
int sf_synth_step(const int32_t *actions, int B, int env0, int num_actions, uint32_t seed, uint32_t step,
                  float *rewards, uint8_t *terminated, void *stream);

/* ---- K2/K3/K8/K9/K15/K18: actor-critic network (fp32 MFMA) -------------------------------------------------------
 * model/encoder.py:90-150 (conv head + MLP), model/actor_critic.py:160-195 (critic_linear + distribution_linear),
 * utils/normalize.py:51-70 + rl_utils.py:36-42 (u8 -> f32, -mean, *1/scale fused into the first layer's loader;
 * the f32 copy of the observations is never materialised).
 *
 * One implicit-GEMM kernel family, out[M,N] = act(gather(in)[M,K] * W[K,N] + bias), for conv fwd, linear fwd,
 * data-gradient and weight-gradient; see DESIGN.md.  Activations are NHWC ([sample, oh, ow, c]); weights are
 * stored K-major [KH*KW*Cin, Cout] (k = (kh*KW + kw)*Cin + c) — conversion to/from the reference's OIHW layout
 * happens in state_dict()/load_state_dict() on the host side. */
typedef struct {
    int32_t Cin, H, W;        /* input  feature map (per sample) */
    int32_t Cout, KH, KW, stride;
    int32_t OH, OW;           /* output feature map */
    int32_t in_u8;            /* 1: input is the raw u8 NCHW observation; loader applies (x - sub_mean) * inv_scale */
    int32_t relu;             /* activation kind: 0 none, 1 ReLU, 2 tanh, 3 ELU(alpha=1) (model/model_utils.py:27-35).
                                 forward: fused in the epilogue; sf_conv_dgrad: the kind that PRODUCED in_act, whose
                                 derivative (through the stored output) is fused into the dgrad epilogue */
    int32_t traj_T;           /* >0: input rows live in a trajectory slab [E, T+1, ...]; logical sample d (flat dataset
                                 index e*T+t, learner.py:1009-1012) is slab row e*(T+1)+t — read in place, no batcher
                                 copy (batcher.py:192-212) and no [:, :-1] reshape copy (learner.py:1005-1012) */
    float sub_mean, inv_scale;
} sf_conv_desc;

/* forward: in = u8 NCHW [n, Cin,H,W] (in_u8) or f32 NHWC [n,H,W,Cin]; sample i of the batch is input row
 * (index ? index[i] : offset+i) * in_sample_stride (elements).  out f32 NHWC [n,OH,OW,Cout].
 * workspace (optional, >= sf_conv_fwd_workspace bytes, 16-byte aligned): lets small launches (e.g. the 3136->512
 * layer at inference batch 4096: 256 tiles for 256 CUs) split the reduction over gridDim.z and finish with a
 * deterministic ordered sum (+bias, ReLU); NULL = never split.
 * Arithmetic: f32 operands, f32 accumulation on the f32 matrix instructions.  One dispatch differs: the Nature-CNN
 * first layer on raw u8 frames (in_u8, 4x84x84, 8x8 stride 4, Cout <= 32, n >= 256, integer sub_mean in [0, 255]) runs
 * sf_conv_fwd / sf_conv_wgrad on the bf16 matrix instruction with operands that are EXACT in bf16 (pixel - mean; the f32
 * weights / output gradients split exactly into three bf16 terms): every product is exact, accumulation is f32, and
 * inv_scale multiplies the accumulated sum (csrc/sf_nn_u8.h; SF_CONV1_BF16=0 selects the f32 kernels). */
int64_t sf_conv_fwd_workspace(int64_t n, const sf_conv_desc *h_desc);
int sf_conv_fwd(const void *in, int64_t in_sample_stride, const int32_t *index, int64_t offset, const float *w,
                const float *bias, float *out, int64_t n, const sf_conv_desc *h_desc, void *workspace,
                int64_t workspace_bytes, void *stream);
/* weight/bias gradient: dw[K,Cout] (+)= gather(in)^T * dout, db[Cout] = sum dout; dout already has the ReLU mask
 * applied (sf_conv_fwd's consumer does it).  Deterministic two-stage split reduction; workspace >=
 * sf_conv_wgrad_workspace(...) bytes. */
int64_t sf_conv_wgrad_workspace(int64_t n, const sf_conv_desc *h_desc);
int sf_conv_wgrad(const void *in, int64_t in_sample_stride, const int32_t *index, int64_t offset, const float *dout,
                  float *dw, float *db, int64_t n, const sf_conv_desc *h_desc, void *workspace, void *stream);
/* data gradient: din[n,H,W,Cin] = conv_transpose(dout, w) * act'(in_act) (in_act = this layer's input activation,
 * i.e. the previous layer's post-activation output, kind = h_desc->relu; NULL = no activation derivative). */
int sf_conv_dgrad(const float *dout, const float *w, const float *in_act, float *din, int64_t n,
                  const sf_conv_desc *h_desc, void *stream);
/* ReLU SIGN-BIT MASKS between a layer's forward and its weight gradient (model/encoder.py:90-119 under autograd keeps
 * the whole activation for ReLU's backward; here 1 bit per element).  sf_conv_fwd_relu_mask = sf_conv_fwd that also
 * records relu_mask[n*OH*OW] (u32 per output pixel, bit c = channel c of that pixel is > 0; Cout = 32);
 * sf_conv_wgrad_relu_mask = sf_conv_wgrad whose `dout` is the gradient wrt the layer's ReLU OUTPUT, NOT yet masked: the
 * kernel applies the recorded bits on the way in (weight AND bias gradient).  The data-gradient launch that produced
 * `dout` is then called with in_act = NULL and never re-reads this layer's activation (1.68 GB per C2 minibatch).
 * sf_conv_relu_mask_supported: 1 for the launches both kernels take (first layer on raw u8 frames on the exact-product
 * bf16 kernels, Cout == 32, ReLU); otherwise use the plain entry points with in_act. */
int sf_conv_relu_mask_supported(int64_t n, const sf_conv_desc *h_desc);
int sf_conv_fwd_relu_mask(const void *in, int64_t in_sample_stride, const int32_t *index, int64_t offset, const float *w,
                          const float *bias, float *out, uint32_t *relu_mask, int64_t n, const sf_conv_desc *h_desc,
                          void *stream);
int sf_conv_wgrad_relu_mask(const void *in, int64_t in_sample_stride, const int32_t *index, int64_t offset,
                            const float *dout, const uint32_t *relu_mask, float *dw, float *db, int64_t n,
                            const sf_conv_desc *h_desc, void *workspace, void *stream);

/* First conv layer on raw u8 frames WITH the observation normaliser applied in the loader — cfg.normalize_input=True on
 * image observations (cfg/cfg.py:337-341: the default; sf_examples/atari/atari_params.py:39).  Replaces, for the launches
 * it takes, the reference's f32 materialisation of every frame (utils/normalize.py:40-70: obs.float() clone, sub_, mul_;
 * running_mean_std.py:79-110: x.sub_(mu).mul_(1/sigma).clamp_(-5, 5)) followed by conv1 (model/encoder.py:90-119):
 * x' = clamp(((float(u8) - sub_mean) * inv_scale - mu[d]) * rstd[d], +-5) is formed where the byte enters LDS, d = the
 * byte's offset inside the NCHW frame; mu / rstd = the f32 tables [Cin*H*W] sf_obsnorm_update maintains (16-byte
 * aligned).  No normalised copy of the frames exists in HBM (28 KB instead of 28 + 2 x 113 KB per
 * frame and pass).  sf_conv_wgrad_norm = the weight / bias gradient against the same normalised input (dout already
 * masked by the activation derivative; workspace >= sf_conv_wgrad_workspace bytes).  sf_conv_norm_supported: 1 for the
 * launches these take (Nature-CNN conv1 geometry 4x84x84 -> 32, 8x8 stride 4, any n); otherwise sf_obsnorm_apply +
 * sf_conv_fwd on the f32 batch.  Sample addressing (index | offset, traj_T) as sf_conv_fwd. */
int sf_conv_norm_supported(int64_t n, const sf_conv_desc *h_desc);
int sf_conv_fwd_norm(const void *in, int64_t in_sample_stride, const int32_t *index, int64_t offset, const float *mu,
                     const float *rstd, const float *w, const float *bias, float *out, int64_t n,
                     const sf_conv_desc *h_desc, void *stream);
int sf_conv_wgrad_norm(const void *in, int64_t in_sample_stride, const int32_t *index, int64_t offset, const float *mu,
                       const float *rstd, const float *dout, float *dw, float *db, int64_t n, const sf_conv_desc *h_desc,
                       void *workspace, void *stream);

/* Inference on vector observations in one launch: [(x - sub_mean) * inv_scale -> (optional) running mean/std
 * normalisation with clamp +-5] -> act(x W1 + b1) -> act(. W2 + b2); utils/normalize.py:24-70 + model/encoder.py:72-87
 * (MlpEncoder with two layers).  x: f32 rows of D values, x_stride floats apart (e.g. slab obs[:, t]); w1 [D, H1], w2
 * [H1, H2] K-major; mu / rstd: [D] tables of the observation normaliser or both NULL; act: 0 none, 1 relu, 2 tanh, 3 elu;
 * out [n, H2].  Same arithmetic (fmaf chain, ascending k) as sf_conv_fwd on the two layers; needs H1, H2 % 8 == 0 and
 * (D*H1 + H1*H2 + 64*(D+H1)) * 4 <= 64 KiB, else returns an error (use the layer kernels). */
int sf_mlp2_fwd(const float *x, int64_t x_stride, int64_t n, int D, float sub_mean, float inv_scale, const float *mu,
                const float *rstd, const float *w1, const float *b1, int H1, const float *w2, const float *b2, int H2,
                int act, float *out, void *stream);

/* rollout: rnn_states[:, t+1] = new_state * (1 - done) (batched_sampling.py:332-335) in one launch; h (and c for an
 * LSTM) are the cell's [B, H] outputs, dones the u8 [B] column traj.dones[:, t] (element stride done_stride), out the
 * [B, H or 2H] view traj.rnn_states[:, t+1] (row stride out_stride floats). */
int sf_rnn_store_state(const float *h, const float *c, const uint8_t *dones, int64_t done_stride, float *out,
                       int64_t out_stride, int64_t B, int H, void *stream);

/* training pass of a recurrent model (learner.py:557-569, rnn_utils.py:59-105): minibatch = Cn chunks of R consecutive
 * dataset rows, chunk c starting at row index[c*R] (index != NULL: a permuted minibatch, every chunk's R rows consecutive
 * in it) or offset + c*R.  keep_tm [R][Cn] f32 = 0 where the step is done or invalid (the state is zeroed after it), else
 * 1; h0 [Cn][S] = rnn_states[chunk start] (dones / valids: u8 per dataset row; rnn_states [rows][S] f32, or — traj_T > 0 —
 * the slab itself, [E][traj_T + 1][S] read in place: dataset row e*T+t is slab row e*(T+1)+t, no compaction copy). */
int sf_rnn_chunk_setup(const uint8_t *dones, const uint8_t *valids, const float *rnn_states, const int32_t *index,
                       int64_t offset, int Cn, int R, int S, int traj_T, float *keep_tm, float *h0, void *stream);

/* ---- fused LSTM sequence passes (config 5: LSTM-512 core, BPTT over recurrence-length chunks) -------------------
 * model/core.py:19-64 + algo/learning/rnn_utils.py:114-158 as ONE persistent launch per pass instead of
 * {recurrent GEMM, cell, carries} x R launches: every work-group keeps its slice of W_hh in LDS for all R steps and
 * exchanges h_t (forward) / gate gradients (backward) with the other work-groups of its row group through L2.
 * Layouts (all f32, row-major, time-major over the R steps of Cn chunks): gx [R][Cn][4H] = x W_ih^T + b_ih;
 * whh [H][4H] (K-major, torch gate order i,f,g,o); keep [R][Cn] = 1 - done_or_invalid; gates [R][Cn][4H] (post-
 * activation); hprev / cprev [R+1][Cn][H] with slot 0 = the chunk-start state on entry, slot t+1 = state_t * keep[t];
 * hout / cout [R][Cn][H] the unmasked new state.  sync: >= 129 device uint32; words 0..127 are the hand-off counters
 * (zeroed by the call), word 128 is the STICKY abort word: set by a pass that gave up waiting for a work-group, never
 * cleared by the library (the caller zeroes it once, e.g. at the start of Learner.train, and hands the same word to
 * sf_adam_step / sf_lamb_step as `skip_flag`, so that the garbage gradients of an aborted pass never reach the weights).  sf_lstm_seq_supported: 1 when (Cn, H) can
 * take this path on the current device (H in {256, 512}: 16 hidden units per work-group with their W_hh slice resident in
 * LDS; grid <= #CUs), else use sf_rnn_cell_fwd/bwd per step.
 * sf_lstm_seq_bwd: dout [R][Cn][H] = dL/d hout; writes dgx [R][Cn][4H] = dL/d(gate pre-activations) (= the gradient of
 * both gx and h W_hh^T + b_hh); the carries of dL/dh and dL/dc live in registers.  Cn <= 8 * 256 rows. */
int sf_lstm_seq_supported(int Cn, int H);
int sf_lstm_seq_fwd(const float *gx, const float *whh, const float *bhh, const float *keep, float *gates, float *hprev,
                    float *hout, float *cprev, float *cout, uint32_t *sync, int R, int Cn, int H, int env_major,
                    void *stream);
int sf_lstm_seq_bwd(const float *dout, const float *gates, const float *cprev, const float *cout, const float *keep,
                    const float *whh, float *dgx, uint32_t *sync, int R, int Cn, int H, int env_major, void *stream);
/* env_major (all four passes): hout / dout are [Cn][R][H] — the row order of the minibatch (chunk c, step t = row c*R + t
 * of the core's output and of its gradient) — instead of time-major [R][Cn][H]: the caller needs no transpose copies
 * around the pass.  Everything else stays time-major. */

/* the same two passes for a GRU-512 core (the reference's DEFAULT: cfg rnn_type=gru, rnn_size=512; model/core.py:19-64).
 * gx [R][Cn][3H], whh [H][3H], bhh [3H] (torch gate order r,z,n); gates [R][Cn][4H] = {r, z, n, hn} with
 * hn = h W_hn^T + b_hn (the recurrent part of the candidate gate, needed by the backward pass); hprev / hout as above.
 * Backward: hprev = the forward pass's buffer; writes dgx [R][Cn][3H] = {dr, dz, dn} (gradient of gx: W_ih, encoder)
 * and dgh [R][Cn][3H] = {dr, dz, dn * r} (gradient of h W_hh^T + b_hh: W_hh / b_hh, and the hand-off payload of the
 * pass); the direct path dL/dh_prev += dh * z and the carry stay in registers.  Supported shapes: sf_lstm_seq_supported. */
int sf_gru_seq_fwd(const float *gx, const float *whh, const float *bhh, const float *keep, float *gates, float *hprev,
                   float *hout, uint32_t *sync, int R, int Cn, int H, int env_major, void *stream);
int sf_gru_seq_bwd(const float *dout, const float *gates, const float *hprev, const float *keep, const float *whh,
                   float *dgx, float *dgh, uint32_t *sync, int R, int Cn, int H, int env_major, void *stream);

/* Forward sequence passes WITH the input projection (model/core.py:37-64: nn.LSTM / nn.GRU compute x W_ih^T + b_ih
 * themselves): instead of gx [R][Cn][G*H] — written by a GEMM launch and read back by the pass — the pass takes
 * x [R][Cn][Kx] (time-major core input), wih_t [G*H][Kx] (W_ih in torch's own [gate column][input] layout) and
 * bih [G*H]; every work-group multiplies its own gate columns, the products of step t+1 run while it waits for the
 * other work-groups' h_t.  gx_t = x_t W_ih^T + b_ih is formed in f32 exactly as the GEMM's epilogue does (MFMA sum, then
 * + b_ih), then used as in sf_lstm_seq_fwd / sf_gru_seq_fwd.  sf_seq_fwd_x_supported: Kx == 64 and row groups of at
 * most 128 rows (Cn <= 1024 on 256 CUs), else project with sf_conv_fwd_t and call the gx form. */
int sf_seq_fwd_x_supported(int Cn, int H, int Kx);
int sf_lstm_seq_fwd_x(const float *x, const float *wih_t, const float *bih, int Kx, const float *whh, const float *bhh,
                      const float *keep, float *gates, float *hprev, float *hout, float *cprev, float *cout,
                      uint32_t *sync, int R, int Cn, int H, int env_major, void *stream);
int sf_gru_seq_fwd_x(const float *x, const float *wih_t, const float *bih, int Kx, const float *whh, const float *bhh,
                     const float *keep, float *gates, float *hprev, float *hout, uint32_t *sync, int R, int Cn, int H,
                     int env_major, void *stream);

/* ---- data-parallel learner replicas (SURVEY.md §8(b) "DP -> sf_allreduce_grads", §8(e)) ----------------------------
 * New capability: the reference runs ONE learner per policy (algo/utils/shared_buffers.py:26-32), so these replace no
 * reference function; a host without torch.distributed binds them for the exchange algo/learning/learner.py performs
 * per SGD step.  One RCCL communicator per rank, created once on the rank's current device: rank 0 calls
 * sf_dp_unique_id and ships the SF_DP_UNIQUE_ID_BYTES bytes to the other ranks by any host channel; every rank then
 * calls sf_dp_comm_create (a collective: returns when all nranks ranks joined).  sf_allreduce_grads: in-place SUM of n
 * fp32 values over the ranks (each replica's gradient already carries the GLOBAL 1/n_valid), enqueued on `stream`;
 * calls on one communicator must be issued in the same order on every rank.  sf_dp_allreduce_f64: the 3-double
 * moment / loss-scalar exchanges (op 0 = sum, 1 = max).  sf_dp_broadcast: root's bytes to everyone (initial weights).
 * librccl.so.1 is loaded on first use (dlopen), never at library load. */
#define SF_DP_UNIQUE_ID_BYTES 128
int sf_dp_unique_id(void *out_id);
int sf_dp_comm_create(const void *id_bytes, int nranks, int rank, void **comm_out);
int sf_dp_comm_destroy(void *comm);
int sf_dp_comm_info(void *comm, int *nranks, int *rank);
int sf_allreduce_grads(void *comm, float *grads, int64_t n, void *stream);
int sf_dp_allreduce_f64(void *comm, double *buf, int64_t n, int op, void *stream);
int sf_dp_broadcast(void *comm, void *buf, int64_t nbytes, int root, void *stream);

/* ---- one-shot ("direct") exchange for small buckets (<= 8 MB never goes round a ring) -----------------
 * Every rank owns a mailbox in its HBM which every peer maps (hipIpc) and reads over xGMI: ONE kernel per rank and call
 * publishes the bucket, waits for the peers' flags and adds the peers' copies in rank order (bit-identical results on all
 * ranks; one hop instead of 2 (W-1)).  Meant for the 0.31 MB conv bucket of the gradient, which completes last in the
 * backward pass, and the 24 ... 400-byte moment / loss-scalar buckets; larger buckets stay on sf_allreduce_grads.
 * sf_dp_oneshot_create: allocates the mailbox for buckets of up to max_bytes and writes its SF_DP_IPC_HANDLE_BYTES handle
 * to handle_out; the host ships every rank's handle to every rank (rank order, any channel) and calls
 * sf_dp_oneshot_connect.  The all-reduce calls are collective, in place, enqueued on `stream`, and must be issued in the
 * same order on every rank (they carry a sequence number).  A peer that does not arrive within 5 s sets the context's error
 * word (sf_dp_oneshot_status) instead of hanging the queue.  No reference counterpart (one learner per policy:
 * sample_factory/algo/utils/shared_buffers.py:26-32). */
#define SF_DP_IPC_HANDLE_BYTES 64
int sf_dp_oneshot_create(int nranks, int rank, int64_t max_bytes, void **ctx_out, void *handle_out);
int sf_dp_oneshot_connect(void *ctx, const void *all_handles /* nranks x SF_DP_IPC_HANDLE_BYTES, rank order */);
int sf_dp_oneshot_allreduce_f32(void *ctx, float *buf, int64_t n, void *stream);            /* SUM */
int sf_dp_oneshot_allreduce_f64(void *ctx, double *buf, int64_t n, int op, void *stream);   /* op 0 = SUM, 1 = MAX */
int sf_dp_oneshot_status(void *ctx);
int sf_dp_oneshot_destroy(void *ctx);

/* action means squashed to [-scale, scale] (continuous_tanh_scale > 0, model/action_parameterization.py:62-66), in
 * place on columns [col0, col0+ncols) of a row-major [n, ld] matrix: y = tanh(x/scale)*scale; backward: g *= 1-(y/scale)^2
 * with y the squashed output. */
int sf_tanh_scale_fwd(float *x, int ld, int64_t n, int col0, int ncols, float scale, void *stream);
int sf_tanh_scale_bwd(float *g, const float *y, int ld, int64_t n, int col0, int ncols, float scale, void *stream);

/* Two linear layers into ONE accumulator: out [n][N] = a1 w1t^T + a2 w2t^T + bias1 + bias2 (a_i: [n][K_i] rows lda_i
 * floats apart; w_it: [N][K_i], i.e. torch's own Linear / LSTM weight layout; biases may be NULL).  One inference step
 * of model/core.py:37-64's nn.LSTM is this launch on (x, weight_ih_l0, bias_ih_l0) and (h, weight_hh_l0, bias_hh_l0)
 * followed by sf_rnn_cell_fwd(gh = NULL): the short first product (K1 = 64) rides in the pipeline of the long one instead
 * of being a launch of its own.  gru_H > 0 (nn.GRU, N = 4 * gru_H, w_it = weight_{ih,hh}_l0 [3H][K_i]): the candidate
 * gate's two parts stay apart — columns [0,2H) both products + both biases, [2H,3H) the first layer's rows 2H.. only,
 * [3H,4H) the second layer's rows 2H.. only — which is what sf_rnn_cell_fwd(kind 0, gh = NULL) reads.  K1, K2 multiples of
 * 32; sf_linear_fwd_dual_supported: the launch fills the chip (else use two sf_conv_fwd launches). */
int sf_linear_fwd_dual_supported(int64_t n, int N, int K1, int K2);
int sf_linear_fwd_dual(const float *a1, int64_t lda1, const float *w1t, const float *bias1, int K1, const float *a2,
                       int64_t lda2, const float *w2t, const float *bias2, int K2, float *out, int64_t n, int N,
                       int gru_H, void *stream);

/* Forward for the dense hot layers through the gfx950 LDS-DMA path: same result contract as sf_conv_fwd, but the
 * weights are given Cout-major, wt[Cout, K] (sf_transpose of the canonical [K, Cout] array), the input must be f32
 * NHWC with Cin % 32 == 0, dense samples (no index gather), 16-byte aligned.  sf_conv_fwd_t_supported says whether a
 * launch qualifies (and is large enough to be worth it); everything else goes through sf_conv_fwd.  A wide layer with
 * a long reduction and few rows (the fc layer of one rollout step) is split along K: sf_conv_fwd_t_workspace gives the
 * scratch bytes such a launch needs (0 = none; 16-byte aligned; the call fails loudly if it is missing). */
int sf_conv_fwd_t_supported(int64_t n, const sf_conv_desc *desc);
int64_t sf_conv_fwd_t_workspace(int64_t n, const sf_conv_desc *desc);
int sf_conv_fwd_t(const float *in, int64_t in_sample_stride, const float *wt, const float *bias, float *out,
                  int64_t n, const sf_conv_desc *desc, void *workspace, int64_t workspace_bytes, void *stream);
int sf_transpose(const float *w, float *wt, int K, int N, void *stream); /* wt[N,K] = w[K,N]^T */

/* Profiling aid (no reference counterpart): the kernel instantiation a conv/linear launch resolves to, spelled as
 * rocprofv3 prints it ("k_conv_fwd<128, 64, 2, 2, 0>").  op: 0 forward, 1 wgrad, 2 dgrad. */
int sf_conv_kernel_name(int op, int64_t n, const sf_conv_desc *desc, int split_k_allowed, char *out, int cap);
/* dense layer: out[M,N] = act(in[M,K] * w[K,N] + bias); wgrad: dw[K,N] = in^T dout, db = colsum(dout);
 * dgrad: din[M,K] = (dout[M,N] * w^T) * relu_mask(in_act). */
int sf_linear_fwd(const float *in, const float *w, const float *bias, float *out, int64_t M, int K, int N, int relu,
                  void *stream);
int64_t sf_linear_wgrad_workspace(int64_t M, int K, int N);
int sf_linear_wgrad(const float *in, const float *dout, float *dw, float *db, int64_t M, int K, int N,
                    void *workspace, void *stream);
int sf_linear_dgrad(const float *dout, const float *w, const float *in_act, float *din, int64_t M, int K, int N,
                    void *stream);
/* elementwise helper for the backward chain: g[i] = (act[i] > 0) ? g[i] : 0 */
int sf_relu_mask(float *g, const float *act, int64_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SF_HIP_H */

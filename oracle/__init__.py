"""CPU oracle for the APPO hot path — TEST INFRASTRUCTURE (see sf_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libsf_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "sf_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libsf_oracle.so"])
    return _SO


def build_ref() -> str:
    """stage the reference's own Python package for the cpu_baseline leg (`make -C oracle ref`); returns the archive path
    ('' when neither /root/reference nor a prebuilt archive exists)"""
    subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
    z = os.path.join(_HERE, "_ref", "sample_factory_ref.zip")
    return z if os.path.exists(z) else ""


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.sfo_prepare_batch.restype = C.c_long
        _lib.sfo_clip_grad_norm.restype = C.c_float
    return _lib


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u8(a):
    return np.ascontiguousarray(np.asarray(a).astype(np.uint8))


def gae(rewards, dones, values, valids, gamma, lam):
    rewards, values = _f32(rewards), _f32(values)
    dones, valids = _u8(dones), _u8(valids)
    E, T = rewards.shape
    adv = np.empty((E, T), np.float32)
    lib().sfo_gae(_p(rewards, C.c_float), _p(dones, C.c_uint8), _p(values, C.c_float), _p(valids, C.c_uint8),
                  E, T, C.c_double(gamma), C.c_double(lam), _p(adv, C.c_float))
    return adv


def rms_update(stats, x):
    stats = np.ascontiguousarray(stats, dtype=np.float64).copy()
    x = _f32(x)
    lib().sfo_rms_update(_p(stats, C.c_double), _p(x, C.c_float), C.c_long(x.size))
    return stats


def rms_apply(stats, x, denormalize=False):
    stats = np.ascontiguousarray(stats, dtype=np.float64)
    x = _f32(x).copy()
    lib().sfo_rms_apply(_p(stats, C.c_double), _p(x, C.c_float), C.c_long(x.size), int(denormalize))
    return x


def prepare_batch(rewards, dones, time_outs, values, policy_id, policy_version, actions, log_prob_actions, *,
                  my_policy_id=0, train_step=0, max_policy_lag=1000, normalize_returns=True, value_bootstrap=False,
                  gamma=0.99, lam=0.95, rms=(0.0, 1.0, 1.0)):
    """Returns dict(rewards, valids, advantages, returns, actions, log_prob_actions, rms, num_invalids)."""
    rewards = _f32(rewards).copy()
    E, T = rewards.shape
    values = _f32(values)
    actions = _f32(actions).copy()
    num_actions = actions.size // (E * T)
    logp = _f32(log_prob_actions).copy()
    dones, time_outs = _u8(dones), _u8(time_outs)
    pid = np.ascontiguousarray(policy_id, dtype=np.int32)
    pver = _f32(policy_version)
    rms = np.ascontiguousarray(rms, dtype=np.float64).copy()
    valids = np.empty((E, T + 1), np.uint8)
    adv = np.empty((E, T), np.float32)
    ret = np.empty((E, T), np.float32)
    ninv = lib().sfo_prepare_batch(
        _p(rewards, C.c_float), _p(dones, C.c_uint8), _p(time_outs, C.c_uint8), _p(values, C.c_float),
        _p(pid, C.c_int32), _p(pver, C.c_float), _p(actions, C.c_float), num_actions, _p(logp, C.c_float), E, T,
        int(my_policy_id), int(train_step), int(max_policy_lag), int(normalize_returns), int(value_bootstrap),
        C.c_double(gamma), C.c_double(lam), _p(rms, C.c_double), _p(valids, C.c_uint8), _p(adv, C.c_float),
        _p(ret, C.c_float))
    return dict(rewards=rewards, valids=valids.astype(bool), advantages=adv, returns=ret, actions=actions,
                log_prob_actions=logp, rms=rms, num_invalids=int(ninv))


def vtrace(ratio, values, rewards, dones, recurrence, gamma, rho_hat=1.0, c_hat=1.0):
    ratio, values, rewards, dones = _f32(ratio), _f32(values), _f32(rewards), _f32(dones)
    N = ratio.size
    vs, adv = np.empty(N, np.float32), np.empty(N, np.float32)
    lib().sfo_vtrace(_p(ratio, C.c_float), _p(values, C.c_float), _p(rewards, C.c_float), _p(dones, C.c_float),
                     C.c_long(N), int(recurrence), C.c_double(gamma), C.c_double(rho_hat), C.c_double(c_hat),
                     _p(vs, C.c_float), _p(adv, C.c_float))
    return vs, adv


SCALAR_NAMES = ("policy_loss", "exploration_loss", "kl_loss", "value_loss", "kl_mean", "kl_max", "adv_mean",
                "adv_std", "n_valid", "entropy_mean")


def ppo_loss(params, values, actions, old_logp, old_params, old_values, adv, targets, valids, *, action_kind=0,
             clip_ratio=0.1, clip_value=1.0, value_loss_coeff=0.5, exploration_coeff=0.003, exploration_kind=1,
             kl_coeff=0.0, ext_moments=None, head_sizes=None):
    params = _f32(params)
    N, A = params.shape
    values, actions, old_logp = _f32(values), _f32(actions), _f32(old_logp)
    old_params, old_values, adv, targets = _f32(old_params), _f32(old_values), _f32(adv), _f32(targets)
    valids = _u8(valids)
    sc = np.zeros(16, np.float32)
    gp = np.zeros((N, A), np.float32)
    gv = np.zeros(N, np.float32)
    lib().sfo_ppo_loss(_p(params, C.c_float), _p(values, C.c_float), _p(actions, C.c_float),
                       _p(old_logp, C.c_float), _p(old_params, C.c_float), _p(old_values, C.c_float),
                       _p(adv, C.c_float), _p(targets, C.c_float), _p(valids, C.c_uint8), C.c_long(N), A,
                       int(action_kind), C.c_double(clip_ratio), C.c_double(clip_value),
                       C.c_double(value_loss_coeff), C.c_double(exploration_coeff), int(exploration_kind),
                       C.c_double(kl_coeff),
                       _p(np.ascontiguousarray(ext_moments, dtype=np.float64), C.c_double) if ext_moments is not None else None,
                       _p(sc, C.c_float), _p(gp, C.c_float), _p(gv, C.c_float),
                       (C.c_int * len(head_sizes))(*[int(x) for x in head_sizes]) if head_sizes else None,
                       len(head_sizes) if head_sizes else 0)
    out = {k: float(sc[i]) for i, k in enumerate(SCALAR_NAMES)}
    out["grad_params"] = gp
    out["grad_values"] = gv
    return out


def clip_grad_norm(g, max_norm):
    g = _f32(g).copy()
    total = lib().sfo_clip_grad_norm(_p(g, C.c_float), C.c_long(g.size), C.c_double(max_norm))
    return g, float(total)


def adam_step(p, g, m, v, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-6):
    p, m, v = _f32(p).copy(), _f32(m).copy(), _f32(v).copy()
    g = _f32(g)
    lib().sfo_adam_step(_p(p, C.c_float), _p(g, C.c_float), _p(m, C.c_float), _p(v, C.c_float), C.c_long(p.size),
                        int(step), C.c_double(lr), C.c_double(b1), C.c_double(b2), C.c_double(eps))
    return p, m, v


def sample_masked(logits, mask, seed, step, row0=0):
    logits, mask = _f32(logits), _u8(mask)
    N, A = logits.shape
    actions, logp = np.zeros(N, np.float32), np.zeros(N, np.float32)
    lib().sfo_sample_masked(_p(logits, C.c_float), _p(mask, C.c_uint8), C.c_long(N), A, C.c_uint32(seed),
                            C.c_uint32(step), C.c_uint32(row0), _p(actions, C.c_float), _p(logp, C.c_float))
    return actions, logp


def sample_tuple(logits, head_sizes, seed, step, row0=0):
    logits = _f32(logits)
    N, H = logits.shape[0], len(head_sizes)
    actions = np.zeros((N, sum(1 if int(x) > 0 else -int(x) for x in head_sizes)), np.float32)  # x < 0: Box(-x) member
    logp = np.zeros(N, np.float32)
    lib().sfo_sample_tuple(_p(logits, C.c_float), C.c_long(N), (C.c_int * H)(*[int(x) for x in head_sizes]), H,
                           C.c_uint32(seed), C.c_uint32(step), C.c_uint32(row0), _p(actions, C.c_float),
                           _p(logp, C.c_float))
    return actions, logp


def lamb_step(p, g, m, v, seg_id, nseg, step, lr=1e-4, b1=0.9, b2=0.999, eps=1e-6, weight_decay=1e-4, min_trust=0.01):
    p, m, v = _f32(p).copy(), _f32(m).copy(), _f32(v).copy()
    g, seg_id = _f32(g), _u8(seg_id)
    lib().sfo_lamb_step(_p(p, C.c_float), _p(g, C.c_float), _p(m, C.c_float), _p(v, C.c_float), _p(seg_id, C.c_uint8),
                        C.c_long(p.size), int(nseg), int(step), C.c_double(lr), C.c_double(b1), C.c_double(b2),
                        C.c_double(eps), C.c_double(weight_decay), C.c_double(min_trust))
    return p, m, v


def categorical(logits, actions):
    logits = _f32(logits)
    N, A = logits.shape
    actions = _f32(actions)
    lp = np.empty((N, A), np.float32)
    la = np.empty(N, np.float32)
    ent = np.empty(N, np.float32)
    lib().sfo_categorical(_p(logits, C.c_float), _p(actions, C.c_float), C.c_long(N), A, _p(lp, C.c_float),
                          _p(la, C.c_float), _p(ent, C.c_float))
    return lp, la, ent


def philox(c, k):
    out = np.empty(4, np.uint32)
    lib().sfo_philox(*[C.c_uint32(int(x)) for x in c], *[C.c_uint32(int(x)) for x in k], _p(out, C.c_uint32))
    return out


def synth_obs(n, env0, obs_bytes, seed, step):
    obs = np.empty((n, obs_bytes), np.uint8)
    lib().sfo_synth_obs(_p(obs, C.c_uint8), int(n), int(env0), C.c_long(obs_bytes), C.c_uint32(seed),
                        C.c_uint32(step))
    return obs


def synth_step(actions, env0, num_actions, seed, step):
    actions = np.ascontiguousarray(actions, dtype=np.int32)
    n = actions.size
    rew = np.empty(n, np.float32)
    term = np.empty(n, np.uint8)
    lib().sfo_synth_step(_p(actions, C.c_int32), n, int(env0), int(num_actions), C.c_uint32(seed), C.c_uint32(step),
                         _p(rew, C.c_float), _p(term, C.c_uint8))
    return rew, term.astype(bool)


def sample_categorical(logits, seed, step, row0=0):
    logits = _f32(logits)
    N, A = logits.shape
    act = np.empty(N, np.float32)
    lp = np.empty(N, np.float32)
    lib().sfo_sample_categorical(_p(logits, C.c_float), C.c_long(N), A, C.c_uint32(seed), C.c_uint32(step),
                                 C.c_uint32(row0), _p(act, C.c_float), _p(lp, C.c_float))
    return act, lp


def sample_normal(params, seed, step, row0=0):
    params = _f32(params)
    N, A = params.shape
    D = A // 2
    act = np.empty((N, D), np.float32)
    lp = np.empty(N, np.float32)
    lib().sfo_sample_normal(_p(params, C.c_float), C.c_long(N), D, C.c_uint32(seed), C.c_uint32(step),
                            C.c_uint32(row0), _p(act, C.c_float), _p(lp, C.c_float))
    return act, lp


def rollout_replay(rew, term, trunc, new_rnn, obs, *, T, reward_scale, reward_clip):
    """numpy restatement of the per-step bookkeeping of BatchedVectorEnvRunner (batched_sampling.py): reward scale + clamp
    (:208-213), dones = terminated | truncated and time_outs = truncated (:317,:325-329), recurrent state zeroed AFTER
    production and stored as the input of the next step (:332-335,:383-385), obs / state at [:, T] and their carry-over
    into step 0 of the next rollout (:289-296), episode statistics on RAW rewards (:215-287).
    rew/term/trunc [steps, B], new_rnn [steps, B, R], obs [steps + 1, B, ...] -> list of per-rollout dicts + stats."""
    steps, B = rew.shape
    assert steps % T == 0
    scale, clip = np.float32(reward_scale), np.float32(reward_clip)
    last_rnn = np.zeros_like(new_rnn[0])
    ep_ret, ep_len = np.zeros(B, np.float32), np.zeros(B, np.int32)
    ep_rewards, ep_lens, out = [], [], []
    for r in range(steps // T):
        cur = dict(rewards=np.empty((B, T), np.float32), dones=np.empty((B, T), bool), time_outs=np.empty((B, T), bool),
                   rnn_states=np.empty((B, T + 1) + new_rnn.shape[2:], np.float32),
                   obs=np.empty((B, T + 1) + obs.shape[2:], obs.dtype))
        for t in range(T):
            k = r * T + t
            cur["obs"][:, t], cur["rnn_states"][:, t] = obs[k], last_rnn
            done = term[k] | trunc[k]
            cur["rewards"][:, t] = np.clip(rew[k] * scale, -clip, clip)
            cur["dones"][:, t], cur["time_outs"][:, t] = done, trunc[k]
            last_rnn = new_rnn[k] * (np.float32(1.0) - done.astype(np.float32))[:, None]
            ep_ret = ep_ret + rew[k]
            ep_len = ep_len + 1
            fin = np.flatnonzero(done)
            ep_rewards.append(ep_ret[fin].copy())
            ep_lens.append(ep_len[fin].copy())
            ep_ret[fin], ep_len[fin] = 0, 0
        cur["obs"][:, T], cur["rnn_states"][:, T] = obs[(r + 1) * T], last_rnn
        out.append(cur)
    return out, dict(ep_reward=np.concatenate(ep_rewards), ep_len=np.concatenate(ep_lens), final_ep_reward=ep_ret,
                     final_ep_len=ep_len, final_last_rnn=last_rnn)

"""CPU baseline for bench.py's `cpu_baseline` leg (kind "port") — TEST/BENCH INFRASTRUCTURE, never on the product path.

A bounded sample of the SAME workload (synthetic Atari-shaped env, Nature-CNN actor-critic, NS-2 hyper-parameters)
executed on the host cores: the network through plain torch CPU fp32 ops with autograd (what the reference's CPU path
executes: model/encoder.py:90-119 under torch), everything else through the C oracle (sf_oracle.c).  It follows the
reference's data flow including the f32 observation materialisation (rl_utils.py:36-42, learner.py:925-941).
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch
import torch.nn.functional as F

import oracle


def _net(params, x):
    h = F.relu(F.conv2d(x, params[0], params[1], stride=4))
    h = F.relu(F.conv2d(h, params[2], params[3], stride=2))
    h = F.relu(F.conv2d(h, params[4], params[5], stride=1))
    h = F.relu(F.linear(h.flatten(1), params[6], params[7]))
    return F.linear(h, params[8], params[9])  # [B, 1+A]: value | logits


def run(num_envs=256, rollout=32, num_minibatches=4, A=6, seed=0, budget_seconds=24.0, threads=None):
    """Times a BOUNDED sample and extrapolates linearly to one full iteration of `num_envs` envs:
    up to `rollout` inference steps (stops after ~1/3 of the budget) and up to `num_minibatches` SGD steps (rest)."""
    cores = len(os.sched_getaffinity(0))
    threads = int(threads or min(cores, 64))  # torch-CPU conv on batch<=2048 stops scaling well before 64 threads
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(seed)
    shapes = [(32, 4, 8, 8), (32,), (64, 32, 4, 4), (64,), (64, 64, 3, 3), (64,), (512, 3136), (512,), (1 + A, 512), (1 + A,)]
    params = []
    for s_ in shapes:
        t = torch.zeros(s_)
        if len(s_) > 1:
            torch.nn.init.orthogonal_(t.view(s_[0], -1), generator=g)
        params.append(t.requires_grad_(True))
    opt = torch.optim.Adam(params, lr=1e-4, betas=(0.9, 0.999), eps=1e-6)
    B, T = num_envs, rollout
    obs = np.zeros((B, T + 1, 4 * 84 * 84), np.uint8)
    actions = np.zeros((B, T), np.float32)
    logits = np.zeros((B, T, A), np.float32)
    logp = np.zeros((B, T), np.float32)
    values = np.zeros((B, T + 1), np.float32)
    rewards = np.zeros((B, T), np.float32)
    dones = np.zeros((B, T), bool)
    for t in range(T + 1):  # frames for every step up front (not timed: the env is not part of the learner's cost)
        obs[:, t] = oracle.synth_obs(B, 0, 28224, seed, t)
    with torch.no_grad():   # untimed warm-up (thread pool, oneDNN primitive cache)
        _net(params, torch.from_numpy(obs[:, 0]).view(B, 4, 84, 84).float().mul_(1.0 / 255.0))
    t0 = time.perf_counter()
    steps_done = 0
    with torch.no_grad():
        for t in range(T):
            x = torch.from_numpy(obs[:, t]).view(B, 4, 84, 84).float().mul_(1.0 / 255.0)   # rl_utils.py:36-42
            heads = _net(params, x).numpy()
            a, lp = oracle.sample_categorical(heads[:, 1:], seed, t)
            actions[:, t], logp[:, t], logits[:, t], values[:, t] = a, lp, heads[:, 1:], heads[:, 0]
            rewards[:, t], dones[:, t] = oracle.synth_step(a.astype(np.int32), 0, A, seed, t)
            steps_done += 1
            if time.perf_counter() - t0 > budget_seconds / 3:
                break
    t_roll = (time.perf_counter() - t0) * T / steps_done
    if steps_done < T:  # fill the untimed remainder so the learner sample has sane inputs
        actions[:, steps_done:] = 0
        logp[:, steps_done:] = -1.79
        logits[:, steps_done:] = 0
    t1 = time.perf_counter()
    pb = oracle.prepare_batch(rewards, dones, np.zeros_like(dones), values, np.zeros((B, T), np.int32),
                              np.zeros((B, T), np.float32), actions, logp)
    N = B * T
    mb = N // num_minibatches
    flat_obs = obs[:, :T].reshape(N, 4, 84, 84)
    mbs_done = 0
    for k in range(num_minibatches):
        sl = slice(k * mb, (k + 1) * mb)
        x = torch.from_numpy(flat_obs[sl]).float().mul_(1.0 / 255.0)                      # learner.py:925-941
        heads = _net(params, x)
        hn = heads.detach().numpy()
        out = oracle.ppo_loss(hn[:, 1:], hn[:, 0], actions.reshape(N)[sl], pb["log_prob_actions"].reshape(N)[sl],
                              logits.reshape(N, A)[sl], values[:, :T].reshape(N)[sl], pb["advantages"].reshape(N)[sl],
                              pb["returns"].reshape(N)[sl], pb["valids"][:, :T].reshape(N)[sl], exploration_coeff=0.01)
        gh = torch.from_numpy(np.concatenate([out["grad_values"][:, None], out["grad_params"]], 1))
        opt.zero_grad(set_to_none=True)
        heads.backward(gh)
        torch.nn.utils.clip_grad_norm_(params, 4.0)
        opt.step()
        mbs_done += 1
        if time.perf_counter() - t0 > budget_seconds:
            break
    t_train = (time.perf_counter() - t1) * num_minibatches / mbs_done
    wall = time.perf_counter() - t0
    return dict(value=N / (t_roll + t_train), unit="env-steps/s", cores=threads, kind="port",
                sample=f"{B} envs: {steps_done}/{T} inference steps + {mbs_done}/{num_minibatches} SGD minibatches of "
                       f"{mb} samples timed ({wall:.1f}s) and extrapolated to one {B}x{T} iteration; torch-CPU fp32 "
                       f"Nature-CNN fwd/bwd (f32 obs materialised as the reference does) + C oracle sampling/GAE/"
                       f"loss; {threads} threads of {cores} host cores")


if __name__ == "__main__":
    print(run())

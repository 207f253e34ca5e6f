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


def run(num_envs=256, rollout=32, num_minibatches=4, A=6, seed=0, max_seconds=40.0):
    cores = len(os.sched_getaffinity(0))
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(seed)
    shapes = [(32, 4, 8, 8), (32,), (64, 32, 4, 4), (64,), (64, 64, 3, 3), (64,), (512, 3136), (512,), (1 + A, 512), (1 + A,)]
    params = []
    for s in shapes:
        t = torch.zeros(s)
        if len(s) > 1:
            torch.nn.init.orthogonal_(t.view(s[0], -1), generator=g)
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
    t0 = time.perf_counter()
    obs[:, 0] = oracle.synth_obs(B, 0, 28224, seed, 0)
    with torch.no_grad():
        for t in range(T):
            x = torch.from_numpy(obs[:, t]).view(B, 4, 84, 84).float().mul_(1.0 / 255.0)
            heads = _net(params, x).numpy()
            a, lp = oracle.sample_categorical(heads[:, 1:], seed, t)
            actions[:, t], logp[:, t], logits[:, t], values[:, t] = a, lp, heads[:, 1:], heads[:, 0]
            rewards[:, t], dones[:, t] = oracle.synth_step(a.astype(np.int32), 0, A, seed, t)
            obs[:, t + 1] = oracle.synth_obs(B, 0, 28224, seed, t + 1)
        x = torch.from_numpy(obs[:, T]).view(B, 4, 84, 84).float().mul_(1.0 / 255.0)
        values[:, T] = _net(params, x).numpy()[:, 0]
    t_roll = time.perf_counter() - t0
    pb = oracle.prepare_batch(rewards, dones, np.zeros_like(dones), values, np.zeros((B, T), np.int32),
                              np.zeros((B, T), np.float32), actions, logp)
    N = B * T
    mb = N // num_minibatches
    flat_obs = obs[:, :T].reshape(N, 4, 84, 84)
    for k in range(num_minibatches):
        sl = slice(k * mb, (k + 1) * mb)
        x = torch.from_numpy(flat_obs[sl]).float().mul_(1.0 / 255.0)
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
        if time.perf_counter() - t0 > max_seconds:
            num_minibatches_done = k + 1
            break
    else:
        num_minibatches_done = num_minibatches
    dt = time.perf_counter() - t0
    frac = (t_roll + (dt - t_roll) * num_minibatches / num_minibatches_done) if num_minibatches_done else dt
    return dict(value=N / frac, unit="env-steps/s", cores=cores, kind="port",
                sample=f"{B} envs x {T} steps rollout + {num_minibatches_done}/{num_minibatches} minibatches of {mb} "
                       f"(torch-CPU fp32 Nature-CNN fwd/bwd + C oracle GAE/loss, {cores} threads, {dt:.1f}s)")


if __name__ == "__main__":
    print(run())

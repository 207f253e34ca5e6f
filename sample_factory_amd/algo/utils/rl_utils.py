"""RL math utilities with the reference's names (sample_factory/algo/utils/rl_utils.py), backed by HIP kernels."""
from __future__ import annotations

import torch

from sample_factory_amd import lib


def trajectories_per_minibatch(cfg) -> int:
    return cfg.batch_size // cfg.rollout


def trajectories_per_training_iteration(cfg) -> int:
    return cfg.num_batches_per_epoch * trajectories_per_minibatch(cfg)


def total_num_envs(cfg) -> int:
    return cfg.num_workers * cfg.num_envs_per_worker


def total_num_agents(cfg, env_info) -> int:
    return total_num_envs(cfg) * env_info.num_agents


def samples_per_trajectory(trajectory) -> int:
    """rl_utils.py:45-48: env steps in a [batch, rollout, ...] trajectory dict"""
    batch, rollout = trajectory["rewards"].shape[:2]
    return int(batch) * int(rollout)


def gae_advantages(rewards, dones, values, valids, γ: float, λ: float):
    """rl_utils.py:78-94: rewards/dones [E,T], values/valids [E,T+1] -> advantages [E,T] (GPU tensors)."""
    adv = torch.empty_like(rewards)
    ret = torch.empty_like(rewards)
    lib.gae_returns(rewards, dones, None, values, valids, None, γ, λ, False, adv, ret)
    return adv


def make_dones(terminated, truncated):
    """rl_utils.py:100-110"""
    return terminated | truncated

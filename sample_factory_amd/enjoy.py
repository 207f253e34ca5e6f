"""Evaluation of a trained policy: the consumer on the other side of the checkpoint format (reference:
sample_factory/enjoy.py:103-290; `python -m sample_factory.enjoy --env=... --experiment=...`).

Same contract — `enjoy(cfg) -> (status, average episode reward)`; the experiment's saved `config.json` is the base
configuration, flags given on the command line override it (`cfg/arguments.py:227-260`); `--load_checkpoint_kind`
latest | best; `--eval_deterministic` replaces sampling by `argmax_actions` (`action_distributions.py:73-81`: arg-max
of the logits per Discrete head, the means for a Box); recurrent state is zeroed when an episode ends; evaluation stops
after `--max_num_episodes` episodes or `--max_num_frames` env steps.

What differs is where it runs: the env's agents are evaluated as ONE batch on the device with the rollout kernels of
the training path (forward in place on the slab slot, `sf_sample_write_step*` with its deterministic switch, episode
returns accumulated by `sf_traj_write_env_step`), T steps per host round trip; nothing is trained and the observation
normaliser is frozen (eval mode).  Rendering, video and hub upload are outside the hot-path scope.
Works on checkpoints written by this engine and by the reference (same `.pth` layout, tests/test_gpu_runner.py).
"""
from __future__ import annotations

import json
import os
import sys
from typing import Tuple

import torch

from sample_factory_amd.algo.utils.misc import ExperimentStatus


from sample_factory_amd.cfg.arguments import load_from_checkpoint  # noqa: E402,F401  (cfg/arguments.py:227-260)
from sample_factory_amd.utils.utils import cfg_file, log  # noqa: E402,F401


def load_state_dict(cfg, actor_critic, device) -> dict:
    """enjoy.py:92-100"""
    from sample_factory_amd.algo.learning.learner import Learner
    prefix = dict(latest="checkpoint", best="best")[cfg.load_checkpoint_kind]
    ckpts = Learner.get_checkpoints(Learner.checkpoint_dir(cfg, getattr(cfg, "policy_index", 0)), f"{prefix}_*")
    ckpt = Learner.load_checkpoint(ckpts, device)
    if not ckpt:
        raise RuntimeError("Could not load checkpoint")
    actor_critic.load_state_dict(ckpt["model"])
    return ckpt


def enjoy(cfg) -> Tuple[int, float]:
    from sample_factory_amd import lib
    from sample_factory_amd.algo.sampling.batched_sampling import BatchedVectorEnvRunner
    from sample_factory_amd.algo.utils.env_info import extract_env_info
    from sample_factory_amd.algo.utils.shared_buffers import alloc_trajectory_tensors
    from sample_factory_amd.cfg.arguments import preprocess_cfg
    from sample_factory_amd.envs.env_utils import create_env
    from sample_factory_amd.model.actor_critic import get_rnn_size
    from sample_factory_amd.model.model_factory import create_actor_critic
    from sample_factory_amd.utils.attr_dict import AttrDict

    cfg = load_from_checkpoint(cfg)
    if getattr(cfg, "device", "gpu") == "cpu" or not torch.cuda.is_available():
        log.error("enjoy(): needs an MI355X (--device=gpu). sample_factory_amd has no CPU execution path.")
        return ExperimentStatus.FAILURE, 0.0
    lib.load()
    dev = torch.device("cuda", torch.cuda.current_device())
    env = create_env(cfg.env, cfg, AttrDict(worker_index=0, vector_index=0, env_id=0))
    from sample_factory_amd.algo.sampling.parallel_env import ParallelHostEnvs, env_is_batched
    if not hasattr(env, "step_into") and not env_is_batched(env):
        # a single-agent gym-style env (what gym.make returns): evaluated behind the same one-agent view the training
        # runner uses for it (make_env.py:97-128: auto-reset on done), in this process (enjoy.py:113-119 makes ONE env)
        from sample_factory_amd.envs.env_utils import registered_env_factory
        env.close()
        env = ParallelHostEnvs(cfg, cfg.env, registered_env_factory(cfg.env), 1, 1, num_splits=1, inline=True).views[0]
    env_info = extract_env_info(env, cfg)
    if cfg.recurrence == -1:
        preprocess_cfg(cfg, env_info)
    actor_critic = create_actor_critic(cfg, env_info.obs_space, env_info.action_space, dev)
    actor_critic.eval()
    ckpt = load_state_dict(cfg, actor_critic, dev)

    n, T = env_info.num_agents, int(cfg.rollout)
    slab = alloc_trajectory_tensors(env_info, n, T, get_rnn_size(cfg), dev)
    version = torch.full((1,), float(ckpt.get("train_step", 0)), dtype=torch.float32)
    sampler = BatchedVectorEnvRunner(cfg, env_info, env, actor_critic, slab, getattr(cfg, "policy_index", 0), version,
                                     sample_seed=(cfg.seed or 0), tag="inf")
    sampler.reset()
    deterministic = bool(getattr(cfg, "eval_deterministic", False))
    max_frames = getattr(cfg, "max_num_frames", None)
    max_episodes = getattr(cfg, "max_num_episodes", None)
    frames, episodes, ret_sum = 0, 0.0, 0.0
    while True:
        sampler.rollout(float(version[0]), deterministic=deterministic)
        sampler.carry_over()
        frames += T
        st = sampler.ep_stats.cpu()            # {sum of episode returns, sum of lengths, episodes} so far
        ret_sum, episodes = float(st[0]), float(st[2])
        if (max_episodes is not None and episodes >= max_episodes) or (max_frames is not None and frames > max_frames):
            break
    if hasattr(env, "close"):
        env.close()
    return ExperimentStatus.SUCCESS, (ret_sum / episodes if episodes else 0.0)


def main(argv=None) -> int:
    """`python -m sample_factory_amd.enjoy --env=<registered env> --experiment=<name> [--train_dir=...]`; scripts with
    their own envs call register_env first and then enjoy(cfg), as with the reference (sf_examples/*/enjoy_*.py)"""
    from sample_factory_amd.cfg.arguments import parse_full_cfg, parse_sf_args
    parser, _ = parse_sf_args(argv, evaluation=True)
    cfg = parse_full_cfg(parser, argv)
    status, avg = enjoy(cfg)
    print(f"Avg episode reward: {avg:.3f}")
    return status


if __name__ == "__main__":
    sys.exit(main())

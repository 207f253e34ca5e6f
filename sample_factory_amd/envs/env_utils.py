"""Env plugin registry — `register_env(env_name, make_env_func)` as in sample_factory/envs/env_utils.py:12-31;
the factory is called as make_env_func(full_env_name, cfg, env_config, render_mode) (envs/create_env.py:38-39)."""
from __future__ import annotations

from typing import Callable, Dict

_ENV_REGISTRY: Dict[str, Callable] = {}


class EnvCriticalError(Exception):
    """an env integration raises it for a failure the env cannot recover from by resetting (envs/env_utils.py:34-35:
    sf_examples/vizdoom/doom/doom_gym.py when the game process is gone); it propagates out of the rollout and stops the run"""


def register_env(env_name: str, make_env_func: Callable) -> None:
    if env_name in _ENV_REGISTRY:
        print(f"[sample_factory_amd] env {env_name} already registered, overwriting")
    assert callable(make_env_func), f"{make_env_func=} must be callable"
    _ENV_REGISTRY[env_name] = make_env_func


def create_env(full_env_name: str, cfg=None, env_config=None, render_mode=None):
    """envs/create_env.py:13-46"""
    if full_env_name not in _ENV_REGISTRY:
        raise ValueError(f"Env name {full_env_name} is not registered. See register_env()! "
                         f"(available names: {list(_ENV_REGISTRY)})")
    return _ENV_REGISTRY[full_env_name](full_env_name, cfg, env_config, render_mode)


def registered_env_factory(full_env_name: str) -> Callable:
    """the factory registered under this name (handed to env worker processes, which call it themselves)"""
    if full_env_name not in _ENV_REGISTRY:
        raise ValueError(f"Env name {full_env_name} is not registered. See register_env()!")
    return _ENV_REGISTRY[full_env_name]


# ---- optional env interfaces (envs/env_utils.py:60-133 of the reference): same names, same calling convention
def find_wrapper_interface(env, interface_type):
    """Unwrap `env` (gym-style `.env` chain, ending at `.unwrapped` if the env has one) until a layer implements
    `interface_type`; None if no layer does."""
    bottom = getattr(env, "unwrapped", None)
    while env is not None:
        if isinstance(env, interface_type):
            return env
        if env is bottom or not hasattr(env, "env"):
            return None
        env = env.env
    return None


class RewardShapingInterface:
    def get_default_reward_shaping(self):
        """dict of str -> float describing the current reward shaping scheme"""
        raise NotImplementedError

    def set_reward_shaping(self, reward_shaping, agent_idx) -> None:
        """agent_idx: int, or a slice of agents in batched mode"""
        raise NotImplementedError


def get_default_reward_shaping(env):
    iface = find_wrapper_interface(env, RewardShapingInterface)
    return iface.get_default_reward_shaping() if iface else None


def set_reward_shaping(env, reward_shaping, agent_idx) -> None:
    if reward_shaping is None:
        return
    iface = find_wrapper_interface(env, RewardShapingInterface)
    if iface:
        iface.set_reward_shaping(reward_shaping, agent_idx)


class TrainingInfoInterface:
    """Envs that want to know how far training has progressed (curricula) inherit this; the runner calls
    set_training_info({'approx_total_training_steps': env_steps}) before every rollout (batched_sampling.py:352-355,
    runner.py:430-440)."""

    def __init__(self):
        self.training_info = dict()

    def set_training_info(self, training_info) -> None:
        self.training_info = training_info


def find_training_info_interface(env):
    return find_wrapper_interface(env, TrainingInfoInterface)


def set_training_info(training_info_interface, training_info) -> None:
    if training_info_interface:
        training_info_interface.set_training_info(training_info)

"""Env plugin registry — `register_env(env_name, make_env_func)` as in sample_factory/envs/env_utils.py:12-31;
the factory is called as make_env_func(full_env_name, cfg, env_config, render_mode) (envs/create_env.py:38-39)."""
from __future__ import annotations

from typing import Callable, Dict

_ENV_REGISTRY: Dict[str, Callable] = {}


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

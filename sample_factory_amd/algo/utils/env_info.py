"""EnvInfo — the fields of sample_factory/algo/utils/env_info.py:22-39 that the hot path reads."""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, List, Optional


@dataclass
class EnvInfo:
    obs_space: Any
    action_space: Any
    num_agents: int
    gpu_actions: bool = True
    gpu_observations: bool = True
    action_splits: Optional[List[int]] = None
    all_discrete: Optional[bool] = None
    frameskip: int = 1
    reward_shaping_scheme: Optional[Dict[str, float]] = None
    env_info_protocol_version: Optional[int] = 1


def extract_env_info(env, cfg) -> EnvInfo:
    """env_info.py:42-78"""
    return EnvInfo(env.observation_space, env.action_space, env.num_agents, bool(cfg.env_gpu_actions),
                   bool(cfg.env_gpu_observations), None, None, int(cfg.env_frameskip))

"""Env construction names of sample_factory/algo/utils/make_env.py:36-46,338-351 that integrations import.

The reference wraps what `make_env_func` returns in a stack of gym wrappers (dict observations, per-agent lists,
auto-reset, tensor conversion: make_env.py:48-237) and, for batched sampling over single-agent envs, in
`SequentialVectorizeWrapper` (:240-331).  Here that stack is ONE object: `ParallelHostEnvs` steps the instances (in
worker processes or inline) and writes their outputs into page-locked arrays in slab row order; a `ParallelVecEnvView`
is the batched env the sampler sees.  Device-resident envs (tensors out, `num_agents` attribute) are used as they are.
"""
from __future__ import annotations

from typing import Any, Optional, Tuple

from sample_factory_amd.envs.env_utils import create_env, registered_env_factory


def get_multiagent_info(env: Any) -> Tuple[bool, int]:
    """(speaks per-agent vectors?, number of agents) from the optional `is_multiagent` / `num_agents` attributes"""
    num_agents = int(getattr(env, "num_agents", 1))
    is_multiagent = bool(getattr(env, "is_multiagent", num_agents > 1))
    assert is_multiagent or num_agents == 1, f"Invalid configuration: {is_multiagent=} and {num_agents=}"
    return is_multiagent, num_agents


def is_multiagent_env(env: Any) -> bool:
    return get_multiagent_info(env)[0]


def make_env_func_batched(cfg, env_config, render_mode: Optional[str] = None):
    """one batched env for cfg.env: a batched / device env as its factory made it, a single-agent gym-style env behind the
    one-agent batched view (dict observations, agent axis, auto-reset) the rollout runner uses for it"""
    env = create_env(cfg.env, cfg=cfg, env_config=env_config, render_mode=render_mode)
    if hasattr(env, "step_into") or is_multiagent_env(env):
        return env
    from sample_factory_amd.algo.sampling.parallel_env import ParallelHostEnvs, probe_info
    probed = probe_info(env)  # spaces / agents of the instance just built: ParallelHostEnvs does not build another probe
    env.close()
    return ParallelHostEnvs(cfg, cfg.env, registered_env_factory(cfg.env), 1, 1, num_splits=1, inline=True,
                            render_mode=render_mode, probed=probed).views[0]

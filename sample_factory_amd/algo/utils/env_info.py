"""EnvInfo (sample_factory/algo/utils/env_info.py:22-145): what the learner, the trajectory slab and the sampler need to
know about an env, how it is extracted, checked against a cached copy, and obtained from a throw-away child process."""
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


ENV_INFO_PROTOCOL_VERSION = 1  # bump when EnvInfo's fields change: cached entries of another version are ignored


def extract_env_info(env, cfg) -> EnvInfo:
    """env_info.py:42-78: spaces, agents, where actions / observations live, frameskip, the env's default reward shaping
    and — for Tuple action spaces — the number of action components per member and whether all members are Discrete"""
    from sample_factory_amd.envs.env_utils import get_default_reward_shaping
    from sample_factory_amd.envs.spaces import calc_num_actions, is_discrete, is_tuple

    action_space = env.action_space
    splits = all_discrete = None
    if is_tuple(action_space):
        splits = [calc_num_actions(sp) for sp in action_space.spaces]
        all_discrete = all(is_discrete(sp) for sp in action_space.spaces)
    return EnvInfo(env.observation_space, action_space, int(env.num_agents), bool(cfg.env_gpu_actions),
                   bool(cfg.env_gpu_observations), splits, all_discrete, int(cfg.env_frameskip),
                   get_default_reward_shaping(env), ENV_INFO_PROTOCOL_VERSION)


def env_info_cache_filename(cfg) -> str:
    """env_info.py:113-114: one cache entry per env name under the per-user temporary directory"""
    import os

    from sample_factory_amd.utils.utils import project_tmp_dir
    return os.path.join(project_tmp_dir(), f"env_info_{cfg.env}")


def _describe(info: EnvInfo) -> tuple:
    """what two EnvInfo objects are compared by: space descriptors have no __eq__ of their own (gymnasium's do, the bundled
    duck-typed ones print their defining fields)"""
    return (repr(info.obs_space), repr(info.action_space), info.num_agents, info.gpu_actions, info.gpu_observations,
            info.action_splits, info.all_discrete, info.frameskip, info.reward_shaping_scheme,
            info.env_info_protocol_version)


def check_env_info(env, env_info: EnvInfo, cfg) -> None:
    """env_info.py:81-100: the env built for training must be the env the (possibly cached) info describes; a stale cache
    entry is deleted and the run stops with the reference's message"""
    import os

    from sample_factory_amd.utils.utils import log
    fresh = extract_env_info(env, cfg)
    if _describe(fresh) == _describe(env_info):
        return
    cache = env_info_cache_filename(cfg)
    log.error(f"Env info does not match the cached value: {env_info} != {fresh}. Deleting the cache entry {cache}")
    try:
        os.remove(cache)
    except OSError:
        pass
    log.error("This is likely because the environment has changed after the cache entry was created. Either restart the "
              "experiment to fix this or run with --use_env_info_cache=False to avoid such problems in the future.")
    raise ValueError("Env info mismatch. See logs above for details.")


def _probe_env(env_name: str, factory, cfg, queue) -> None:
    """child process: build one batched env the way training will, report its EnvInfo (or the exception) and exit"""
    try:
        from sample_factory_amd.algo.utils.make_env import make_env_func_batched
        from sample_factory_amd.envs.env_utils import register_env
        register_env(env_name, factory)  # a spawned interpreter starts with an empty registry
        env = make_env_func_batched(cfg, env_config=None)
        info = extract_env_info(env, cfg)
        env.close()
        queue.put(info)
    except BaseException as exc:  # noqa: BLE001 - reported to the parent, which raises
        queue.put(RuntimeError(f"env probe for {env_name!r} failed: {type(exc).__name__}: {exc}"))


def obtain_env_info_in_a_separate_process(cfg, timeout: float = 600.0) -> EnvInfo:
    """env_info.py:117-145: EnvInfo of cfg.env without ever constructing the env in THIS process (simulators with GL
    contexts, big assets, or a runtime that must not be initialised before the workers fork off): a spawned child builds
    it, answers through a queue and exits.  `cfg.use_env_info_cache` keeps the answer on disk per env name."""
    import os
    import pickle
    import queue as _queue

    from sample_factory_amd.algo.utils.multiprocessing_utils import get_mp_ctx
    from sample_factory_amd.envs.env_utils import registered_env_factory
    from sample_factory_amd.utils.utils import log

    cache = env_info_cache_filename(cfg)
    use_cache = bool(getattr(cfg, "use_env_info_cache", False))
    if use_cache and os.path.isfile(cache) and _cache_entry_is_ours(cache):
        try:
            with open(cache, "rb") as f:
                info = pickle.load(f)
            if getattr(info, "env_info_protocol_version", None) == ENV_INFO_PROTOCOL_VERSION:
                log.debug(f"Loading env info from cache: {cache}")
                return info
        except Exception as exc:  # noqa: BLE001 - an unreadable entry is a cache miss
            log.warning(f"ignoring unreadable env info cache entry {cache}: {exc}")
    ctx = get_mp_ctx(serial=False)
    q = ctx.Queue()
    p = ctx.Process(target=_probe_env, args=(cfg.env, registered_env_factory(cfg.env), cfg, q), daemon=True)
    p.start()
    import time
    info, deadline = None, time.monotonic() + timeout
    try:
        while info is None:
            try:
                info = q.get(timeout=1.0)
            except _queue.Empty:
                if not p.is_alive():  # a probe that died without a word (segfault in a simulator, OOM kill): fail NOW
                    try:
                        info = q.get(timeout=0.5)  # ... unless its answer was still in flight
                    except _queue.Empty:
                        raise RuntimeError(f"the env probe process for {cfg.env!r} died with exit code {p.exitcode} "
                                           f"without reporting env info") from None
                elif time.monotonic() > deadline:
                    p.kill()
                    raise RuntimeError(f"no env info for {cfg.env!r} after {timeout:.0f} s (the probe process is still "
                                       f"running; killed)") from None
    finally:
        p.join(timeout=10)
    if isinstance(info, Exception):
        raise info
    if use_cache:
        with open(cache, "wb") as f:
            pickle.dump(info, f)
    return info


def _cache_entry_is_ours(path: str) -> bool:
    """an env-info cache entry is unpickled only when the file and its directory belong to this user and nobody else can
    write there (the path under the shared temporary directory is predictable)"""
    import os
    import stat

    from sample_factory_amd.utils.utils import log
    try:
        for p_ in (os.path.dirname(path), path):
            st = os.stat(p_)
            if st.st_uid != os.getuid() or (st.st_mode & (stat.S_IWGRP | stat.S_IWOTH)):
                log.warning(f"ignoring env info cache entry {path}: {p_} is not exclusively owned by this user")
                return False
        return True
    except OSError:
        return False

"""Gym -> Gymnasium compatibility names (sample_factory/algo/utils/gymnasium_utils.py:23-133), imported by env
integrations that build their spaces with another library (`convert_space` in sf_examples/brax/train_brax.py and the
isaacgym example — the device-env pattern of SURVEY.md §2.2).

This engine duck-types spaces (`envs/spaces.py`: `.n` = Discrete, `.shape` + `.dtype` = Box, `.spaces` = Dict / Tuple), so
conversion is: gymnasium spaces pass through; anything else is rebuilt from its duck-typed fields as a gymnasium space
when gymnasium is importable, otherwise as the bundled descriptor."""
from __future__ import annotations

from sample_factory_amd.envs import spaces as _sp
from sample_factory_amd.utils.utils import log

try:  # pragma: no cover - depends on the installation
    import gymnasium as _gymnasium
except Exception:  # noqa: BLE001
    _gymnasium = None


def _target():
    return _gymnasium.spaces if _gymnasium is not None else _sp


def convert_space(space):
    if _gymnasium is not None and isinstance(space, _gymnasium.Space):
        return space
    t = _target()
    if hasattr(space, "spaces"):
        if hasattr(space.spaces, "items"):
            return t.Dict({k: convert_space(v) for k, v in space.spaces.items()})
        return t.Tuple([convert_space(v) for v in space.spaces])
    if hasattr(space, "n"):
        return t.Discrete(int(space.n))
    if hasattr(space, "shape") and hasattr(space, "dtype"):
        if _gymnasium is None and isinstance(space, _sp.Box):
            return space
        return t.Box(getattr(space, "low", -float("inf")), getattr(space, "high", float("inf")), tuple(space.shape),
                     space.dtype)
    raise ValueError(f"The space is of type {type(space)}: neither a Gymnasium space nor something with Discrete / Box / "
                     "Dict / Tuple fields")


def patch_non_gymnasium_env(env):
    """an env built on another space library: give it spaces this engine (and gymnasium code) understands.  The 5-tuple
    step / (obs, info) reset API is the env author's responsibility — there is no shimmy here."""
    try:
        env.observation_space = convert_space(env.observation_space)
        env.action_space = convert_space(env.action_space)
    except AttributeError:
        log.warning("Could not patch spaces for the environment. Consider switching to Gymnasium API.")
    return env

"""Type aliases of the plugin surface under the reference's module path (sample_factory/utils/typing.py:11-34): user
scripts annotate their factories with these (`def make_custom_encoder(cfg: Config, obs_space: ObsSpace) -> Encoder`).
Spaces are duck-typed in this engine (`envs/spaces.py`), so the space aliases are gymnasium's classes when gymnasium is
importable and `Any` otherwise — annotations only, never isinstance-checked by the hot path."""
from __future__ import annotations

import argparse
from typing import Any, Callable, Dict, Optional, Tuple, Union

import torch

from sample_factory_amd.utils.attr_dict import AttrDict

Config = Union[argparse.Namespace, AttrDict]
StatusCode = int
PolicyID = int
Device = str
MpQueue = Any
MpLock = Any
Env = Any
try:  # pragma: no cover - depends on the installation
    from gymnasium import spaces as _spaces
    ObsSpace = Union[_spaces.Space, _spaces.Dict]
    ActionSpace = _spaces.Space
except Exception:  # noqa: BLE001 - gymnasium is optional here
    ObsSpace = Any
    ActionSpace = Any
# make_env_func(full_env_name, cfg, env_config, render_mode) -> env  (envs/create_env.py:38-39)
CreateEnvFunc = Callable[[str, Optional[Config], Optional[AttrDict], Optional[str]], Env]
ActionDistribution = Any
InitModelData = Tuple[PolicyID, Dict, torch.device, int]

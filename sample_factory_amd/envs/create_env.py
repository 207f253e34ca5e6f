"""`create_env` under its reference path (sample_factory/envs/create_env.py:13-46)."""
from sample_factory_amd.envs.env_utils import create_env  # noqa: F401

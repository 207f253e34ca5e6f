"""Global registries — sample_factory/algo/utils/context.py:1-60: `global_model_factory()` (model plugin) and
`global_env_registry()` (env plugin); one process per GPU, so "global" is per process as in the reference."""
from sample_factory_amd.envs.env_utils import _ENV_REGISTRY
from sample_factory_amd.model.model_factory import ModelFactory, global_model_factory  # noqa: F401


def global_env_registry():
    return _ENV_REGISTRY

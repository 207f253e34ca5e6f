"""Global registries — sample_factory/algo/utils/context.py:1-60: `global_model_factory()` (model plugin) and
`global_env_registry()` (env plugin); one process per GPU, so "global" is per process as in the reference.
`reset_global_context()` (tests) empties both registries in place: the registry objects are shared with env worker
processes and the alias package by identity, so they are cleared, never replaced."""
from sample_factory_amd.envs.env_utils import _ENV_REGISTRY
from sample_factory_amd.model.model_factory import ModelFactory, global_model_factory  # noqa: F401


class SampleFactoryContext:
    """the two registries as one object (what `sf_global_context()` hands out in the reference)"""

    @property
    def env_registry(self):
        return _ENV_REGISTRY

    @property
    def model_factory(self) -> ModelFactory:
        return global_model_factory()


_CONTEXT = SampleFactoryContext()


def sf_global_context() -> SampleFactoryContext:
    return _CONTEXT


def global_env_registry():
    return _ENV_REGISTRY


def reset_global_context() -> None:
    """forget every registered env and model component (call after a test that registered something)"""
    _ENV_REGISTRY.clear()
    global_model_factory().reset()

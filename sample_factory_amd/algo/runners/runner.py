"""`Runner` / `AlgoObserver` under the reference's module path (sample_factory/algo/runners/runner.py:52-73,76-…): the
one-process-per-GPU runner lives in sample_factory_amd/train.py."""
from sample_factory_amd.train import AlgoObserver, Runner  # noqa: F401

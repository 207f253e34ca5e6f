"""`ParameterServer` under its reference path (sample_factory/algo/utils/model_sharing.py:17-43); same process, same
weights: publishing = bumping the version counter (see algo/learning/learner.py)."""
from sample_factory_amd.algo.learning.learner import ParameterServer  # noqa: F401

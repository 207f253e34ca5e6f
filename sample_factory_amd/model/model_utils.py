"""sample_factory/model/model_utils.py:11-24 under its reference path."""
from sample_factory_amd.model.actor_critic import ACT_KIND, get_rnn_size  # noqa: F401

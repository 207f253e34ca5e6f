"""sample_factory/utils/algo_version.py: the experiment-format version stamped into launcher / wandb run names by user
scripts (`from sample_factory.utils.algo_version import ALGO_VERSION`)."""
ALGO_VERSION = 2

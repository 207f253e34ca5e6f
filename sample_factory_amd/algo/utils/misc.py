"""Message keys and experiment status codes under the reference's module path (sample_factory/algo/utils/misc.py:6-33):
the key STRINGS and the status VALUES are the plugin API (message handlers, `run_rl`'s return value)."""

EPS = 1e-8

# keys of the report dictionaries passed to Runner message handlers
(EPISODIC, LEARNER_ENV_STEPS, TRAIN_STATS, TIMING_STATS, STATS_KEY, SAMPLES_COLLECTED, POLICY_ID_KEY) = (
    "episodic", "learner_env_steps", "train", "timing", "stats", "samples_collected", "policy_id")

# fill values the reference's tests look for in uninitialised buffers
MAGIC_FLOAT, MAGIC_INT = -4242.42, 43


class ExperimentStatus:
    """what Runner.init() / Runner.run() / run_rl() return"""
    SUCCESS = 0
    FAILURE = 1
    INTERRUPTED = 2

"""Message keys and status codes — sample_factory/algo/utils/misc.py:7-33 (same names, same values)."""
EPISODIC = "episodic"
LEARNER_ENV_STEPS = "learner_env_steps"
TRAIN_STATS = "train"
STATS_KEY = "stats"
POLICY_ID_KEY = "policy_id"
SAMPLES_COLLECTED = "samples_collected"
TIMING_STATS = "timing"


class ExperimentStatus:
    SUCCESS, FAILURE, INTERRUPTED = range(3)

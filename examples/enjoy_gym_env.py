"""Evaluate the policy examples/train_gym_env.py trained (the reference's counterpart: sf_examples/enjoy_gym_env.py): same
component registration, the evaluation flag set of the argument parser, `sample_factory.enjoy.enjoy(cfg)`.

  python examples/enjoy_gym_env.py --env=CartPole-v1 --experiment=example_gym_cartpole-v1 --max_num_episodes=100 \
      --eval_deterministic=True
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), HERE]

import train_gym_env as training_script  # noqa: E402
from sample_factory.enjoy import enjoy  # noqa: E402

if __name__ == "__main__":
    training_script.register_custom_components()
    status, mean_return = enjoy(training_script.parse_custom_args(evaluation=True))
    print(f"Avg episode reward: {mean_return:.3f}")
    sys.exit(status)

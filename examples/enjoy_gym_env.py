"""Evaluate what examples/train_gym_env.py trained — the shape of the reference's sf_examples/enjoy_gym_env.py:
register the same components, parse the evaluation flags, call enjoy(cfg).

  python examples/enjoy_gym_env.py --env=CartPole-v1 --experiment=example_gym_cartpole-v1 --max_num_episodes=100 \
      --eval_deterministic=True
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from sample_factory.enjoy import enjoy  # noqa: E402
from train_gym_env import parse_custom_args, register_custom_components  # noqa: E402


def main():
    """Script entry point."""
    register_custom_components()
    cfg = parse_custom_args(evaluation=True)
    status, avg_reward = enjoy(cfg)
    print(f"Avg episode reward: {avg_reward:.3f}")
    return status


if __name__ == "__main__":
    sys.exit(main())

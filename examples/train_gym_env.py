"""A training script written against the `sample_factory` import surface — the shape of the reference's
sf_examples/train_gym_env.py:14-47 (BASELINE.json configs[0]: CartPole-v1, serial mode) — running on the MI355X-native
engine.  gymnasium is not installed on the boxes, so the env factory returns the bundled vectorised CartPole instead of
gym.make(); everything else is what a Sample Factory user writes.

  python examples/train_gym_env.py --env=CartPole-v1 --use_rnn=False --serial_mode=True --async_rl=False \
      --num_workers=1 --num_envs_per_worker=1 --worker_num_splits=1 --batch_size=512 --rollout=32 \
      --train_for_env_steps=20000 --experiment=example_gym_cartpole-v1
"""
import os
import sys
from typing import Optional

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from sample_factory.cfg.arguments import parse_full_cfg, parse_sf_args  # noqa: E402
from sample_factory.envs.env_utils import register_env  # noqa: E402
from sample_factory.train import run_rl  # noqa: E402


def make_gym_env_func(full_env_name, cfg=None, env_config=None, render_mode: Optional[str] = None):
    """sf_examples/train_gym_env.py:14-16: `return gym.make(full_env_name, render_mode=render_mode)`.  Where gymnasium is
    installed that is what runs; on the boxes (no gymnasium) --env_agents=0 gives the bundled single cart-pole with the same
    single-env API, and --env_agents=N >= 1 (default 16) the bundled VECTORISED cart-pole (one batched host env)."""
    n = int(getattr(cfg, "env_agents", 16))
    if n <= 0:
        try:
            import gymnasium as gym
            return gym.make(full_env_name, render_mode=render_mode)
        except ImportError:
            from sample_factory.envs.cartpole import CartPoleEnv
            return CartPoleEnv(seed=(cfg.seed or 0) + int(getattr(env_config, "env_id", 0) or 0), render_mode=render_mode)
    from sample_factory.envs.cartpole import CartPoleVecEnv
    return CartPoleVecEnv(num_agents=n, seed=(cfg.seed or 0) if cfg is not None else 0)


def register_custom_components():
    register_env("CartPole-v1", make_gym_env_func)


def parse_custom_args(argv=None, evaluation=False):
    parser, cfg = parse_sf_args(argv=argv, evaluation=evaluation)
    parser.add_argument("--env_agents", default=16, type=int,
                        help="number of CartPole copies in the bundled vector env; 0 = ONE gym-style env per instance "
                             "(gym.make where gymnasium is installed) — BASELINE configs[0]: --num_envs_per_worker=2")
    cfg = parse_full_cfg(parser, argv)
    return cfg


def main():
    """Script entry point."""
    register_custom_components()
    cfg = parse_custom_args()
    status = run_rl(cfg)
    return status


if __name__ == "__main__":
    sys.exit(main())

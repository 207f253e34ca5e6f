"""BASELINE.json configs[0] on the MI355X-native engine: a CartPole training script that touches ONLY the
`sample_factory.*` import surface a Sample Factory user programs against (env registration, argument parsing, run_rl) —
compare the reference's sf_examples/train_gym_env.py:14-47.  `import sample_factory` resolves to this repository's engine.

gymnasium is not installed on the GPU boxes, so the registered factory has three outcomes:
  --env_agents=N (N >= 1, default 16)  the bundled VECTORISED cart-pole: one batched host env of N copies;
  --env_agents=0 with gymnasium        gymnasium's own CartPole-v1 (one single-agent env per instance, as in the reference);
  --env_agents=0 without gymnasium     the bundled single cart-pole with the same single-env gymnasium API.

  python examples/train_gym_env.py --env=CartPole-v1 --use_rnn=False --serial_mode=True --async_rl=False \
      --num_workers=1 --num_envs_per_worker=2 --env_agents=0 --batch_size=512 --rollout=32 \
      --train_for_env_steps=20000 --experiment=example_gym_cartpole-v1
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from sample_factory.cfg.arguments import parse_full_cfg, parse_sf_args  # noqa: E402
from sample_factory.envs.env_utils import register_env  # noqa: E402
from sample_factory.train import run_rl  # noqa: E402

ENV_NAME = "CartPole-v1"


def _seed_of(cfg, env_config) -> int:
    base = getattr(cfg, "seed", None) or 0
    return int(base) + int(getattr(env_config, "env_id", 0) or 0)


def cartpole_factory(full_env_name, cfg=None, env_config=None, render_mode=None):
    """the CreateEnvFunc registered under CartPole-v1 (signature: sample_factory/utils/typing.py:26)"""
    copies = int(getattr(cfg, "env_agents", 16))
    if copies >= 1:
        from sample_factory.envs.cartpole import CartPoleVecEnv
        return CartPoleVecEnv(num_agents=copies, seed=int(getattr(cfg, "seed", None) or 0))
    try:
        import gymnasium
    except ImportError:
        from sample_factory.envs.cartpole import CartPoleEnv
        return CartPoleEnv(seed=_seed_of(cfg, env_config), render_mode=render_mode)
    return gymnasium.make(full_env_name, render_mode=render_mode)


def register_custom_components() -> None:
    register_env(ENV_NAME, cartpole_factory)


def parse_custom_args(argv=None, evaluation: bool = False):
    """the reference's two-stage parse (cfg/arguments.py:24-62) with this script's one extra flag in between"""
    parser, _partial = parse_sf_args(argv=argv, evaluation=evaluation)
    parser.add_argument("--env_agents", type=int, default=16,
                        help="copies of the cart-pole inside the bundled vector env; 0 = one single-agent env per instance "
                             "(BASELINE configs[0]: --num_envs_per_worker=2 --env_agents=0)")
    return parse_full_cfg(parser, argv)


if __name__ == "__main__":
    register_custom_components()
    sys.exit(run_rl(parse_custom_args()))

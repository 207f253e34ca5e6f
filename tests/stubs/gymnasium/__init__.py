"""TEST-SIDE stand-in for the `gymnasium` package (not installable here: no network).

The reference's example scripts start with `import gymnasium as gym`; to execute them BYTE-FOR-BYTE UNMODIFIED against
this engine (tests/test_gpu_reference_scripts.py) the tests put this directory on PYTHONPATH.  It maps the handful of
names those scripts touch — `gym.Env`, `gym.spaces.{Box, Discrete, Dict, Tuple}`, `gym.make("CartPole-v1")` — onto the
engine's own space descriptors and bundled cart-pole.  It is test infrastructure: nothing under sample_factory_amd/
imports it, and with a real gymnasium installed it is simply not on the path.
"""
from . import spaces  # noqa: F401
from .spaces import Space  # noqa: F401

__version__ = "0.0-test-stub"


class Env:
    """gymnasium.Env: attribute defaults only (the scripts' envs define reset / step / render themselves)"""
    metadata: dict = {"render_modes": []}
    render_mode = None
    reward_range = (-float("inf"), float("inf"))
    spec = None
    observation_space = None
    action_space = None

    @property
    def unwrapped(self):
        return self

    def reset(self, *, seed=None, options=None):
        return None, {}

    def close(self):
        pass


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.observation_space, self.action_space = env.observation_space, env.action_space

    @property
    def unwrapped(self):
        return self.env.unwrapped

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.env, name)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def step(self, action):
        return self.env.step(action)

    def close(self):
        return self.env.close()


def make(env_id, render_mode=None, **kwargs):
    if env_id == "CartPole-v1":
        from sample_factory_amd.envs.cartpole import CartPoleEnv
        return CartPoleEnv(render_mode=render_mode, **kwargs)
    raise ValueError(f"test stub of gymnasium.make knows CartPole-v1 only, not {env_id}")

"""Vectorised CartPole on the host (numpy) — BASELINE.json configs[0] ("CartPole-v1 ... CPU only (plumbing)").

gymnasium is not installed on the boxes, so this is a small re-statement of the classic cart-pole dynamics (Barto,
Sutton & Anderson 1983; same constants and termination rules as gymnasium's CartPole-v1: force 10 N, tau 0.02 s,
|x| > 2.4 or |theta| > 12 deg terminates, 500-step time limit -> truncated) with N independent copies and auto-reset.
It is a CPU env: observations arrive as host arrays and go through the runner's generic ingest path (copy into the
slab), actions come back as a device tensor -> exercises the non-zero-copy side of the rollout runner.
"""
from __future__ import annotations

import numpy as np

from sample_factory_amd.envs import spaces


class CartPoleVecEnv:
    def __init__(self, num_agents=2, seed=0, max_steps=500):
        self.num_agents = int(num_agents)
        self.observation_space = spaces.Dict({"obs": spaces.Box(-np.inf, np.inf, (4,), np.float32)})
        self.action_space = spaces.Discrete(2)
        self.rng = np.random.default_rng(seed)
        self.max_steps = max_steps
        self.state = np.zeros((self.num_agents, 4), np.float64)
        self.steps = np.zeros(self.num_agents, np.int64)

    def _reset_rows(self, rows):
        self.state[rows] = self.rng.uniform(-0.05, 0.05, (int(rows.sum()) if rows.dtype == bool else len(rows), 4))
        self.steps[rows] = 0

    def reset(self, **kwargs):
        self._reset_rows(np.ones(self.num_agents, bool))
        return {"obs": self.state.astype(np.float32)}, {}

    def step(self, actions):
        a = np.asarray(actions.cpu() if hasattr(actions, "cpu") else actions).reshape(-1).astype(np.int64)
        x, x_dot, th, th_dot = self.state.T
        force = np.where(a == 1, 10.0, -10.0)
        g, mc, mp, length = 9.8, 1.0, 0.1, 0.5
        total, pml = mc + mp, mp * length
        cos, sin = np.cos(th), np.sin(th)
        temp = (force + pml * th_dot ** 2 * sin) / total
        th_acc = (g * sin - cos * temp) / (length * (4.0 / 3.0 - mp * cos ** 2 / total))
        x_acc = temp - pml * th_acc * cos / total
        tau = 0.02
        self.state = np.stack([x + tau * x_dot, x_dot + tau * x_acc, th + tau * th_dot, th_dot + tau * th_acc], 1)
        self.steps += 1
        terminated = (np.abs(self.state[:, 0]) > 2.4) | (np.abs(self.state[:, 2]) > 12 * 2 * np.pi / 360)
        truncated = (self.steps >= self.max_steps) & ~terminated
        rew = np.ones(self.num_agents, np.float32)
        done = terminated | truncated
        if done.any():
            self._reset_rows(done)
        return {"obs": self.state.astype(np.float32)}, rew, terminated, truncated, {}

    def close(self):
        pass


class CartPoleEnv:
    """ONE cart-pole with the gymnasium single-env API (reset(seed=...) -> (obs, info); step(a) -> (obs, reward,
    terminated, truncated, info); no auto-reset) — what `gym.make("CartPole-v1")` returns, for boxes without gymnasium.
    The engine wraps such envs as the reference does (make_env.py:97-128): one agent each, auto-reset on done, stepped in
    this process (serial_mode) or in env worker processes."""

    def __init__(self, seed=0, max_steps=500, render_mode=None):
        self._v = CartPoleVecEnv(num_agents=1, seed=seed, max_steps=max_steps)
        self.observation_space = spaces.Box(-np.inf, np.inf, (4,), np.float32)
        self.action_space = spaces.Discrete(2)
        self.render_mode = render_mode

    def reset(self, seed=None, options=None):
        if seed is not None:
            self._v.rng = np.random.default_rng(seed)
        o, info = self._v.reset()
        return o["obs"][0], info

    def step(self, action):
        v = self._v
        a = np.asarray([action]).reshape(-1).astype(np.int64)
        x, x_dot, th, th_dot = v.state.T
        force = np.where(a == 1, 10.0, -10.0)
        g, mc, mp, length = 9.8, 1.0, 0.1, 0.5
        total, pml = mc + mp, mp * length
        cos, sin = np.cos(th), np.sin(th)
        temp = (force + pml * th_dot ** 2 * sin) / total
        th_acc = (g * sin - cos * temp) / (length * (4.0 / 3.0 - mp * cos ** 2 / total))
        x_acc = temp - pml * th_acc * cos / total
        tau = 0.02
        v.state = np.stack([x + tau * x_dot, x_dot + tau * x_acc, th + tau * th_dot, th_dot + tau * th_acc], 1)
        v.steps += 1
        terminated = bool((np.abs(v.state[0, 0]) > 2.4) | (np.abs(v.state[0, 2]) > 12 * 2 * np.pi / 360))
        truncated = bool(v.steps[0] >= v.max_steps) and not terminated
        return v.state[0].astype(np.float32), 1.0, terminated, truncated, {}

    def close(self):
        pass


def make_cartpole_env(full_env_name, cfg=None, env_config=None, render_mode=None):
    n = getattr(cfg, "cartpole_num_agents", 2) if cfg is not None else 2
    seed = ((getattr(cfg, "seed", None) or 0) if cfg is not None else 0) + 1000 * int(getattr(env_config, "env_id", 0) or 0)
    return CartPoleVecEnv(num_agents=n, seed=seed)

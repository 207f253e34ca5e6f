"""Synthetic device-resident vector env (SURVEY.md §8d "C2 synthetic inputs"): N agents, u8 [C,H,W] frames from a
counter-based Philox4x32-10 stream, reward 1 when action == (step + env_id) % num_actions, Bernoulli(1/1024)
termination.  To Sample Factory it looks like the reference's GPU envs (brax / isaacgym: `num_agents=N`, tensors on
the device, auto-reset; sf_examples/brax/train_brax.py:160-204).  Additionally it implements the zero-copy hook
`step_into(actions, obs_out)`: the frame generator writes straight into slot t+1 of the trajectory slab (K1).
"""
from __future__ import annotations

import numpy as np
import torch

from sample_factory_amd import lib
from sample_factory_amd.envs import spaces


class SyntheticVecEnv:
    def __init__(self, num_agents=4096, obs_shape=(4, 84, 84), num_actions=6, seed=0, env0=0, device="cuda"):
        self.num_agents = int(num_agents)
        self.obs_shape = tuple(obs_shape)
        self.obs_bytes = int(np.prod(obs_shape))
        assert self.obs_bytes % 16 == 0, "frame size must be a multiple of 16 bytes"
        self.observation_space = spaces.Dict({"obs": spaces.Box(0, 255, self.obs_shape, np.uint8)})
        self.action_space = spaces.Discrete(num_actions)
        self.num_actions = num_actions
        self.seed_, self.env0 = int(seed), int(env0)
        self.device = torch.device(device)
        self.step_count = 0
        self._rew = torch.zeros(self.num_agents, dtype=torch.float32, device=self.device)
        self._term = torch.zeros(self.num_agents, dtype=torch.bool, device=self.device)
        self._trunc = torch.zeros(self.num_agents, dtype=torch.bool, device=self.device)
        self._obs = None

    # -- zero-copy interface used by the native rollout runner
    def write_obs(self, obs_out: torch.Tensor) -> None:
        """obs_out: [N, C, H, W] u8 view (possibly strided over dim 0, e.g. slab[:, t])."""
        assert obs_out.dtype == torch.uint8 and obs_out.is_cuda and obs_out.shape[0] == self.num_agents
        assert obs_out[0].is_contiguous()
        lib.synth_obs(obs_out.data_ptr(), obs_out.stride(0), self.num_agents, self.env0, self.obs_bytes, self.seed_,
                      self.step_count)

    def reset_into(self, obs_out: torch.Tensor) -> None:
        self.step_count = 0
        self.write_obs(obs_out)

    def step_into(self, actions: torch.Tensor, obs_out: torch.Tensor):
        """actions int32 [N] on device. Returns (rewards f32, terminated bool, truncated bool) device tensors."""
        lib.synth_step(actions, self.env0, self.num_actions, self.seed_, self.step_count, self._rew, self._term)
        self.step_count += 1
        self.write_obs(obs_out)
        return self._rew, self._term, self._trunc

    # -- gymnasium-style interface (what a stock Sample Factory runner would call; make_env.py:147-237)
    def reset(self, **kwargs):
        if self._obs is None:
            self._obs = torch.empty((self.num_agents,) + self.obs_shape, dtype=torch.uint8, device=self.device)
        self.reset_into(self._obs)
        return {"obs": self._obs}, {}

    def step(self, actions):
        if self._obs is None:
            self.reset()
        a = torch.as_tensor(actions, device=self.device).to(torch.int32).reshape(-1).contiguous()
        rew, term, trunc = self.step_into(a, self._obs)
        return {"obs": self._obs}, rew, term, trunc, {}

    def close(self):
        pass


class SyntheticContinuousEnv:
    """Ant-shaped device-resident env (SURVEY.md §8d C5 stand-in): f32 vector observations, Box actions.  One HIP
    launch per step (sf_synth_vec_step, counter-based Philox noise keyed by the global env id): obs' = 0.9*obs +
    0.1*noise, reward = -mean(action^2) + 0.1*obs[:,0], termination ~ Bernoulli(1/256) with auto-reset.  Like the image
    env it offers the zero-copy hooks reset_into / step_into (the next observation lands in slot t+1 of the slab) next to
    the gymnasium-style reset / step."""

    def __init__(self, num_agents=2048, obs_dim=27, act_dim=8, seed=0, env0=0, device="cuda"):
        self.num_agents, self.obs_dim, self.act_dim = int(num_agents), int(obs_dim), int(act_dim)
        self.observation_space = spaces.Dict({"obs": spaces.Box(-np.inf, np.inf, (obs_dim,), np.float32)})
        self.action_space = spaces.Box(-1.0, 1.0, (act_dim,), np.float32)
        self.device = torch.device(device)
        self.seed_, self.env0, self.step_count = int(seed), int(env0), 0
        self.state = torch.zeros((self.num_agents, obs_dim), dtype=torch.float32, device=self.device)
        self._obs = torch.zeros_like(self.state)
        self._rew = torch.zeros(self.num_agents, dtype=torch.float32, device=self.device)
        self._term = torch.zeros(self.num_agents, dtype=torch.bool, device=self.device)
        self._trunc = torch.zeros(self.num_agents, dtype=torch.bool, device=self.device)

    def reset_into(self, obs_out: torch.Tensor) -> None:
        self.step_count = 0
        lib.synth_vec_step(self.state, None, obs_out, self.env0, self.seed_, 0xFFFFFFFF, True, self._rew, self._term)

    def step_into(self, actions: torch.Tensor, obs_out: torch.Tensor):
        """actions f32 [N, act_dim] device view (e.g. traj.actions[:, t]); returns (rewards, terminated, truncated)"""
        lib.synth_vec_step(self.state, actions, obs_out, self.env0, self.seed_, self.step_count, False, self._rew, self._term)
        self.step_count += 1
        return self._rew, self._term, self._trunc

    def reset(self, **kwargs):
        self.reset_into(self._obs)
        return {"obs": self._obs}, {}

    def step(self, actions):
        a = torch.as_tensor(actions, device=self.device, dtype=torch.float32).reshape(self.num_agents, self.act_dim).contiguous()
        rew, term, trunc = self.step_into(a, self._obs)
        return {"obs": self._obs}, rew, term, trunc, {}

    def close(self):
        pass


class HostFrameVecEnv:
    """Atari-shaped HOST vector env (BASELINE configs[2] stand-in: envpool is not installed on the boxes): N agents,
    u8 [4,84,84] frames living in ordinary host memory, returned as numpy arrays exactly as envpool hands over its own
    buffers (batched_sampling.py:62-82) — so every step costs the real ingest path: host array -> pinned staging ->
    pitched H2D DMA into the slab, and a D2H of the int32 actions.  Frames come from a small pre-generated ring (the
    simulator's own cost is not what is being measured); rewards follow the synthetic env's rule."""

    def __init__(self, num_agents=1024, obs_shape=(4, 84, 84), num_actions=6, seed=0, ring=4, sim_ms=0.0):
        self.num_agents, self.obs_shape, self.num_actions = int(num_agents), tuple(obs_shape), int(num_actions)
        self.observation_space = spaces.Dict({"obs": spaces.Box(0, 255, self.obs_shape, np.uint8)})
        self.action_space = spaces.Discrete(num_actions)
        rng = np.random.default_rng(seed)
        self.ring = rng.integers(0, 256, size=(ring, self.num_agents) + self.obs_shape, dtype=np.uint8)
        self.step_count = 0
        self.sim_ms = float(sim_ms)  # optional simulated emulator time per step (host sleep)
        self._ids = np.arange(self.num_agents)
        self._rng = rng

    def reset(self, **kwargs):
        self.step_count = 0
        return {"obs": self.ring[0]}, {}

    def step(self, actions):
        a = np.asarray(actions).reshape(-1)
        rew = (a == (self.step_count + self._ids) % self.num_actions).astype(np.float32)
        term = self._rng.random(self.num_agents) < (1.0 / 1024.0)
        self.step_count += 1
        if self.sim_ms > 0:
            import time
            time.sleep(self.sim_ms * 1e-3)
        return {"obs": self.ring[self.step_count % len(self.ring)]}, rew, term, np.zeros(self.num_agents, bool), {}

    def close(self):
        pass


def make_host_frame_env(full_env_name, cfg=None, env_config=None, render_mode=None):
    n = getattr(cfg, "synthetic_num_agents", 1024) if cfg is not None else 1024
    seed = ((getattr(cfg, "seed", None) or 0) if cfg is not None else 0) + int(getattr(env_config, "env_id", 0) or 0)
    return HostFrameVecEnv(num_agents=n, seed=seed, sim_ms=getattr(cfg, "host_env_sim_ms", 0.0) if cfg is not None else 0.0)


class SyntheticTupleEnv(SyntheticVecEnv):
    """Multi-head variant (Tuple(Discrete(n0), Discrete(n1), ...), the VizDoom-style action space of
    action_distributions.py:197-287): same frames; the reward rule looks at head 0, the other heads are free.
    A negative entry -D of `head_sizes` is a Box(-1, 1, (D,)) member (a mixed Tuple: the env then receives the list of
    per-member arrays `preprocess_actions` builds, batched_sampling.py:51-59)."""

    def __init__(self, head_sizes=(6, 3), **kw):
        assert int(head_sizes[0]) > 0, "member 0 carries the reward rule: Discrete"
        super().__init__(num_actions=int(head_sizes[0]), **kw)
        self.action_space = spaces.Tuple([spaces.Discrete(int(n)) if int(n) > 0 else
                                          spaces.Box(-1.0, 1.0, (-int(n),), np.float32) for n in head_sizes])
        self.last_actions = None  # what the sampler handed over at the last step (tests look at the format)

    def _head0(self, actions):
        self.last_actions = actions
        if isinstance(actions, (list, tuple)):  # mixed Tuple: one array per member
            return torch.as_tensor(actions[0], device=self.device).to(torch.int32).reshape(self.num_agents).contiguous()
        return torch.as_tensor(actions, device=self.device).to(torch.int32).reshape(self.num_agents, -1)[:, 0].contiguous()

    def step_into(self, actions, obs_out: torch.Tensor):
        return super().step_into(self._head0(actions), obs_out)

    def step(self, actions):
        return super().step(self._head0(actions))


class MaskedBanditEnv:
    """GPU vector env with an action mask in the observation dict (obs["action_mask"], u8 [N, A]): a contextual bandit
    whose rewarded action (obs one-hot) is always allowed while a random subset of the others is masked out."""

    def __init__(self, num_agents=64, num_actions=6, seed=0, device="cuda"):
        self.num_agents, self.A = int(num_agents), int(num_actions)
        self.observation_space = spaces.Dict({"obs": spaces.Box(0, 1, (self.A,), np.float32),
                                              "action_mask": spaces.Box(0, 1, (self.A,), np.uint8)})
        self.action_space = spaces.Discrete(self.A)
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed))
        self._draw()

    def _draw(self):
        self.target = torch.randint(0, self.A, (self.num_agents,), generator=self.gen, device=self.device)
        self.mask = (torch.rand((self.num_agents, self.A), generator=self.gen, device=self.device) < 0.5)
        self.mask[torch.arange(self.num_agents, device=self.device), self.target] = True
        self.obs = torch.nn.functional.one_hot(self.target, self.A).float()

    def _out(self):
        return {"obs": self.obs, "action_mask": self.mask.to(torch.uint8)}

    def reset(self, **kwargs):
        self._draw()
        return self._out(), {}

    def step(self, actions):
        a = torch.as_tensor(actions, device=self.device).reshape(-1).long()
        self.last_allowed = self.mask[torch.arange(self.num_agents, device=self.device), a].clone()
        rew = (a == self.target).float()
        term = torch.ones(self.num_agents, dtype=torch.bool, device=self.device)  # one-step episodes, auto-reset
        self._draw()
        return self._out(), rew, term, torch.zeros_like(term), {}

    def close(self):
        pass


class DictObsBanditEnv:
    """GPU vector env whose observation is a dict of SEVERAL keys (the reference's MultiInputEncoder case,
    model/encoder.py:33-69): "obs" = a random u8 image (pure distraction), "measurements" = one-hot of the rewarded
    action.  The policy can only learn the bandit through the vector key."""

    def __init__(self, num_agents=64, num_actions=4, image_shape=(4, 36, 36), seed=0, device="cuda"):
        self.num_agents, self.A, self.image_shape = int(num_agents), int(num_actions), tuple(image_shape)
        self.observation_space = spaces.Dict({"obs": spaces.Box(0, 255, self.image_shape, np.uint8),
                                              "measurements": spaces.Box(0, 1, (self.A,), np.float32)})
        self.action_space = spaces.Discrete(self.A)
        self.device = torch.device(device)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed))
        self._draw()

    def _draw(self):
        self.target = torch.randint(0, self.A, (self.num_agents,), generator=self.gen, device=self.device)
        self.img = torch.randint(0, 256, (self.num_agents,) + self.image_shape, generator=self.gen, device=self.device,
                                 dtype=torch.int32).to(torch.uint8)

    def _out(self):
        return {"obs": self.img, "measurements": torch.nn.functional.one_hot(self.target, self.A).float()}

    def reset(self, **kwargs):
        self._draw()
        return self._out(), {}

    def step(self, actions):
        a = torch.as_tensor(actions, device=self.device).reshape(-1).long()
        rew = (a == self.target).float()
        term = torch.ones(self.num_agents, dtype=torch.bool, device=self.device)  # one-step episodes, auto-reset
        self._draw()
        return self._out(), rew, term, torch.zeros_like(term), {}

    def close(self):
        pass


def make_dict_obs_bandit_env(full_env_name, cfg=None, env_config=None, render_mode=None):
    n = getattr(cfg, "synthetic_num_agents", 64) if cfg is not None else 64
    return DictObsBanditEnv(num_agents=n, seed=(getattr(cfg, "seed", None) or 0) if cfg is not None else 0)


def make_masked_bandit_env(full_env_name, cfg=None, env_config=None, render_mode=None):
    n = getattr(cfg, "synthetic_num_agents", 64) if cfg is not None else 64
    return MaskedBanditEnv(num_agents=n, seed=(getattr(cfg, "seed", None) or 0) if cfg is not None else 0)


def make_synthetic_tuple_env(full_env_name, cfg=None, env_config=None, render_mode=None):
    n = getattr(cfg, "synthetic_num_agents", 4096) if cfg is not None else 4096
    seed = (getattr(cfg, "seed", None) or 0) if cfg is not None else 0
    return SyntheticTupleEnv(head_sizes=getattr(cfg, "synthetic_head_sizes", (6, 3)) if cfg is not None else (6, 3),
                             num_agents=n, seed=seed)


def make_synthetic_continuous_env(full_env_name, cfg=None, env_config=None, render_mode=None):
    n = getattr(cfg, "synthetic_num_agents", 2048) if cfg is not None else 2048
    env0 = (getattr(cfg, "synthetic_env0", 0) or 0) if cfg is not None else 0
    env0 += int(getattr(env_config, "env_id", 0) or 0) * n if env_config is not None else 0
    return SyntheticContinuousEnv(num_agents=n, seed=(getattr(cfg, "seed", None) or 0) if cfg is not None else 0, env0=env0)


def make_synthetic_env(full_env_name, cfg=None, env_config=None, render_mode=None):
    n = getattr(cfg, "synthetic_num_agents", 4096) if cfg is not None else 4096
    seed = (getattr(cfg, "seed", None) or 0) if cfg is not None else 0
    env0 = getattr(cfg, "synthetic_env0", 0) if cfg is not None else 0
    # instance e of a multi-instance run simulates global envs [env0 + e*n, env0 + (e+1)*n): the union over instances
    # is the env set of ONE instance with E*n agents (bit-identical rollouts, tests/test_gpu_nn.py)
    env0 += int(getattr(env_config, "env_id", 0) or 0) * n if env_config is not None else 0
    return SyntheticVecEnv(num_agents=n, seed=seed, env0=env0)

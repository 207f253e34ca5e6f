"""Multi-process host env stepping (sample_factory_amd/algo/sampling/parallel_env.py) — the reference's rollout-worker
processes (rollout_worker.py:79-308) — on the CPU: worker processes must produce exactly what the same env instances
produce when stepped in this process, in slab row order, for batched AND single-agent envs, one or two splits."""
import numpy as np
import pytest

from sample_factory_amd.cfg.arguments import default_cfg
from sample_factory_amd.envs import spaces
from sample_factory_amd.envs.cartpole import make_cartpole_env
from sample_factory_amd.utils.attr_dict import AttrDict


class CountingEnv:
    """single-agent gym-style env (no num_agents): obs = [env_id, episode, step, last action], reward = action, an episode
    lasts 3 + env_id % 3 steps (terminated) — deterministic, so the auto-reset of the worker wrapper is checkable"""

    def __init__(self, env_id):
        self.env_id, self.episode, self.t, self.last = int(env_id), -1, 0, 0
        self.observation_space = spaces.Box(-1e6, 1e6, (4,), np.float32)
        self.action_space = spaces.Discrete(5)

    def _obs(self):
        return np.array([self.env_id, self.episode, self.t, self.last], np.float32)

    def reset(self, **kw):
        self.episode += 1
        self.t, self.last = 0, 0
        return self._obs(), {}

    def step(self, action):
        assert isinstance(action, int), type(action)  # the action axis is squeezed for a single-agent Discrete env
        self.t += 1
        self.last = action
        term = self.t >= 3 + self.env_id % 3
        return self._obs(), float(action), term, False, {}

    def close(self):
        pass


def make_counting_env(full_env_name, cfg=None, env_config=None, render_mode=None):
    return CountingEnv(env_config.env_id)


def _serial_reference(make, cfg, W, K, S, steps, actions):
    """the same instances stepped here, rows ordered (split; worker, instance-in-split, agent)"""
    from sample_factory_amd.algo.sampling.parallel_env import env_is_batched
    per = K // S
    out = []
    for split in range(S):
        envs = []
        for w in range(W):
            for j in range(per):
                v = split * per + j
                envs.append(make("x", cfg, AttrDict(worker_index=w, vector_index=v, env_id=w * K + v), None))
        batched = env_is_batched(envs[0])
        n_inst = getattr(envs[0], "num_agents", 1) if batched else 1
        rows = []

        def obs_of(e, reset=False, a=None):
            if reset:
                try:
                    o, _ = e.reset(seed=0)
                except TypeError:
                    o, _ = e.reset()
                return o
            return None

        cur = []
        for i, e in enumerate(envs):
            try:
                o, _ = e.reset(seed=i)
            except TypeError:
                o, _ = e.reset()
            cur.append(o["obs"] if isinstance(o, dict) else o)
        hist = [np.concatenate([np.asarray(c).reshape(n_inst, -1) for c in cur])]
        rews, terms = [], []
        for t in range(steps):
            cur, rr, tt = [], [], []
            for i, e in enumerate(envs):
                a = actions[split][t][i * n_inst:(i + 1) * n_inst]
                if batched:
                    o, r, te, tr, _ = e.step(a)
                else:
                    o, r, te, tr, _ = e.step(int(a[0]))
                    if te or tr:
                        o, _ = e.reset()
                cur.append(o["obs"] if isinstance(o, dict) else o)
                rr.append(np.asarray(r, np.float32).reshape(-1))
                tt.append(np.asarray(te).reshape(-1))
            hist.append(np.concatenate([np.asarray(c).reshape(n_inst, -1) for c in cur]))
            rews.append(np.concatenate(rr))
            terms.append(np.concatenate(tt))
        out.append((hist, rews, terms))
    return out


@pytest.mark.parametrize("kind,W,K,S", [("cartpole", 2, 2, 1), ("cartpole", 2, 4, 2), ("counting", 3, 2, 1), ("counting", 2, 2, 2)])
def test_worker_processes_equal_in_process_stepping(kind, W, K, S):
    from sample_factory_amd.algo.sampling.parallel_env import ParallelHostEnvs
    cfg = default_cfg(env=kind, seed=3, cartpole_num_agents=3)
    make = make_cartpole_env if kind == "cartpole" else make_counting_env
    penv = ParallelHostEnvs(cfg, kind, make, W, K, num_splits=S)
    try:
        assert len(penv.views) == S
        n = penv.views[0].num_agents
        assert n == W * (K // S) * (3 if kind == "cartpole" else 1)
        steps, A = 12, 2 if kind == "cartpole" else 5
        rng = np.random.default_rng(0)
        actions = [[rng.integers(0, A, n).astype(np.int32) for _ in range(steps)] for _ in range(S)]
        want = _serial_reference(make, cfg, W, K, S, steps, actions)
        got = [([], [], []) for _ in range(S)]
        for s, v in enumerate(penv.views):
            o, _ = v.reset()
            got[s][0].append(o["obs"].reshape(n, -1).copy())
        for t in range(steps):  # both splits in flight at once (double-buffered sampling)
            for s, v in enumerate(penv.views):
                v.step_async(actions[s][t])
            for s, v in enumerate(penv.views):
                o, r, te, tr, _ = v.step_wait()
                got[s][0].append(o["obs"].reshape(n, -1).copy())
                got[s][1].append(r.copy())
                got[s][2].append(te.copy())
        for s in range(S):
            for t in range(steps + 1):
                np.testing.assert_array_equal(got[s][0][t], want[s][0][t].astype(np.float32), err_msg=f"obs split {s} step {t}")
            for t in range(steps):
                np.testing.assert_array_equal(got[s][1][t], want[s][1][t])
                np.testing.assert_array_equal(got[s][2][t], want[s][2][t])
        if kind == "counting":  # auto-reset happened inside the workers: episode counters advanced
            assert got[0][0][-1][:, 1].max() >= 2
    finally:
        penv.close()


def test_worker_failure_is_reported():
    from sample_factory_amd.algo.sampling.parallel_env import ParallelHostEnvs
    cfg = default_cfg(env="boom", seed=0)
    with pytest.raises(RuntimeError, match="failed to create|did not start"):
        ParallelHostEnvs(cfg, "boom", make_failing_env, 1, 1)


class _Probe:
    observation_space = spaces.Box(-1, 1, (2,), np.float32)
    action_space = spaces.Discrete(2)

    def close(self):
        pass


_calls = {"n": 0}


def make_failing_env(full_env_name, cfg=None, env_config=None, render_mode=None):
    import multiprocessing
    if multiprocessing.current_process().name.startswith("sf-env-worker"):
        raise ValueError("no such simulator on this host")
    return _Probe()


def make_single_cartpole(full_env_name, cfg=None, env_config=None, render_mode=None):
    from sample_factory_amd.envs.cartpole import CartPoleEnv
    return CartPoleEnv(seed=3 + env_config.env_id)


def test_inline_mode_wraps_single_agent_gym_envs_like_the_reference():
    """serial_mode: the same wrappers without processes — K single-agent gym-style envs (reset(seed) / step, no auto-reset,
    what gym.make returns) behind one batched view with auto-reset on done (make_env.py:97-128); BASELINE configs[0] is two
    CartPole copies of this kind"""
    from sample_factory_amd.algo.sampling.parallel_env import ParallelHostEnvs
    from sample_factory_amd.envs.cartpole import CartPoleEnv
    cfg = default_cfg(env="c", seed=3)
    penv = ParallelHostEnvs(cfg, "c", make_single_cartpole, 1, 2, num_splits=1, inline=True)
    v = penv.views[0]
    assert v.num_agents == 2 and not penv.register_with_device()
    o, _ = v.reset()
    envs = [CartPoleEnv(seed=3 + i) for i in range(2)]
    np.testing.assert_array_equal(o["obs"], np.stack([e.reset(seed=i)[0] for i, e in enumerate(envs)]))
    rng, resets = np.random.default_rng(0), 0
    for t in range(300):
        a = rng.integers(0, 2, 2).astype(np.int32)
        o, r, te, tr, _ = v.step(a)
        for i, e in enumerate(envs):
            oo, rr, t1, t2, _ = e.step(int(a[i]))
            if t1 or t2:
                oo, _ = e.reset()
                resets += 1
            np.testing.assert_array_equal(o["obs"][i], oo)
            assert te[i] == t1 and tr[i] == t2 and r[i] == rr
    assert resets >= 5
    penv.close()


class _DyingEnv(CountingEnv):
    def step(self, action):
        if self.t >= 2:
            import os
            os._exit(7)  # the simulator takes the whole process down (no exception, no message)
        return super().step(action)


def make_dying_env(full_env_name, cfg=None, env_config=None, render_mode=None):
    import multiprocessing
    return _DyingEnv(env_config.env_id) if multiprocessing.current_process().name.startswith("sf-env-worker") \
        else CountingEnv(env_config.env_id)


def test_a_worker_that_dies_silently_is_noticed():
    from sample_factory_amd.algo.sampling.parallel_env import ParallelHostEnvs
    cfg = default_cfg(env="dying", seed=0)
    penv = ParallelHostEnvs(cfg, "dying", make_dying_env, 1, 1)
    v = penv.views[0]
    v.reset()
    a = np.zeros(1, np.int32)
    v.step(a)
    v.step(a)
    with pytest.raises(RuntimeError, match="exited"):
        v.step(a)


@pytest.mark.parametrize("case", ["rollout_ff_sync", "rollout_tuple_heads", "rollout_box_actions"])
def test_action_format_handed_to_the_env_equals_the_reference(case):
    """`_format_actions` (what a worker hands `env.step`) against what the reference's `preprocess_actions`
    (batched_sampling.py:30-82) handed the scripted env when the fixture was recorded: int32 with the action axis squeezed
    for one Discrete head, the [agents, heads] int32 array for an all-Discrete Tuple, float32 rows for a Box."""
    import os
    from sample_factory_amd.algo.sampling.parallel_env import _format_actions
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", case + ".npz"), allow_pickle=True)
    seen, ref = d["env_seen_actions"], d["ref_actions"]          # [steps, agents(, heads / dims)]
    kind = str(d["action_kind"])
    continuous = kind == "box"
    heads = [7] * ref.shape[2] if (ref.ndim == 3 and not continuous) else [6]
    for t in range(seen.shape[0]):
        rows = ref[t].astype(np.float32 if continuous else np.int32)  # the worker's shared `act` array
        got = _format_actions(rows, heads, continuous, batched=True)
        assert got.dtype == seen.dtype and got.shape == seen[t].shape, (got.dtype, got.shape, seen.dtype, seen[t].shape)
        assert np.array_equal(got, seen[t])
    # a single-agent env gets its agent's row without the agent axis (make_env.py:97-99)
    one = _format_actions(ref[0][:1].astype(np.float32 if continuous else np.int32), heads, continuous, batched=False)
    if continuous:
        assert one.shape == seen[0][0].shape and one.dtype == np.float32
    elif len(heads) > 1:
        assert one.shape == (len(heads),) and np.array_equal(one, seen[0][0])
    else:
        assert isinstance(one, int) and one == int(seen[0][0])


def test_mixed_tuple_members_handed_to_the_env_equal_the_reference():
    """Tuple(Discrete(3), Box(2), Discrete(4)): `_format_actions` hands the env one array per member, the same arrays (dtype,
    shape, values) the reference's `preprocess_actions` (batched_sampling.py:51-59) handed the scripted env when
    tests/golden/rollout_tuple_mixed.npz was recorded."""
    import os
    from sample_factory_amd.algo.sampling.parallel_env import _format_actions
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "rollout_tuple_mixed.npz"), allow_pickle=True)
    heads, ref = [int(v) for v in d["head_sizes"]], d["ref_actions"]       # [steps, agents, action columns] f32
    assert bool(d["env_seen_is_list"]) and heads == [3, -2, 4]
    for t in range(ref.shape[0]):
        got = _format_actions(ref[t], heads, continuous=False, batched=True)
        assert isinstance(got, list) and len(got) == len(heads)
        for i, a in enumerate(got):
            want = d[f"env_seen_member{i}"][t]
            assert a.dtype == want.dtype and a.shape == want.shape, (i, a.dtype, a.shape, want.dtype, want.shape)
            assert np.array_equal(a, want)
    one = _format_actions(ref[0][:1], heads, continuous=False, batched=False)  # single-agent env: no agent axis
    assert isinstance(one[0], int) and one[0] == int(d["env_seen_member0"][0, 0])
    assert one[1].shape == (2,) and np.array_equal(one[1], d["env_seen_member1"][0, 0])


class _HostVecEnv:
    """batched HOST env (num_agents attribute, numpy observations): the reference steps such an env as it is"""
    num_agents = 4

    def __init__(self):
        self.observation_space = spaces.Box(-1.0, 1.0, (3,), np.float32)
        self.action_space = spaces.Discrete(2)

    def reset(self, **kw):
        return np.zeros((4, 3), np.float32), {}

    def step(self, a):
        return np.zeros((4, 3), np.float32), np.zeros(4, np.float32), np.zeros(4, bool), np.zeros(4, bool), {}

    def close(self):
        pass


@pytest.mark.parametrize("env,serial,mode,want", [
    ("plan_single", True, "auto", "inline"),     # BASELINE configs[0]: single-agent gym envs, serial_mode -> this process
    ("plan_single", False, "auto", "process"),   # ... parallel mode -> cfg.num_workers worker processes
    ("plan_vec", True, "auto", "direct"),        # a batched host env in serial mode is stepped as it is
    ("plan_vec", False, "auto", "process"),
    ("plan_vec", False, "inline", "direct"),     # cfg.env_workers_mode overrides the serial_mode rule
    ("plan_single", True, "process", "process"),
])
def test_env_deployment_plan_follows_serial_mode_and_env_kind(env, serial, mode, want):
    """Runner._env_plan (train.py): which of the three deployments a (cfg, env) pair gets — decided from a probe instance,
    on the host, before anything touches the GPU"""
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.train import Runner
    register_env("plan_single", make_counting_env)
    register_env("plan_vec", lambda name, cfg=None, env_config=None, render_mode=None: _HostVecEnv())
    cfg = default_cfg(env=env, serial_mode=serial, env_workers_mode=mode, num_workers=2, num_envs_per_worker=2)
    r = Runner(cfg)
    assert r._env_plan() == want
    assert (getattr(r, "_probe_env", None) is not None) == (want == "direct")  # the probe is reused only as instance 0


class ShapedEnv:
    """single-agent env whose reward is scheme["w"] and which also listens for training info: the optional env interfaces of
    envs/env_utils.py:60-133 must reach instances that live in worker processes"""

    def __init__(self):
        from sample_factory_amd.envs.env_utils import RewardShapingInterface, TrainingInfoInterface

        class _Impl(RewardShapingInterface, TrainingInfoInterface):
            def __init__(s):
                TrainingInfoInterface.__init__(s)
                s.scheme = {"w": 1.0}

            def get_default_reward_shaping(s):
                return {"w": 1.0}

            def set_reward_shaping(s, reward_shaping, agent_idx):
                s.scheme = dict(reward_shaping)

        self.env = _Impl()  # one layer down, as behind a gym wrapper
        self.observation_space = spaces.Box(-1, 1, (2,), np.float32)
        self.action_space = spaces.Discrete(2)

    def reset(self, **kw):
        return np.zeros(2, np.float32), {}

    def step(self, action):
        steps = float(self.env.training_info.get("approx_total_training_steps", 0))
        return np.array([steps, 0], np.float32), float(self.env.scheme["w"]), False, False, {}

    def close(self):
        pass


def make_shaped_env(full_env_name, cfg=None, env_config=None, render_mode=None):
    return ShapedEnv()


@pytest.mark.parametrize("inline", [True, False])
def test_reward_shaping_and_training_info_reach_the_instances(inline):
    from sample_factory_amd.algo.sampling.parallel_env import ParallelHostEnvs
    from sample_factory_amd.envs.env_utils import get_default_reward_shaping, set_reward_shaping
    cfg = default_cfg(env="shaped", seed=0)
    host = ParallelHostEnvs(cfg, "shaped", make_shaped_env, 2, 2, num_splits=2, inline=inline)
    try:
        view = host.views[0]
        assert get_default_reward_shaping(view) == {"w": 1.0}  # env_info.py:52 asks the batched env for it
        for v in host.views:
            v.reset()
        _, rew, *_ = view.step(np.zeros(view.num_agents, np.int32))
        assert np.all(rew == 1.0)
        set_reward_shaping(view, {"w": 3.0}, slice(None))
        view.set_training_info(dict(approx_total_training_steps=500))
        for v in host.views:  # instances of BOTH splits got the messages
            obs, rew, *_ = v.step(np.zeros(v.num_agents, np.int32))
            assert np.all(rew == 3.0) and np.all(obs["obs"][:, 0] == 500.0)
        # a per-agent update (an int index into ONE view's agent axis, env_utils.py:106-111) reaches the instance that owns
        # that row — and only it — with the instance's local index (2 workers x 1 instance per split: rows 0 and 1 of a view)
        set_reward_shaping(host.views[1], {"w": 7.0}, 1)
        _, rew0, *_ = host.views[0].step(np.zeros(2, np.int32))
        _, rew1, *_ = host.views[1].step(np.zeros(2, np.int32))
        assert np.all(rew0 == 3.0) and rew1.tolist() == [3.0, 7.0]
    finally:
        host.close()


def test_local_agent_index():
    from sample_factory_amd.algo.sampling.parallel_env import local_agent_index as f
    assert f(None, 4, 2) == slice(None) and f(slice(None), 4, 2) == slice(None)
    assert f(5, 4, 2) == 1 and f(3, 4, 2) is None and f(6, 4, 2) is None
    assert f(slice(4, 6), 4, 2) == slice(None) and f(slice(0, 5), 4, 2) == slice(0, 1) and f(slice(6, 9), 4, 2) is None
    with pytest.raises(ValueError):
        f(slice(0, 8, 2), 4, 2)

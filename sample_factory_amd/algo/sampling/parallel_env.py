"""Multi-process HOST env stepping — what `RolloutWorker` processes are for in the reference
(sample_factory/algo/sampling/rollout_worker.py:79-308, algo/runners/runner_parallel.py:15-65, the per-env wrappers of
algo/utils/make_env.py:57-128,240-331): Python / C++ envs that live on the CPU are stepped by `num_workers` worker
processes, `num_envs_per_worker` env instances each, in parallel with each other AND with the policy's inference.

Design (one process per GPU owns everything that touches the device; workers never import torch.cuda):

  * every worker process creates its env instances through the SAME factory the reference calls
    (`make_env_func(full_env_name, cfg, env_config, render_mode)` with `env_config = AttrDict(worker_index, vector_index,
    env_id)`, batched_sampling.py:160-170) and steps them sequentially (rollout_worker.py: one thread per worker);
  * env outputs are written straight into SHARED MEMORY arrays laid out `[agents, ...]` in slab row order
    (observations per key, rewards f32, terminated / truncated bool) — the reference's `traj_tensors` in shared memory
    (shared_buffers.py:34-57) — and the main process registers those pages with the HIP runtime, so `sf_h2d_rows`
    DMAs a step's observations from the workers' own pages into slab column t (no staging copy, no pickling);
  * double-buffered sampling (`worker_num_splits = 2`, rollout_worker.py:96-117): the env instances of a worker are dealt
    to the splits; each split is one `ParallelVecEnvView` (an env-like object with `step_async` / `step_wait`), and the
    Runner pipelines them — while the workers step split A's envs the GPU runs split B's inference;
  * single-agent gym envs are wrapped as in make_env.py:97-128 (1-agent lists, auto-reset on done); batched /
    multi-agent envs (`num_agents` attribute) return per-agent vectors and reset themselves (make_env.py:147-237).

Commands travel over one `multiprocessing.Pipe` per worker; completion is one semaphore per (worker, split).
"""
from __future__ import annotations

import multiprocessing as mp
import traceback
from multiprocessing import shared_memory
from typing import Any, Callable, Dict, List, Optional, Tuple

import numpy as np

from sample_factory_amd.envs.env_utils import RewardShapingInterface
from sample_factory_amd.envs.spaces import action_head_sizes, heads_action_cols, heads_mixed, is_box, split_tuple_actions
from sample_factory_amd.utils.attr_dict import AttrDict

CMD_RESET, CMD_STEP, CMD_CLOSE, CMD_TRAINING_INFO, CMD_REWARD_SHAPING = 0, 1, 2, 3, 4


def _as_obs_dict(obs) -> Dict[str, Any]:
    return obs if isinstance(obs, dict) else {"obs": obs}


def env_is_batched(env) -> bool:
    """make_env.py:30-39: `num_agents` > 1 or an explicit `is_multiagent` -> the env speaks per-agent vectors"""
    n = getattr(env, "num_agents", 1)
    return bool(getattr(env, "is_multiagent", n > 1))


def probe_info(env):
    """(observation_space, action_space, agents per instance) of one env instance"""
    return env.observation_space, env.action_space, (int(getattr(env, "num_agents", 1)) if env_is_batched(env) else 1)


def obs_space_dict(space) -> Dict[str, Any]:
    return dict(space.spaces) if hasattr(space, "spaces") else {"obs": space}


class _Shm:
    """a numpy array on a named shared-memory block (created by the main process, attached by the workers)"""

    def __init__(self, shape, dtype, name: Optional[str] = None):
        self.shape, self.dtype = tuple(int(s) for s in shape), np.dtype(dtype)
        nbytes = max(1, int(np.prod(self.shape)) * self.dtype.itemsize)
        self.owner = name is None
        self.shm = shared_memory.SharedMemory(create=True, size=nbytes) if name is None else \
            shared_memory.SharedMemory(name=name)
        # (Python < 3.13 registers ATTACHED segments with the resource tracker too, bpo-38119.  The env workers are spawn /
        # fork children of the owner and therefore talk to the owner's tracker: the second registration is a no-op in its
        # set, nothing is unlinked when a worker exits, and the owner's unlink() removes the one entry.  An attaching
        # process must NOT unregister: that would delete the owner's registration as well.)
        self.array = np.ndarray(self.shape, dtype=self.dtype, buffer=self.shm.buf)
        if self.owner:
            self.array.fill(0)

    def spec(self):
        return (self.shm.name, self.shape, self.dtype.str)

    def close(self):
        self.array = None
        try:
            self.shm.close()
            if self.owner:
                self.shm.unlink()
        except Exception:  # noqa: BLE001 - best effort at shutdown
            pass


def local_agent_index(agent_idx, row0: int, nrows: int):
    """view-level agent index (int | slice | None) -> the index an instance owning rows [row0, row0 + nrows) understands, or
    None when none of its agents is meant.  A slice is cut to the instance's rows (step 1 only; the whole view = slice(None))."""
    if agent_idx is None:
        return slice(None)
    if isinstance(agent_idx, slice):
        if agent_idx.step not in (None, 1):
            raise ValueError(f"set_reward_shaping: strided agent slices are not supported ({agent_idx})")
        if agent_idx.start is None and agent_idx.stop is None:
            return slice(None)
        lo = max(agent_idx.start or 0, row0)
        hi = row0 + nrows if agent_idx.stop is None else min(agent_idx.stop, row0 + nrows)
        if lo >= hi:
            return None
        return slice(None) if (lo == row0 and hi == row0 + nrows) else slice(lo - row0, hi - row0)
    i = int(agent_idx)
    return i - row0 if row0 <= i < row0 + nrows else None


def _format_actions(act_rows: np.ndarray, heads: List[int], continuous: bool, batched: bool):
    """what `preprocess_actions` hands the env (batched_sampling.py:30-82): int32, the action axis squeezed for ONE Discrete
    head; an all-Discrete Tuple space gets the [agents, heads] int32 array (the `all_discrete` branch — pinned by
    tests/golden/rollout_tuple_heads.npz); a single-agent env gets its agent's row without the agent axis
    (make_env.py:97-99)"""
    if continuous:
        a = act_rows.astype(np.float32, copy=False)
        return a if batched else a[0]
    if heads_mixed(heads):  # Tuple with a Box member: one array per member (batched_sampling.py:51-59)
        return split_tuple_actions(act_rows, heads, batched)
    if len(heads) > 1:
        a = act_rows.reshape(act_rows.shape[0], len(heads))
        return a if batched else a[0]
    a = act_rows.reshape(-1)
    return a if batched else int(a[0])


class _InstanceStepper:
    """the env instances of ONE worker (or, inline, of the whole runner) and the arrays they write into: reset / step of one
    split, exactly what a rollout worker does per message (rollout_worker.py:201-259, make_env.py:97-128,147-237)"""

    def __init__(self, make_env_func, env_name, cfg, widx, instances, arrays, heads, continuous, render_mode=None):
        from sample_factory_amd.envs.env_utils import RewardShapingInterface, find_training_info_interface, find_wrapper_interface
        self.arrays, self.heads, self.continuous, self.envs = arrays, heads, continuous, []
        self.training_info_ifaces, self.reward_shaping_ifaces = [], []
        self._shaping_rows = []  # (split, first row, rows, interface) of every instance that can be re-shaped
        for split, vidx, env_id, row0, nrows in instances:
            env = make_env_func(env_name, cfg, AttrDict(worker_index=widx, vector_index=vidx, env_id=env_id), render_mode)
            self.envs.append((split, env, env_is_batched(env), row0, nrows, env_id))
            iface = find_training_info_interface(env)
            if iface is not None:
                self.training_info_ifaces.append(iface)
            iface = find_wrapper_interface(env, RewardShapingInterface)
            if iface is not None:
                self.reward_shaping_ifaces.append(iface)
                self._shaping_rows.append((split, row0, nrows, iface))

    def default_reward_shaping(self):
        """the scheme of the first instance that has one (env_utils.py:96-103 asks ONE env; instances of an env agree)"""
        return self.reward_shaping_ifaces[0].get_default_reward_shaping() if self.reward_shaping_ifaces else None

    def set_reward_shaping(self, reward_shaping, agent_idx, split=None) -> None:
        """`agent_idx` indexes the batched agent axis of ONE view (env_utils.py:106-111: an int, or a slice in batched mode):
        it is translated to the instance that owns those rows and handed over as that instance's LOCAL index — a per-agent
        (PBT-style) update reaches one agent, not every instance.  split=None / a slice over the whole view: everybody."""
        for sp_, row0, nrows, iface in self._shaping_rows:
            if split is not None and sp_ != split:
                continue
            local = local_agent_index(agent_idx, row0, nrows)
            if local is not None:
                iface.set_reward_shaping(reward_shaping, local)

    def set_training_info(self, training_info) -> None:
        """curricula (batched_sampling.py:352-355): every instance that implements TrainingInfoInterface gets the dict"""
        for iface in self.training_info_ifaces:
            iface.set_training_info(training_info)

    @staticmethod
    def _put_obs(a, obs, row0, nrows, batched):
        for k, v in _as_obs_dict(obs).items():
            dst = a["obs." + k]
            if batched:
                if isinstance(v, (list, tuple)):
                    v = np.stack([np.asarray(x) for x in v])
                v = np.asarray(v.cpu() if hasattr(v, "cpu") else v)
                np.copyto(dst[row0:row0 + nrows], v.reshape(dst[row0:row0 + nrows].shape), casting="unsafe")
            else:
                np.copyto(dst[row0], np.asarray(v).reshape(dst[row0].shape), casting="unsafe")

    def run(self, cmd: int, split: int) -> None:
        a = self.arrays[split]
        for sp_, env, batched, row0, nrows, env_id in self.envs:
            if sp_ != split:
                continue
            if cmd == CMD_RESET:
                try:
                    obs, _info = env.reset(seed=env_id)  # gymnasium >= 0.26 seeds in reset (make_env.py:206-214)
                except TypeError:
                    obs, _info = env.reset()
                self._put_obs(a, obs, row0, nrows, batched)
                continue
            act = _format_actions(a["act"][row0:row0 + nrows], self.heads, self.continuous, batched)
            obs, rew, term, trunc, _info = env.step(act)
            if batched:
                a["rew"][row0:row0 + nrows] = np.asarray(rew.cpu() if hasattr(rew, "cpu") else rew, dtype=np.float32).reshape(-1)
                a["term"][row0:row0 + nrows] = np.asarray(term.cpu() if hasattr(term, "cpu") else term).reshape(-1)
                a["trunc"][row0:row0 + nrows] = np.asarray(trunc.cpu() if hasattr(trunc, "cpu") else trunc).reshape(-1)
            else:
                if term or trunc:  # auto-reset (make_env.py:100-102); the terminal observation is dropped as there
                    obs, _ = env.reset()
                a["rew"][row0], a["term"][row0], a["trunc"][row0] = rew, bool(term), bool(trunc)
            self._put_obs(a, obs, row0, nrows, batched)

    def close(self) -> None:
        for _sp, env, *_ in self.envs:
            try:
                env.close()
            except Exception:  # noqa: BLE001
                pass


def _worker_main(widx: int, conn, done_sems, make_env_func: Callable, env_name: str, cfg, instances, specs, heads,
                 continuous: bool):
    """instances: [(split, vector_index, env_id, row0, nrows)] of this worker; specs[split] = {name: shm spec}"""
    arrays, shms, stepper = {}, [], None
    try:
        for split, sp in specs.items():
            arrays[split] = {}
            for name, (shm_name, shape, dt) in sp.items():
                s = _Shm(shape, dt, name=shm_name)
                shms.append(s)
                arrays[split][name] = s.array
        stepper = _InstanceStepper(make_env_func, env_name, cfg, widx, instances, arrays, heads, continuous)
        conn.send(("ready", widx, len(stepper.training_info_ifaces), stepper.default_reward_shaping()))
        while True:
            cmd, split = conn.recv()
            if cmd == CMD_CLOSE:
                break
            if cmd == CMD_TRAINING_INFO:  # payload in place of the split index; no completion signal
                stepper.set_training_info(split)
                continue
            if cmd == CMD_REWARD_SHAPING:  # payload = (scheme, agent index | slice fields, is_slice, split); no completion signal
                scheme, idx, is_slice, sp_ = split
                stepper.set_reward_shaping(scheme, slice(*idx) if is_slice else idx, sp_)
                continue
            stepper.run(cmd, split)
            done_sems[split].release()
    except BaseException:  # noqa: BLE001 - reported to the main process, which raises
        try:
            conn.send(("error", traceback.format_exc()))
        except Exception:  # noqa: BLE001
            pass
        for s in done_sems:
            s.release()
    finally:
        if stepper is not None:
            stepper.close()
        for s in shms:
            s.close()


class ParallelVecEnvView(RewardShapingInterface):
    """One split's agents over all workers, presented as ONE batched host env (reset / step / step_async / step_wait).
    It answers for its instances where the reference walks a wrapper chain: `get_default_reward_shaping(view)` /
    `set_reward_shaping(view, ...)` (envs/env_utils.py:96-111) reach the env instances in their worker processes."""

    def __init__(self, parent: "ParallelHostEnvs", split: int, num_agents: int):
        self.parent, self.split, self.num_agents = parent, split, int(num_agents)
        self.observation_space, self.action_space = parent.observation_space, parent.action_space
        self.is_multiagent = True
        self._pending = False
        self.pages_registered = False  # set by ParallelHostEnvs.register_with_device()

    def _arrays(self):
        return self.parent.arrays[self.split]

    def _obs(self):
        a = self._arrays()
        return {k: a["obs." + k] for k in self.parent.obs_keys}

    def reset(self, **kwargs):
        self.parent._command(self.split, CMD_RESET)
        self.parent._wait(self.split)
        return self._obs(), {}

    def step_async(self, actions) -> None:
        assert not self._pending, "step_async() twice without step_wait()"
        act = self._arrays()["act"]
        np.copyto(act, np.asarray(actions.cpu() if hasattr(actions, "cpu") else actions).reshape(act.shape), casting="unsafe")
        self.parent._command(self.split, CMD_STEP)
        self._pending = True

    def step_wait(self):
        assert self._pending
        self.parent._wait(self.split)
        self._pending = False
        a = self._arrays()
        return self._obs(), a["rew"], a["term"], a["trunc"], {}

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def set_training_info(self, training_info) -> None:
        """forwarded to every env instance that implements TrainingInfoInterface (in its worker process)"""
        if self.split == 0:  # one message per worker and rollout covers all splits
            self.parent.set_training_info(training_info)

    def get_default_reward_shaping(self):
        """the instances' default reward shaping scheme (None when no instance implements RewardShapingInterface)"""
        return self.parent.default_reward_shaping

    def set_reward_shaping(self, reward_shaping, agent_idx) -> None:
        whole = agent_idx is None or (isinstance(agent_idx, slice) and agent_idx.start is None and agent_idx.stop is None)
        if whole:
            if self.split == 0:  # one message per worker covers the instances of every split
                self.parent.set_reward_shaping(reward_shaping, agent_idx)
        else:                    # rows of THIS view: only the instance that owns them is re-shaped
            self.parent.set_reward_shaping(reward_shaping, agent_idx, split=self.split)

    def close(self):
        self.parent.close()


class ParallelHostEnvs:
    def __init__(self, cfg, env_name: str, make_env_func: Callable, num_workers: int, envs_per_worker: int,
                 num_splits: int = 1, start_method: Optional[str] = None, inline: bool = False, render_mode=None,
                 probed=None):
        """inline=True: no processes — the same instances, wrappers and row layout stepped in THIS process (serial_mode:
        single-agent gym envs such as BASELINE configs[0]'s two CartPole copies behind one batched view).
        probed = (observation_space, action_space, agents_per_instance) when the caller already looked at an instance."""
        assert envs_per_worker % num_splits == 0, f"{envs_per_worker=} must be a multiple of {num_splits=}"
        self.cfg, self.num_workers, self.envs_per_worker, self.num_splits = cfg, num_workers, envs_per_worker, num_splits
        self._closed, self._conns, self._procs, self._shms = False, [], [], []
        self._registered: List[int] = []  # page-locked base addresses (register_with_device), unregistered in close()
        # ---- probe ONE instance here for spaces / agents per instance (the reference spawns a process for this,
        # env_info.py:81-127; the instance is closed again before the workers start)
        if probed is None:
            probe = make_env_func(env_name, cfg, AttrDict(worker_index=0, vector_index=0, env_id=0), None)
            probed = probe_info(probe)
            try:
                probe.close()
            except Exception:  # noqa: BLE001
                pass
        self.observation_space, self.action_space, self.agents_per_instance = probed
        self._training_info_instances = 0
        self.default_reward_shaping = None
        spaces_ = obs_space_dict(self.observation_space)
        if not hasattr(self.observation_space, "spaces"):  # make_env.py:46-66: a bare space becomes Dict(obs=space)
            from sample_factory_amd.envs import spaces as _sp
            self.observation_space = _sp.Dict(spaces_)
        self.obs_keys = list(spaces_.keys())
        self.heads = action_head_sizes(self.action_space)
        self.continuous = is_box(self.action_space)
        per_split = envs_per_worker // num_splits
        self.agents_per_view = num_workers * per_split * self.agents_per_instance
        # ---- shared-memory arrays per split, rows in (worker, instance-in-split, agent) order
        self.arrays: List[Dict[str, np.ndarray]] = []
        specs: List[Dict[str, Tuple]] = []
        n = self.agents_per_view
        for _ in range(num_splits):
            arr, spec = {}, {}

            def add(name, shape, dtype):
                if inline:
                    arr[name], spec[name] = np.zeros(tuple(int(v) for v in shape), dtype=dtype), None
                    return
                s = _Shm(shape, dtype)
                self._shms.append(s)
                arr[name], spec[name] = s.array, s.spec()

            for k, sp in spaces_.items():
                dt = np.float32 if np.dtype(sp.dtype) == np.float64 else sp.dtype
                add("obs." + k, (n,) + tuple(sp.shape), dt)
            add("rew", (n,), np.float32)
            add("term", (n,), np.bool_)
            add("trunc", (n,), np.bool_)
            if self.continuous:
                add("act", (n, int(self.action_space.shape[0])), np.float32)
            elif heads_mixed(self.heads):  # Tuple with a Box member: the slab's f32 action rows
                add("act", (n, heads_action_cols(self.heads)), np.float32)
            else:
                add("act", (n, len(self.heads)) if len(self.heads) > 1 else (n,), np.int32)
            self.arrays.append(arr)
            specs.append(spec)
        self.inline, self._steppers = inline, []
        if inline:
            for w in range(num_workers):
                inst = []
                for v in range(envs_per_worker):
                    split, j = v // per_split, v % per_split
                    inst.append((split, v, w * envs_per_worker + v, (w * per_split + j) * self.agents_per_instance,
                                 self.agents_per_instance))
                self._steppers.append(_InstanceStepper(make_env_func, env_name, cfg, w, inst,
                                                       {s_: self.arrays[s_] for s_ in range(num_splits)}, self.heads,
                                                       self.continuous, render_mode))
                self._training_info_instances += len(self._steppers[-1].training_info_ifaces)
                if self.default_reward_shaping is None:
                    self.default_reward_shaping = self._steppers[-1].default_reward_shaping()
            self.views = [ParallelVecEnvView(self, s_, n) for s_ in range(num_splits)]
            return
        # ---- workers
        method = start_method or getattr(cfg, "env_worker_start_method", None) or "spawn"
        ctx = mp.get_context(method)
        self._done = [[ctx.Semaphore(0) for _ in range(num_splits)] for _ in range(num_workers)]
        for w in range(num_workers):
            inst = []
            for v in range(envs_per_worker):  # vector_index -> split as rollout_worker.py:96-117 deals them: blocks per split
                split, j = v // per_split, v % per_split
                row0 = (w * per_split + j) * self.agents_per_instance
                inst.append((split, v, w * envs_per_worker + v, row0, self.agents_per_instance))
            parent_conn, child_conn = ctx.Pipe()
            p = ctx.Process(target=_worker_main, name=f"sf-env-worker-{w}", daemon=True,
                            args=(w, child_conn, self._done[w], make_env_func, env_name, cfg, inst,
                                  {s: specs[s] for s in range(num_splits)}, self.heads, self.continuous))
            p.start()
            self._conns.append(parent_conn)
            self._procs.append(p)
        for w, c in enumerate(self._conns):
            if not c.poll(float(getattr(cfg, "env_worker_start_timeout", 300.0))):
                self.close()
                raise RuntimeError(f"env worker {w} did not start")
            msg = c.recv()
            if msg[0] != "ready":
                self.close()
                raise RuntimeError(f"env worker {w} failed to create its envs:\n{msg[1]}")
            self._training_info_instances += int(msg[2])
            if self.default_reward_shaping is None:
                self.default_reward_shaping = msg[3]
        self.views = [ParallelVecEnvView(self, s, n) for s in range(num_splits)]

    # ---- pinned pages: let the DMA engine read the workers' pages directly
    def register_with_device(self) -> bool:
        """hipHostRegister the observation pages (main process only; a no-op without a GPU).  Returns True if the pages are
        now page-locked: the rollout runner then hands them to sf_h2d_rows without a staging copy."""
        if self.inline:
            return False
        try:
            import torch
            if not torch.cuda.is_available():
                return False
            rt = torch.cuda.cudart()
            for arr in self.arrays:
                for k, a in arr.items():
                    if k.startswith("obs."):
                        err = rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0)
                        if int(err) != 0:
                            return False
                        self._registered.append(int(a.ctypes.data))
            for v in self.views:
                v.pages_registered = True
            return True
        except Exception:  # noqa: BLE001 - registration is an optimisation only
            return False

    def set_training_info(self, training_info) -> None:
        """rollout_worker.py: the runner's training info reaches the envs where they live; a no-op (no messages) when no
        instance implements TrainingInfoInterface"""
        if not self._training_info_instances or self._closed:
            return
        if self.inline:
            for st in self._steppers:
                st.set_training_info(training_info)
            return
        for c in self._conns:
            c.send((CMD_TRAINING_INFO, dict(training_info)))

    def set_reward_shaping(self, reward_shaping, agent_idx, split=None) -> None:
        """env_utils.py:106-111: a new reward shaping scheme for the env instances (where they live); `agent_idx` is a row
        of the view of `split` (None: every instance of every split)"""
        if self._closed or reward_shaping is None:
            return
        if self.inline:
            for st in self._steppers:
                st.set_reward_shaping(reward_shaping, agent_idx, split)
            return
        idx = (agent_idx.start, agent_idx.stop, agent_idx.step) if isinstance(agent_idx, slice) else agent_idx
        for c in self._conns:
            c.send((CMD_REWARD_SHAPING, (dict(reward_shaping), idx, isinstance(agent_idx, slice), split)))

    def _command(self, split: int, cmd: int) -> None:
        if self.inline:
            for st in self._steppers:
                st.run(cmd, split)
            return
        for c in self._conns:
            c.send((cmd, split))

    def _wait(self, split: int) -> None:
        if self.inline:
            return
        timeout = float(getattr(self.cfg, "env_worker_step_timeout", 600.0))
        for w in range(self.num_workers):
            waited = 0.0
            while not self._done[w][split].acquire(timeout=1.0):  # short waits: notice a worker that died without a word
                waited += 1.0
                if not self._procs[w].is_alive():
                    self.close()
                    raise RuntimeError(f"env worker {w} exited (exit code {self._procs[w].exitcode}) while stepping")
                if waited >= timeout:
                    self.close()
                    raise RuntimeError(f"env worker {w} did not answer within {timeout} s")
            if self._conns[w].poll(0):
                msg = self._conns[w].recv()
                if msg[0] == "error":
                    self.close()
                    raise RuntimeError(f"env worker {w} died:\n{msg[1]}")

    def close(self) -> None:
        if getattr(self, "_closed", True):
            return
        self._closed = True
        for st in getattr(self, "_steppers", []):
            st.close()
        for c in self._conns:
            try:
                c.send((CMD_CLOSE, 0))
            except Exception:  # noqa: BLE001
                pass
        for p in self._procs:
            p.join(timeout=5.0)
            if p.is_alive():
                p.terminate()
        if self._registered:  # the pages must leave the HIP runtime's tables before they are unmapped
            try:
                import torch
                rt = torch.cuda.cudart()
                for ptr in self._registered:
                    rt.cudaHostUnregister(ptr)
            except Exception:  # noqa: BLE001 - best effort at shutdown
                pass
            self._registered = []
        for v in getattr(self, "views", []):
            v.pages_registered = False
        for s in self._shms:
            s.close()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

"""Batched rollout on the device — the data path of sample_factory/algo/sampling/batched_sampling.py:298-388
(BatchedVectorEnvRunner.advance_rollouts / generate_policy_request / _finalize_trajectories) and of
inference_worker.py:313-341 (InferenceWorker._handle_policy_steps), collapsed into one stream-ordered loop:

  per step t:   policy forward on slab obs[:, t] (in place)          K2+K3  sf_conv_fwd stack
                sample + write policy outputs into traj[:, t]         K4+K5  sf_sample_write_step
                env.step -> obs straight into slab obs[:, t+1]        K1     (zero-copy for envs with step_into)
                rewards/dones/time_outs/policy_id + episode stats     K6     sf_traj_write_env_step

No host synchronisation happens inside a rollout (the reference syncs the device and round-trips dones/rewards to the
CPU every step: batched_sampling.py:216,323,390-392; torch_utils.py:58-66).
"""
from __future__ import annotations

import ctypes as C
import time
from typing import Dict, Optional

import numpy as np
import torch

from sample_factory_amd import lib
from sample_factory_amd.algo.utils.tensor_dict import TensorDict
from sample_factory_amd.envs.spaces import action_head_sizes, heads_mixed, is_box, split_tuple_actions


class BatchedVectorEnvRunner:
    def __init__(self, cfg, env_info, env, actor_critic, traj: TensorDict, policy_id: int = 0,
                 policy_versions: Optional[torch.Tensor] = None, sample_seed: int = 0, row0: int = 0, tag: str = "inf"):
        self.cfg, self.env_info, self.env, self.ac = cfg, env_info, env, actor_critic
        self.tag = tag  # activation / workspace namespace inside the model: one per concurrently running env group
        self.traj = traj
        self.policy_id = policy_id
        self.policy_versions = policy_versions
        self.B = env.num_agents
        self.T = cfg.rollout
        dev = actor_critic.device
        self.device = dev
        assert traj["rewards"].shape == (self.B, self.T), "one slab row per agent (sync mode: one rollout per dataset)"
        self.heads = action_head_sizes(env_info.action_space)  # [n] | [n1, -D2, ...] (Tuple; -D = a Box(D) member) | [] (Box)
        self.mixed = heads_mixed(self.heads)  # Tuple with a Box member: the env reads the slab's f32 action row, split per member
        self.env_actions = torch.zeros((self.B, len(self.heads)) if len(self.heads) > 1 else self.B, dtype=torch.int32,
                                       device=dev)
        self.ep_return = torch.zeros(self.B, dtype=torch.float32, device=dev)
        self.ep_len = torch.zeros(self.B, dtype=torch.int32, device=dev)
        self.ep_stats = torch.zeros(3, dtype=torch.float64, device=dev)  # sum_return, sum_len, episodes
        self.sample_seed = int(sample_seed)
        self.row0 = int(row0)  # global index of this replica's first env (Philox key of the action sampler)
        self.global_step = 0
        # observation keys besides the action mask; more than one: the model (torch fallback) takes {key: view} dicts
        self.obs_keys = sorted(k for k in traj["obs"].keys() if k != "action_mask")
        self.multi_key = len(self.obs_keys) > 1
        self.zero_copy = hasattr(env, "step_into") and not self.multi_key
        self.obs = traj["obs"]["obs" if "obs" in traj["obs"] else self.obs_keys[0]]
        self.rnn = actor_critic.rnn_kind is not None
        traj["rnn_states"].zero_()
        self._dummy_state_rows = {traj["rnn_states"].data_ptr()}
        self._started = False
        self.A = actor_critic.num_action_params
        self.ld = actor_critic.heads_ld
        self.continuous = is_box(self.env_info.action_space)  # Box(D): params = [means | log_std]
        self.host_env = None
        self.async_env = hasattr(env, "step_async") and hasattr(env, "step_wait")  # stepped by worker processes
        self._pin, self._pin_flip, self._direct_ok = {}, {}, {}
        self._act_event = torch.cuda.Event()
        self.h2d_bytes = 0  # host-env ingest volume (bench --workload c3)
        # optional host-timeline probe of the ingest path (bench --workload c3): seconds spent waiting for the actions
        # (= exposed inference latency), inside env.step, staging into pinned memory; HIP-event pairs around obs DMAs
        self.ingest_prof: Optional[dict] = None
        self._rew = torch.zeros(self.B, dtype=torch.float32, device=dev)
        self._term = torch.zeros(self.B, dtype=torch.bool, device=dev)
        self._trunc = torch.zeros(self.B, dtype=torch.bool, device=dev)
        # obs["action_mask"] (inference_worker.py:324-331): kept in the slab like every other obs key, used by the sampler
        # launch programs (lib.LaunchProgram): the library calls of step t — policy forward + sampler, and the trajectory
        # write + state store behind the env step — recorded once per (t, slab, stream, model layout) and replayed with
        # one foreign call per launch.  The two values that change between replays live in ctypes cells.
        self._c_step, self._c_ver = C.c_uint32(0), C.c_float(0.0)
        self._progs: Dict = {}
        self._progs_layout = None
        self.program_replays = 0
        self._launch_key = getattr(actor_critic, "launch_key", None) if lib.LAUNCH_PROGRAMS else None
        self.masked = "action_mask" in traj["obs"]
        if self.masked and (self.continuous or len(self.heads) != 1):
            raise NotImplementedError("action masks are supported for a single Discrete action space")
        if self.masked and self.zero_copy:
            raise NotImplementedError("action masks with a zero-copy (step_into) env")

    def reset(self) -> None:
        """First observation into slab obs[:, 0] (batched_sampling.py:172-206)."""
        if self.zero_copy:
            self.env.reset_into(self.obs[:, 0])
        else:
            o, _ = self.env.reset()
            first = (o["obs"] if "obs" in o else o[self.obs_keys[0]]) if isinstance(o, dict) else o
            self.host_env = not (isinstance(first, torch.Tensor) and first.is_cuda)
            self._store_obs(o, 0)
        self._started = True

    # ---- host (CPU) envs: pinned staging, asynchronous H2D straight into the slab slot.  The only
    # host sync of a step is the one the data dependency demands: the env needs the sampled actions.
    def _pinned(self, key, shape, dtype, nbuf=1):
        bufs = self._pin.get(key)
        if bufs is None or bufs[0].shape != torch.Size(shape) or bufs[0].dtype != dtype:
            bufs = [torch.empty(shape, dtype=dtype, pin_memory=True) for _ in range(nbuf)]
            self._pin[key] = bufs
        self._pin_flip[key] = (self._pin_flip.get(key, -1) + 1) % len(bufs)
        return bufs[self._pin_flip[key]]

    def _to_device(self, key, src, dst: torch.Tensor) -> None:
        """dst (device view, e.g. slab[:, t]) <- src (device tensor | numpy / list from a host env)"""
        if isinstance(src, torch.Tensor) and src.is_cuda:
            dst.copy_(src)
            return
        prof = self.ingest_prof
        t0 = time.perf_counter() if prof is not None else 0.0
        a = np.asarray(src.cpu() if isinstance(src, torch.Tensor) else src)
        stage = None
        if getattr(self.env, "pages_registered", False) and isinstance(src, np.ndarray) and src.flags.c_contiguous and \
                src.dtype == torch.empty(0, dtype=dst.dtype).numpy().dtype and src.size == dst.numel():
            # the env workers' shared pages are registered with the HIP runtime (parallel_env.register_with_device): the
            # DMA engine reads them in place.  Safe without a second buffer: the workers only overwrite them after the
            # NEXT actions were read back, which is stream-ordered behind this copy.
            cand = torch.from_numpy(src).view(dst.shape)
            ok = self._direct_ok.get(key)
            if ok is None:  # (asked once per key: is_pinned() is a runtime query)
                ok = self._direct_ok[key] = bool(cand.is_pinned())
            stage = cand if ok else None
        if stage is None:
            stage = self._pinned(key, dst.shape, dst.dtype, nbuf=2)  # two buffers: the previous H2D may still be in flight
            np.copyto(stage.numpy(), a.reshape(stage.shape), casting="unsafe")
        timed = prof is not None and key.startswith("obs")
        if prof is not None:
            prof["stage_s"] = prof.get("stage_s", 0.0) + time.perf_counter() - t0
        if timed:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        if dst.dim() >= 1 and (dst.is_contiguous() or (dst.dim() > 1 and dst[0].is_contiguous())):
            lib.h2d_rows(dst, stage)  # ONE pitched DMA straight into slab[:, t] (no contiguous device temp + copy kernel)
        else:
            dst.copy_(stage, non_blocking=True)
        if timed:
            ev1.record()
            prof.setdefault("dma_events", []).append((ev0, ev1, stage.numel() * stage.element_size()))
        self.h2d_bytes += stage.numel() * stage.element_size()

    def _actions_to_host(self, env_actions: torch.Tensor):
        """device actions -> pinned host array the env can read in place (ONE event wait)"""
        stage = self._pinned("act", env_actions.shape, env_actions.dtype)
        stage.copy_(env_actions, non_blocking=True)
        self._act_event.record()
        t0 = time.perf_counter()
        self._act_event.synchronize()
        if self.ingest_prof is not None:
            self.ingest_prof["act_wait_s"] = self.ingest_prof.get("act_wait_s", 0.0) + time.perf_counter() - t0
        return stage.numpy()

    def _store_obs(self, o, t: int) -> None:
        if self.multi_key:
            for k in self.obs_keys:
                self._to_device("obs." + k, o[k], self.traj["obs"][k][:, t])
        else:
            self._to_device("obs", o["obs"] if isinstance(o, dict) else o, self.obs[:, t])
        if self.masked:
            self._to_device("mask", o["action_mask"], self.traj["obs"]["action_mask"][:, t])

    def policy_version(self) -> float:
        return float(self.policy_versions[self.policy_id].item()) if self.policy_versions is not None else 0.0

    def rollout(self, policy_version: Optional[float] = None, deterministic: bool = False) -> None:
        """Collect cfg.rollout steps for all agents into the slab; leaves obs[:, T] = last observation
        (_finalize_trajectories, batched_sampling.py:289-296)."""
        self.begin_rollout(policy_version, deterministic)
        for t in range(self.T):
            self.rollout_step(t)

    def begin_rollout(self, policy_version: Optional[float] = None, deterministic: bool = False) -> None:
        if not self._started:
            self.reset()
        self._ver = self.policy_version() if policy_version is None else float(policy_version)
        self._deterministic = bool(deterministic)

    def rollout_step(self, t: int) -> None:
        """step t of the current rollout (the Runner interleaves the steps of several env groups, each on its own HIP
        stream: the reference's double-buffered sampling, worker_num_splits / batched_sampling.py:298-388)"""
        self.rollout_step_begin(t)
        self.rollout_step_finish(t)

    def _program(self, half: str, t: int, extra=()):
        """(program to replay | None, key to record under | None) for one half of step t.  A key is run plainly the first
        time it is seen (buffers are allocated lazily), recorded the second time, replayed from the third on; anything
        that moves an address the calls hold changes the key (the model's launch_key, the slab rows, the stream)."""
        if self._launch_key is None:
            return None, None
        layout, weights = self._launch_key(self.tag)
        if layout != self._progs_layout:  # a buffer was (re)allocated: drop every program (and the memory they hold)
            self._progs.clear()
            self._progs_layout = layout
        key = (half, t, self.traj["rewards"].data_ptr(), self._deterministic, torch.cuda.current_stream().cuda_stream,
               weights, extra)
        slot = self._progs.get(key)
        if slot is None:                       # first sight: run plainly
            if len(self._progs) > 64 * self.T:
                self._progs.clear()
            self._progs[key] = False
            return None, None
        if slot is False:                      # second sight: record
            return None, key
        return (slot, None) if slot.unsafe is None else (None, None)

    def rollout_step_begin(self, t: int) -> None:
        """policy forward + sampling of step t and, for in-process envs, the env step itself"""
        self._c_step.value, self._c_ver.value = self.global_step & 0xFFFFFFFF, self._ver
        prog, rec_key = self._program("policy", t)
        if prog is not None:
            prog.replay()
            self.program_replays += 1
        elif rec_key is not None:
            with lib.record_launches() as prog:
                self._policy_and_sample(t)
            self._progs[rec_key] = prog
        else:
            self._policy_and_sample(t)
        tr = self.traj
        # Box: f32 [B, D] view of the slab; Tuple with a Box member: f32 [B, columns] view, split per member for the env
        env_actions = tr["actions"][:, t] if (self.continuous or self.mixed) else self.env_actions
        if self.async_env:  # worker processes step the envs from here on (parallel_env.py); rollout_step_finish collects
            self.env.step_async(self._actions_to_host(env_actions))
            return
        self._env_step_and_record(t, env_actions)

    def _policy_and_sample(self, t: int) -> None:
        tr, T, B, A = self.traj, self.T, self.B, self.A
        # (ctypes cells, not numbers: a recorded program reads their CURRENT value at every replay)
        ver, deterministic, cfg, step = self._c_ver, self._deterministic, self.cfg, self._c_step
        rnn = dict(states=tr["rnn_states"][:, t]) if self.rnn else None  # the state INPUT of step t (parity trap 13)
        x = {k: tr["obs"][k][:, t] for k in self.obs_keys} if self.multi_key else self.obs[:, t]
        heads = self.ac.forward_heads(x, B, sample_stride=self.obs.stride(0), tag=self.tag, rnn=rnn)[-1]
        if self.masked:
            mk = tr["obs"]["action_mask"][:, t]
            lib.sample_write_step_masked(heads[:, 1:], self.ld, heads[:, 0], self.ld, mk, mk.stride(0), B, A, T, t,
                                         self.sample_seed, step, self.row0, ver, deterministic,
                                         tr["actions"], tr["action_logits"], tr["log_prob_actions"], tr["values"],
                                         tr["policy_version"], self.env_actions)
        elif len(self.heads) > 1:  # Tuple space: one categorical per Discrete member, a diagonal normal per Box member
            lib.sample_write_step_tuple(heads[:, 1:], self.ld, heads[:, 0], self.ld, B, self.heads, T, t,
                                        self.sample_seed, step, self.row0, ver, deterministic,
                                        tr["actions"], tr["action_logits"], tr["log_prob_actions"], tr["values"],
                                        tr["policy_version"], None if self.mixed else self.env_actions)
        else:
            lib.sample_write_step(heads[:, 1:], self.ld, heads[:, 0], self.ld, B, A, T, t, self.sample_seed,
                                  step, self.row0, ver, deterministic, tr["actions"],
                                  tr["action_logits"], tr["log_prob_actions"], tr["values"], tr["policy_version"],
                                  None if self.continuous else self.env_actions, action_kind=int(self.continuous))

    def rollout_step_finish(self, t: int) -> None:
        """second half of a step for envs that are stepped asynchronously by worker processes (`step_async` / `step_wait`):
        wait for the env outputs, ingest them, record the step.  Between `rollout_step_begin(t)` and this call the Runner
        runs the inference of the OTHER sampling split — the reference's double-buffered sampling (rollout_worker.py:96-117)."""
        if self.async_env:
            self._env_step_and_record(t, None)

    def _env_step_and_record(self, t: int, env_actions) -> None:
        tr, T, cfg = self.traj, self.T, self.cfg
        if self.zero_copy:
            if self.mixed:  # batched_sampling.py:51-59: a list with one (device) array per Tuple member
                env_actions = split_tuple_actions(env_actions, self.heads)
            rew, term, trunc = self.env.step_into(env_actions, self.obs[:, t + 1])
        else:
            if self.host_env is None:  # decided by what reset() returned
                self.host_env = False
            t0 = time.perf_counter()
            if self.async_env:
                o, rew, term, trunc, _ = self.env.step_wait()
            else:
                acts_in = self._actions_to_host(env_actions) if self.host_env else env_actions
                if self.mixed:  # batched_sampling.py:51-59: a list with one array per Tuple member
                    acts_in = split_tuple_actions(acts_in, self.heads)
                o, rew, term, trunc, _ = self.env.step(acts_in)
            if self.ingest_prof is not None:
                self.ingest_prof["env_step_s"] = self.ingest_prof.get("env_step_s", 0.0) + time.perf_counter() - t0
            self._store_obs(o, t + 1)
            if self.host_env:
                self._to_device("rew", rew, self._rew)
                self._to_device("term", term, self._term)
                self._to_device("trunc", trunc, self._trunc)
                rew, term, trunc = self._rew, self._term, self._trunc
            else:
                rew = torch.as_tensor(rew, dtype=torch.float32, device=self.device).contiguous()
                term = torch.as_tensor(term, dtype=torch.bool, device=self.device).contiguous()
                trunc = torch.as_tensor(trunc, dtype=torch.bool, device=self.device).contiguous()
        parts = self.ac.new_rnn_parts_of(self.tag) if self.rnn else None
        if (self.rnn and parts is None) or not (self.zero_copy or self.host_env):
            prog = rec_key = None  # torch model path (the state store is a torch op) / env outputs in fresh tensors every step
        else:
            # (the shaping constants are launch arguments: a cfg update between rollouts must not replay the old ones)
            prog, rec_key = self._program("record", t, (rew.data_ptr(), term.data_ptr(), trunc.data_ptr(),
                                                        float(cfg.reward_scale), float(cfg.reward_clip)))
        if prog is not None:
            prog.replay()
        elif rec_key is not None:
            with lib.record_launches() as prog:
                self._record_step(t, rew, term, trunc)
            self._progs[rec_key] = prog
        else:
            self._record_step(t, rew, term, trunc)
        self.global_step += 1

    def _record_step(self, t: int, rew, term, trunc) -> None:
        """env outputs of step t -> slab (rewards shaped, dones, episode statistics), next-step recurrent state"""
        tr, T, cfg = self.traj, self.T, self.cfg
        lib.traj_write_env_step(rew, term, trunc, T, t, cfg.reward_scale, cfg.reward_clip, self.policy_id,
                                tr["rewards"], tr["dones"], tr["time_outs"], tr["policy_id"], self.ep_return,
                                self.ep_len, self.ep_stats)
        if self.rnn:  # batched_sampling.py:332-335: next-step state = new_rnn_states * (1 - done)
            # the state THIS runner's forward produced (keyed by its tag: the learner thread may have run its bootstrap
            # forward through the same model object since)
            parts = self.ac.new_rnn_parts_of(self.tag)
            if parts is not None:  # native model: mask + store [h | c] in one launch per recurrent layer
                dst, SL = tr["rnn_states"][:, t + 1], tr["rnn_states"].shape[-1] // len(parts)
                for l, (h_, c_) in enumerate(parts):  # stacked layers: layer l owns columns [l * SL, (l + 1) * SL) (core.py:54-58)
                    lib.rnn_store_state(h_, c_, tr["dones"][:, t], dst[:, l * SL:(l + 1) * SL])
            else:                  # torch model path
                keep = (~tr["dones"][:, t]).to(torch.float32).unsqueeze(1)
                torch.mul(self.ac.new_rnn_states_of(self.tag), keep, out=tr["rnn_states"][:, t + 1])

    def set_slab(self, traj: TensorDict, carry_from: Optional[TensorDict] = None) -> None:
        """async mode: switch to another slab; its step 0 continues from the last step of `carry_from`"""
        self.traj = traj
        self.obs = traj["obs"]["obs" if "obs" in traj["obs"] else self.obs_keys[0]]
        if not self.rnn and traj["rnn_states"].data_ptr() not in self._dummy_state_rows:
            # feed-forward policies carry a width-1 dummy state that is always zero and that the reference still writes
            # at every step (parity trap 13, model_utils.py:11-24): zero the rows ONCE, nothing overwrites them later
            traj["rnn_states"].zero_()
            self._dummy_state_rows.add(traj["rnn_states"].data_ptr())
        if carry_from is not None:  # column copies through the library's own row-copy kernel (sf_copy_rows)
            for k in self.obs_keys:
                lib.copy_rows(traj["obs"][k][:, 0], carry_from["obs"][k][:, self.T])
            if self.masked:
                lib.copy_rows(traj["obs"]["action_mask"][:, 0], carry_from["obs"]["action_mask"][:, self.T])
            if self.rnn:
                lib.copy_rows(traj["rnn_states"][:, 0], carry_from["rnn_states"][:, self.T])
        elif self.rnn:
            traj["rnn_states"][:, 0].zero_()

    def carry_over(self) -> None:
        """The next rollout starts from the last observation: slab obs[:, 0] <- obs[:, T] (one frame per agent)."""
        for k in self.obs_keys:
            lib.copy_rows(self.traj["obs"][k][:, 0], self.traj["obs"][k][:, self.T])
        if self.masked:
            mk = self.traj["obs"]["action_mask"]
            lib.copy_rows(mk[:, 0], mk[:, self.T])
        if self.rnn:
            lib.copy_rows(self.traj["rnn_states"][:, 0], self.traj["rnn_states"][:, self.T])

    def episode_stats(self) -> Dict[str, float]:
        s = self.ep_stats.cpu()
        n = float(s[2])
        return dict(episodes=n, mean_return=float(s[0]) / n if n else 0.0, mean_len=float(s[1]) / n if n else 0.0)

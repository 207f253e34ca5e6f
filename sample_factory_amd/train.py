"""`run_rl(cfg)` / `make_runner(cfg)` — the entry points of sample_factory/train.py:12-41 for the native engine.

One process per GPU owns env, inference, GAE and learning (the design the reference itself recommends for GPU envs:
docs/07-advanced-topics/profiling.md:178-180, serial_mode + sync in sf_examples/brax/train_brax.py:200-201).  With
torchrun (WORLD_SIZE>1) each rank runs a replica on its own env shard and gradients are all-reduced over RCCL.
"""
from __future__ import annotations

import os
import time
from collections import deque
from typing import Any, Callable, Dict, List, Tuple

import numpy as np

import torch

from sample_factory_amd.algo.learning.learner import Learner, ParameterServer
from sample_factory_amd.algo.sampling.batched_sampling import BatchedVectorEnvRunner
from sample_factory_amd.algo.utils.env_info import extract_env_info
from sample_factory_amd.algo.utils.shared_buffers import alloc_trajectory_tensors
from sample_factory_amd.cfg.arguments import preprocess_cfg
from sample_factory_amd.envs.env_utils import create_env, find_training_info_interface, set_training_info
from sample_factory_amd.model.actor_critic import get_rnn_size
from sample_factory_amd.utils.attr_dict import AttrDict


class ExperimentStatus:
    SUCCESS, FAILURE, INTERRUPTED = 0, 1, 2


EPISODIC, POLICY_ID_KEY, TRAIN_STATS, LEARNER_ENV_STEPS = "episodic", "policy_id", "train", "learner_env_steps"  # misc.py:7-16


class AlgoObserver:
    """Extension hooks of the reference runner (algo/runners/runner.py:52-73), same names and call points."""

    def on_init(self, runner: "Runner") -> None:
        """after Runner.init() built env / learner / slab, before anything runs"""

    def on_connect_components(self, runner: "Runner") -> None:
        """the reference connects extra signal-slot pairs here; called once right after on_init"""

    def on_start(self, runner: "Runner") -> None:
        """right before the first iteration of Runner.run()"""

    def on_training_step(self, runner: "Runner", training_iteration_since_resume: int) -> None:
        """after each Learner.train()"""

    def extra_summaries(self, runner: "Runner", policy_id: int, env_steps: int, writer) -> None:
        """called at every report with writer=None (no tensorboard writer in this engine)"""

    def on_stop(self, runner: "Runner") -> None:
        """after the last iteration"""


class Runner:
    """The slice of sample_factory/algo/runners/runner.py the hot path needs: init(), run(), FPS accounting
    (runner.py:748-765: env_steps / wall-clock), the observer / message-handler plugin hooks (runner.py:232-249,
    263-278, 481-495) fed from device-side episode statistics."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.status = ExperimentStatus.SUCCESS
        self.observers: List[AlgoObserver] = []
        self.env_steps = 0
        self.fps = 0.0
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.report_interval_sec = 5.0  # runner.py:117
        self.training_iteration_since_resume = 0
        # key -> per-policy deque(maxlen=cfg.stats_avg), filled by the default episodic-stats handler
        self.policy_avg_stats: Dict[str, List[deque]] = dict()
        self.msg_handlers: Dict[str, List[Callable]] = dict()
        self.policy_msg_handlers: Dict[str, List[Callable]] = {EPISODIC: [Runner._episodic_stats_handler]}

    def register_observer(self, observer: AlgoObserver) -> None:
        self.observers.append(observer)

    def _observers_call(self, name: str, *args, **kwargs) -> None:
        for observer in self.observers:
            getattr(observer, name)(*args, **kwargs)

    # ---- message handlers (runner.py:232-249, 481-488): handler(runner, msg) / handler(runner, msg, policy_id)
    def register_msg_handler(self, key, func: Callable[[Any, dict], None]) -> None:
        self.msg_handlers.setdefault(key, []).append(func)

    def register_policy_msg_handler(self, key, func: Callable[[Any, dict, int], None]) -> None:
        self.policy_msg_handlers.setdefault(key, []).append(func)

    def register_episodic_stats_handler(self, func: Callable[[Any, dict, int], None]) -> None:
        self.policy_msg_handlers.setdefault(EPISODIC, []).append(func)

    def _process_msg(self, msgs) -> None:
        if isinstance(msgs, dict):
            msgs = (msgs,)
        for msg in msgs:
            policy_id = msg.get(POLICY_ID_KEY, None)
            for key in list(msg):
                for handler in self.msg_handlers.get(key, ()):
                    handler(self, msg)
                if policy_id is not None:
                    for handler in self.policy_msg_handlers.get(key, ()):
                        handler(self, msg, policy_id)

    @staticmethod
    def _episodic_stats_handler(runner: "Runner", msg: Dict, policy_id: int) -> None:
        """runner.py:263-278"""
        for key, value in msg[EPISODIC].items():
            if key not in runner.policy_avg_stats:
                runner.policy_avg_stats[key] = [deque(maxlen=runner.cfg.stats_avg) for _ in range(runner.cfg.num_policies)]
            if isinstance(value, np.ndarray) and value.ndim > 0:
                if len(value) > runner.policy_avg_stats[key][policy_id].maxlen:
                    runner.policy_avg_stats[key][policy_id] = deque(maxlen=len(value))
                runner.policy_avg_stats[key][policy_id].extend(value)
            else:
                runner.policy_avg_stats[key][policy_id].append(value)

    def _emit_episodic_stats(self) -> None:
        """Episode statistics live on the device (sum of returns, sum of lengths, count per env instance — the
        reference round-trips dones/rewards to the host every step, batched_sampling.py:215-287).  At every report they
        are read back ONCE, turned into the reference's message {EPISODIC: {reward, len}, policy_id} (one entry = the
        mean over the episodes that finished since the previous report, plus their count under "episodes") and reset."""
        tot = sum(sm.ep_stats.cpu() for sm in self.samplers)
        for sm in self.samplers:
            sm.ep_stats.zero_()
        k = float(tot[2])
        if k > 0:
            self._process_msg({EPISODIC: dict(reward=float(tot[0]) / k, len=float(tot[1]) / k, episodes=k),
                               POLICY_ID_KEY: 0})

    def init(self) -> int:
        cfg = self.cfg
        if self.world > 1:
            if not torch.distributed.is_initialized():
                torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
                torch.distributed.init_process_group(os.environ.get("SF_DP_BACKEND", "nccl"))  # nccl = RCCL on ROCm
            cfg.data_parallel = True
            if getattr(cfg, "synthetic_env0", None) in (None, 0):  # env shard of this replica (weak scaling)
                cfg.synthetic_env0 = (self.rank * getattr(cfg, "synthetic_num_agents", 4096) *
                                      max(1, int(cfg.num_workers) * int(cfg.num_envs_per_worker)))
        # Env instances: num_workers * num_envs_per_worker vector envs, as in the reference (rollout_worker.py:96-117,
        # each with its own env_config); here they all live in this process and fill consecutive row blocks of ONE
        # slab.  With more than one instance their rollouts run on cfg.worker_num_splits HIP streams (the reference's
        # double-buffered sampling: while one group's envs step, the other group's policy forward runs), which fills
        # the tails of the small-batch kernels: measured in tools/split_probe.py and DESIGN.md §5.
        E = max(1, int(cfg.num_workers) * int(cfg.num_envs_per_worker))
        self.envs = [create_env(cfg.env, cfg, AttrDict(worker_index=e // int(cfg.num_envs_per_worker),
                                                       vector_index=e % int(cfg.num_envs_per_worker), env_id=e))
                     for e in range(E)]
        self.env = self.envs[0]
        self.env_info = extract_env_info(self.env, cfg)
        n = self.env_info.num_agents
        if any(env.num_agents != n for env in self.envs):
            raise ValueError("all env instances must have the same number of agents")
        if not preprocess_cfg(cfg, self.env_info):
            raise ValueError("Invalid config! See above for details.")
        self.policy_versions = torch.zeros(cfg.num_policies, dtype=torch.int32)
        self.learner = Learner(cfg, self.env_info, self.policy_versions, 0, ParameterServer(0, self.policy_versions))
        self.learner.init()
        dev = self.learner.device
        self.num_rows = E * n
        self.traj = alloc_trajectory_tensors(self.env_info, self.num_rows, cfg.rollout, get_rnn_size(cfg), dev)
        # one sampling stream for the whole job: key = (seed, global env row) -> a G-replica (or G-instance) rollout is
        # bit-identical to the single-replica rollout of the concatenated env set
        self.samplers = [BatchedVectorEnvRunner(cfg, self.env_info, env, self.learner.actor_critic,
                                                self.traj[e * n:(e + 1) * n], 0, self.policy_versions,
                                                sample_seed=(cfg.seed or 0), row0=self.rank * self.num_rows + e * n,
                                                tag="inf" if e == 0 else f"inf{e}")
                         for e, env in enumerate(self.envs)]
        self.sampler = self.samplers[0]
        S = max(1, min(int(cfg.worker_num_splits), E))
        self.split_streams = [torch.cuda.Stream() for _ in range(S)] if E > 1 else None
        self._ev_fork = torch.cuda.Event()
        self._ev_join = [torch.cuda.Event() for _ in range(S)]
        self._training_info_ifaces = [find_training_info_interface(env) for env in self.envs]
        if cfg.async_rl:  # rollout k+1 overlaps Learner.train(k): two slabs, two streams, published weight snapshots
            self.traj2 = alloc_trajectory_tensors(self.env_info, self.num_rows, cfg.rollout, get_rnn_size(cfg), dev)
            self.slabs = [self.traj, self.traj2]
            self.rollout_stream = torch.cuda.Stream()
            self.ev_rollout = [torch.cuda.Event(), torch.cuda.Event()]
            self.ev_train = [torch.cuda.Event(), torch.cuda.Event()]
            self.ev_publish = torch.cuda.Event()
            self.learner.actor_critic.enable_weight_snapshots()
            self.k = 0
            self.published_version = float(self.learner.train_step)
        self._observers_call("on_init", self)
        self._observers_call("on_connect_components", self)
        return ExperimentStatus.SUCCESS

    def _rollout_all(self, policy_version: float, slab=None, carry_from=None) -> None:
        """one rollout of every env instance into its row block of `slab` (default: the current slab), enqueued behind
        everything already on the current stream; returns with the current stream waiting for all of them"""
        n = self.env_info.num_agents
        for iface in self._training_info_ifaces:  # curricula: batched_sampling.py:352-355
            set_training_info(iface, dict(approx_total_training_steps=int(self.learner.env_steps)))
        if slab is not None:
            for e, sm in enumerate(self.samplers):
                sm.set_slab(slab[e * n:(e + 1) * n], carry_from=carry_from[e * n:(e + 1) * n] if carry_from is not None else None)
        if self.split_streams is None:
            self.sampler.rollout(policy_version=policy_version)
            return
        base, S = torch.cuda.current_stream(), len(self.split_streams)
        for sm in self.samplers:
            sm.begin_rollout(policy_version)  # (first call: env reset into slab obs[:, 0], on the base stream)
        self._ev_fork.record(base)
        for st in self.split_streams:
            st.wait_event(self._ev_fork)
        for t in range(self.cfg.rollout):  # steps of the groups interleaved on the host, concurrent on the device
            for e, sm in enumerate(self.samplers):
                with torch.cuda.stream(self.split_streams[e % S]):
                    sm.rollout_step(t)
        for i, st in enumerate(self.split_streams):
            self._ev_join[i].record(st)
            base.wait_event(self._ev_join[i])

    def episode_stats(self):
        """episode statistics over all env instances"""
        tot = sum(sm.ep_stats.cpu() for sm in self.samplers)
        k = float(tot[2])
        return dict(episodes=k, mean_return=float(tot[0]) / k if k else 0.0, mean_len=float(tot[1]) / k if k else 0.0)

    def iteration(self):
        """one dataset: rollout of all envs, then Learner.train on the slab in place"""
        if self.cfg.async_rl:
            stats = self.iteration_async()
        else:
            self._rollout_all(float(self.learner.train_step))
            stats = self.learner.train(self.traj)
            for sm in self.samplers:
                sm.carry_over()
        if stats is not None:
            self.training_iteration_since_resume += 1
            if self.msg_handlers or len(self.policy_msg_handlers) > 1:  # learner report -> registered handlers
                self._process_msg({k: v for k, v in stats.items() if k != POLICY_ID_KEY} | {POLICY_ID_KEY: 0})
            self._observers_call("on_training_step", self, self.training_iteration_since_resume)
        return stats

    def iteration_async(self):
        """Asynchronous APPO as stream-level overlap (the reference's async mode is process-level: batcher.py:214-218,
        inference_worker.py:175-181).  Iteration k enqueues rollout k on the rollout stream (reading weight snapshot
        k % 2, i.e. the weights after train(k-2): policy lag of one dataset, recorded in policy_version as the
        reference does) and then trains on the slab of rollout k-1 on the main stream.  Slab / snapshot hand-offs are
        HIP events; the only host syncs are the learner's own (one per dataset, one per epoch)."""
        k, ac = self.k, self.learner.actor_critic
        cur, prev = self.slabs[k % 2], self.slabs[(k + 1) % 2]
        main = torch.cuda.current_stream()
        with torch.cuda.stream(self.rollout_stream):
            if k >= 2:
                self.rollout_stream.wait_event(self.ev_train[k % 2])   # the learner is done with this slab ...
            if k >= 1:
                self.rollout_stream.wait_event(self.ev_publish)        # ... and snapshot k % 2 is complete
            ac.snap_read = k % 2
            self._rollout_all(self.published_version, slab=cur, carry_from=prev if k >= 1 else None)
            self.ev_rollout[k % 2].record(self.rollout_stream)
        stats = None
        if k >= 1:
            main.wait_event(self.ev_rollout[(k + 1) % 2])
            stats = self.learner.train(prev)
            ac.publish_weights((k + 1) % 2)                            # read by rollout k+1
            self.ev_publish.record(main)
            self.ev_train[(k + 1) % 2].record(main)
            self.published_version = float(self.learner.train_step)
        self.k += 1
        return stats

    def _report(self, t0: float) -> None:
        """the periodic console report of the reference (runner.py:314-346) + observers' extra_summaries"""
        self._emit_episodic_stats()
        fps = self.learner.env_steps / max(1e-9, time.time() - t0)
        if self.rank == 0:
            print(f"Fps is {fps:.1f}. Total num frames: {self.learner.env_steps}.")
            if "reward" in self.policy_avg_stats and len(self.policy_avg_stats["reward"][0]):
                avg = float(np.mean(self.policy_avg_stats["reward"][0]))
                print("Avg episode reward: %r" % [(0, f"{avg:.3f}")])
        self._observers_call("extra_summaries", self, 0, int(self.learner.env_steps), None)

    def run(self) -> int:
        cfg = self.cfg
        t0 = time.time()
        last_report = t0
        self._observers_call("on_start", self)
        while self.learner.env_steps < cfg.train_for_env_steps and time.time() - t0 < cfg.train_for_seconds:
            self.iteration()
            if time.time() - last_report >= self.report_interval_sec:
                last_report = time.time()
                self._report(t0)
        torch.cuda.synchronize()
        self.env_steps = self.learner.env_steps
        self.fps = self.env_steps / max(1e-9, time.time() - t0)
        self._emit_episodic_stats()
        self._observers_call("on_stop", self)
        if self.rank == 0:
            print(f"Collected {{0: {self.env_steps}}}, FPS: {self.fps:.1f}")
        return self.status


def make_runner(cfg) -> Tuple[object, Runner]:
    return cfg, Runner(cfg)


def run_rl(cfg):
    cfg, runner = make_runner(cfg)
    status = runner.init()
    if status == ExperimentStatus.SUCCESS:
        status = runner.run()
    return status

"""`run_rl(cfg)` / `make_runner(cfg)` — the entry points of sample_factory/train.py:12-41 for the native engine.

One process per GPU owns env, inference, GAE and learning (the design the reference itself recommends for GPU envs:
docs/07-advanced-topics/profiling.md:178-180, serial_mode + sync in sf_examples/brax/train_brax.py:200-201).  With
torchrun (WORLD_SIZE>1) each rank runs a replica on its own env shard and gradients are all-reduced over RCCL.
"""
from __future__ import annotations

import json
import os
import threading
import time
from collections import deque
from typing import Any, Callable, Dict, List, Tuple

import numpy as np

import torch

from sample_factory_amd.algo.learning.learner import Learner, ParameterServer
from sample_factory_amd.algo.sampling.batched_sampling import BatchedVectorEnvRunner
from sample_factory_amd.algo.utils.env_info import extract_env_info
from sample_factory_amd.algo.learning.batcher import Batcher
from sample_factory_amd.algo.utils.shared_buffers import BufferMgr
from sample_factory_amd.cfg.arguments import cfg_dict, preprocess_cfg
from sample_factory_amd.envs.env_utils import create_env, find_training_info_interface, set_training_info
from sample_factory_amd.algo.sampling.parallel_env import ParallelVecEnvView
from sample_factory_amd.utils.attr_dict import AttrDict
from sample_factory_amd.utils.timing import Timing
from sample_factory_amd.utils.utils import init_file_logger, log


from sample_factory_amd.algo.utils.misc import (EPISODIC, LEARNER_ENV_STEPS, POLICY_ID_KEY, TRAIN_STATS,  # noqa: F401
                                                 ExperimentStatus)


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
        # wall-clock profile of the loop (utils/timing.py; printed as a tree when run() ends, runner.py:731-735).  Around
        # asynchronous launches this is host time spent ENQUEUEING — device times come from HIP events (bench.py) / rocprofv3
        self.timing = Timing("Runner profile")
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
        if self.threaded and self._thread is not None:  # the sampler thread reads + resets between two rounds
            with self._cv:
                self._ep_stats_request = True
                while self._ep_stats_request and self._thread_error is None and self._thread.is_alive():
                    self._cv.wait(0.05)
                tot = self._ep_stats_result if self._ep_stats_result is not None else torch.zeros(3, dtype=torch.float64)
                self._ep_stats_result = None
        else:
            tot = self._read_ep_stats()
        k = float(tot[2])
        if k > 0:
            self._process_msg({EPISODIC: dict(reward=float(tot[0]) / k, len=float(tot[1]) / k, episodes=k),
                               POLICY_ID_KEY: 0})

    # ------------------------------------------------------------------------------------------ experiment directory
    def _handle_restart(self) -> None:
        """runner.py:207-230: resume (default) keeps the directory, restart moves it aside, overwrite removes it"""
        import shutil
        exp_dir = os.path.join(self.cfg.train_dir, self.cfg.experiment)
        if not os.path.isdir(exp_dir) or self.cfg.restart_behavior == "resume":
            return
        if self.cfg.restart_behavior == "restart":
            attempt, old = 0, exp_dir
            while os.path.isdir(old):
                attempt += 1
                old = f"{exp_dir}_old{attempt:04d}"
            shutil.move(exp_dir, old)
        elif self.cfg.restart_behavior == "overwrite":
            shutil.rmtree(exp_dir)
        else:
            raise ValueError(f"Unknown restart behavior {self.cfg.restart_behavior}")

    def _save_cfg(self) -> None:
        """runner.py:497-501: config.json next to the checkpoints (what enjoy / resume read back)"""
        d = os.path.join(self.cfg.train_dir, self.cfg.experiment)
        os.makedirs(d, exist_ok=True)
        out = {k: v for k, v in cfg_dict(self.cfg).items() if isinstance(v, (int, float, str, bool, list, type(None)))}
        if getattr(self, "parallel_envs", None) is not None:  # the file keeps what the user asked for, not the internal
            out["num_workers"], out["num_envs_per_worker"] = out.pop("env_workers"), out.pop("env_instances_per_worker")
            out["worker_num_splits"] = int(getattr(self.cfg, "env_worker_splits_requested", out["worker_num_splits"]))
        with open(os.path.join(d, "config.json"), "w") as f:
            json.dump(out, f, indent=2)

    def _read_ep_stats(self):
        """sum of {return, length, count} over the env instances since the last call, then reset — on the stream the
        rollout kernels add to them on (async: the rollout stream), so no episode is lost between read and reset"""
        st = self.rollout_stream if self.cfg.async_rl else torch.cuda.current_stream()
        with torch.cuda.stream(st):  # read AND reset in the rollout stream's order (.cpu() waits for that stream)
            tot = sum(sm.ep_stats.cpu() for sm in self.samplers)
            for sm in self.samplers:
                sm.ep_stats.zero_()
        return tot

    def init(self) -> int:
        """runner.py:521-541: ExperimentStatus.SUCCESS, or FAILURE for a configuration this engine cannot run (reported
        through the log, as the reference reports invalid configurations) — exceptions are for bugs, not for user input"""
        import copy
        # the internal env layout below (one "worker" whose instances are the splits) lives on a COPY: the object the
        # user handed in keeps what they asked for (the reference's make_runner hands Runner a fresh AttrDict as well)
        self.user_cfg, self.cfg = self.cfg, copy.copy(self.cfg)
        if getattr(self.cfg, "device", "gpu") == "cpu":
            log.error("--device=cpu: this engine has no CPU execution path (the hot path is HIP kernels for gfx950; there "
                      "is deliberately no fallback).  Run with --device=gpu on an MI355X.")
            return ExperimentStatus.FAILURE
        if not torch.cuda.is_available():
            log.error("Runner.init(): no GPU visible. sample_factory_amd has no CPU path.")
            return ExperimentStatus.FAILURE
        from sample_factory_amd import lib
        lib.load()  # the HIP library itself: fails loudly (SfHipError) when it was not built
        try:
            return self._init()
        except BaseException:
            self.close_envs()  # env worker processes must not outlive a failed init
            raise

    def _init(self) -> int:
        cfg = self.cfg
        if self.world > 1:
            if not torch.distributed.is_initialized():
                torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
                torch.distributed.init_process_group(os.environ.get("SF_DP_BACKEND", "nccl"))  # nccl = RCCL on ROCm
            cfg.data_parallel = True
            if getattr(cfg, "synthetic_env0", None) in (None, 0):  # env shard of this replica (weak scaling)
                cfg.synthetic_env0 = (self.rank * getattr(cfg, "synthetic_num_agents", 4096) *
                                      max(1, int(cfg.num_workers) * int(cfg.num_envs_per_worker)))
        if self.rank == 0:
            self._handle_restart()
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            torch.distributed.barrier()  # nobody looks for checkpoints before rank 0 has dealt with the directory
        # Env instances: num_workers * num_envs_per_worker vector envs, as in the reference (rollout_worker.py:96-117,
        # each with its own env_config); here they all live in this process.  The instances of one worker are grouped
        # into cfg.worker_num_splits SAMPLING UNITS (the reference's double-buffered sampling); a unit fills one free
        # slice of the slab per rollout and the units' rollouts run on worker_num_splits HIP streams, which fills the
        # tails of the small-batch kernels: measured in tools/split_probe.py and DESIGN.md §5.
        E = max(1, int(cfg.num_workers) * int(cfg.num_envs_per_worker))
        self.parallel_envs = None
        plan = self._env_plan()
        if plan != "direct":
            # HOST envs in worker PROCESSES (rollout_worker.py:79-308): cfg.num_workers processes x cfg.num_envs_per_worker
            # instances, dealt to cfg.worker_num_splits splits; every split is one batched host env for the code below
            from sample_factory_amd.algo.sampling.parallel_env import ParallelHostEnvs
            from sample_factory_amd.envs.env_utils import registered_env_factory
            S = max(1, int(cfg.worker_num_splits))
            if int(cfg.num_envs_per_worker) % S != 0:
                S = 1
            self.parallel_envs = ParallelHostEnvs(cfg, cfg.env, registered_env_factory(cfg.env), int(cfg.num_workers),
                                                  int(cfg.num_envs_per_worker), num_splits=S, inline=plan == "inline",
                                                  probed=getattr(self, "_probed", None))
            self.parallel_envs.register_with_device()
            self.envs = list(self.parallel_envs.views)
            # from here on: one "worker" whose env instances are the splits (slab rows, sampling units, streams follow)
            cfg.env_workers, cfg.env_instances_per_worker = int(cfg.num_workers), int(cfg.num_envs_per_worker)
            cfg.env_worker_splits_requested = int(cfg.worker_num_splits)
            cfg.num_workers, cfg.num_envs_per_worker, cfg.worker_num_splits = 1, S, S
            E = S
        else:
            probe = getattr(self, "_probe_env", None)
            self.envs = [probe if (e == 0 and probe is not None) else
                         create_env(cfg.env, cfg, AttrDict(worker_index=e // int(cfg.num_envs_per_worker),
                                                           vector_index=e % int(cfg.num_envs_per_worker), env_id=e))
                         for e in range(E)]
        self.env = self.envs[0]
        self.env_info = extract_env_info(self.env, cfg)
        n = self.env_info.num_agents
        if any(env.num_agents != n for env in self.envs):
            raise ValueError("all env instances must have the same number of agents")
        if int(cfg.num_envs_per_worker) % int(cfg.worker_num_splits) != 0:
            cfg.worker_num_splits = 1  # (the reference rejects this; one unit per worker is the natural reading)
        if not preprocess_cfg(cfg, self.env_info):
            self.close_envs()
            return ExperimentStatus.FAILURE
        self.user_cfg.recurrence = cfg.recurrence  # (the reference resolves recurrence=-1 on the user's object too)
        if self.rank == 0:
            self._save_cfg()
            init_file_logger(cfg)
        dev = torch.device("cuda", torch.cuda.current_device())
        # Slab + free-slice queue + policy versions (shared_buffers.py:152-239).  Rows: agents x env instances, twice
        # that when rollouts overlap training (async), never fewer than the learner's dataset(s).
        cfg.batched_sampling = True  # this engine only has the batched sampler
        self.buffer_mgr = BufferMgr(cfg, self.env_info, dev)
        self.batcher = Batcher(self.buffer_mgr, cfg)
        self.traj = self.buffer_mgr.traj_tensors
        self.policy_versions = self.buffer_mgr.policy_versions
        self.learner = Learner(cfg, self.env_info, self.policy_versions, 0, ParameterServer(0, self.policy_versions))
        self.learner.init()
        self.num_rows = E * n  # trajectories per sampling round (all units)
        # sampling units: (worker, split) -> env instances [e0, e0 + per_unit)
        per_unit = int(cfg.num_envs_per_worker) // int(cfg.worker_num_splits)
        self.unit_rows = per_unit * n
        assert self.unit_rows == self.buffer_mgr.sampling_trajectories_per_iteration
        self.units = [list(range(e0, e0 + per_unit)) for e0 in range(0, E, per_unit)]
        # one sampling stream for the whole job: key = (seed, global env row) -> a G-replica (or G-instance) rollout is
        # bit-identical to the single-replica rollout of the concatenated env set
        self.samplers = [BatchedVectorEnvRunner(cfg, self.env_info, env, self.learner.actor_critic,
                                                self.traj[e * n:(e + 1) * n], 0, self.policy_versions,
                                                sample_seed=(cfg.seed or 0), row0=self.rank * self.num_rows + e * n,
                                                tag="inf" if e == 0 else f"inf{e}")
                         for e, env in enumerate(self.envs)]
        self.sampler = self.samplers[0]
        self._prev_rows: List = [None] * E          # slab view each env instance wrote its previous rollout into
        S = max(1, min(int(cfg.worker_num_splits), E))
        # SF_ROLLOUT_PRIORITY=1 (async mode): sampling streams on the HIGH hardware-queue priority.  Off by default: over
        # 3 x 100-step alternations on the host-env line it measured 478 vs 488 k env-steps/s (profiles/r04_d_c3_priority_ab*).
        prio = -1 if (cfg.async_rl and os.environ.get("SF_ROLLOUT_PRIORITY", "0") == "1") else 0
        self.split_streams = [torch.cuda.Stream(priority=prio) for _ in range(S)] if E > 1 else None
        self._ev_fork = torch.cuda.Event()
        self._ev_join = [torch.cuda.Event() for _ in range(S)]
        # curricula: envs in this process through their TrainingInfoInterface layer, envs behind a ParallelVecEnvView
        # through the view (which forwards the dict to the worker processes)
        self._training_info_ifaces = [env if isinstance(env, ParallelVecEnvView) else find_training_info_interface(env)
                                      for env in self.envs]
        self._cv = threading.Condition(threading.RLock())   # guards queue / merger / event maps / publish state
        self._ready: List[slice] = []               # complete datasets waiting for the learner
        self._round_events: Dict[int, torch.cuda.Event] = {}   # sampling slice start -> "rollout written" event
        self._free_events: Dict[int, torch.cuda.Event] = {}    # sampling slice start -> "learner done with rows" event
        self.sampling_rounds = 0
        if cfg.async_rl:  # rollouts overlap Learner.train: second stream, published weight snapshots
            self.rollout_stream = torch.cuda.Stream(priority=prio)
            self.ev_publish = torch.cuda.Event()
            self.learner.actor_critic.enable_weight_snapshots()
            self._pub_slot = 0                      # snapshot slot holding the most recently published weights
            self._slot_last_read = [None, None]     # last sampling-round event that read each slot
            self._reading_slot = None               # slot the round being enqueued right now reads (sampler thread)
            self.published_version = float(self.learner.train_step)
        # A HOST env blocks the Python thread once per step (it needs the actions), so stream-level overlap alone would
        # serialise sampling and learning; with a sampler THREAD (the reference's rollout-worker / learner-worker split,
        # rollout_worker.py:79-308, learner_worker.py:50-164) the learner's launches are issued while the sampler waits
        # for its env.  cfg.sampler_thread: None = automatic (async mode with a host env), True / False = forced.
        for sm in self.samplers:
            sm.reset()  # first observation into slab obs[:, 0]; tells us whether the env lives on the host
        if cfg.async_rl:  # the resets (reset kernel / H2D into obs[:, 0]) ran on THIS stream; round 0 samples on another
            ev_reset = torch.cuda.Event()
            ev_reset.record()
            self.rollout_stream.wait_event(ev_reset)
        want = getattr(cfg, "sampler_thread", None)
        self.threaded = bool(cfg.async_rl and (want if want is not None else any(sm.host_env for sm in self.samplers)))
        self._thread, self._stop, self._thread_error = None, False, None
        self._ep_stats_request, self._ep_stats_result = False, None
        self._observers_call("on_init", self)
        self._observers_call("on_connect_components", self)
        return ExperimentStatus.SUCCESS

    def _env_plan(self) -> str:
        """"process": host envs in cfg.num_workers worker processes (the reference's deployment for serial_mode=False);
        "inline": single-agent gym-style envs stepped in this process behind one batched view (serial_mode=True, e.g.
        BASELINE configs[0]'s CartPole copies); "direct": the env instances are batched vector envs used as they are
        (device-resident envs always; batched host envs in serial mode).  cfg.env_workers_mode = "process" | "inline"
        overrides the serial_mode rule ("auto")."""
        from sample_factory_amd.algo.sampling.parallel_env import env_is_batched, probe_info
        cfg = self.cfg
        mode = getattr(cfg, "env_workers_mode", None) or "auto"
        if mode == "process":
            return "process"
        # ask instance 0 what kind of env this is (kept and reused as instance 0 when the answer is "direct")
        probe = create_env(cfg.env, cfg, AttrDict(worker_index=0, vector_index=0, env_id=0))
        device_env = hasattr(probe, "step_into")
        if not device_env and env_is_batched(probe):
            o, _ = probe.reset()
            first = next(iter(o.values())) if isinstance(o, dict) else o
            device_env = isinstance(first, torch.Tensor) and first.is_cuda
        in_process = mode == "inline" or bool(cfg.serial_mode)
        if device_env or (in_process and env_is_batched(probe)):
            self._probe_env = probe
            return "direct"
        self._probed = probe_info(probe)  # ParallelHostEnvs does not have to build a second instance for the spaces
        try:
            probe.close()
        except Exception:  # noqa: BLE001
            pass
        return "inline" if in_process else "process"

    @property
    def slabs(self):
        """the row blocks of the slab one sampling round fills (async mode: two of them in flight)"""
        R = self.num_rows
        return [self.traj[i:i + R] for i in range(0, self.buffer_mgr.num_buffers, R)]

    # ------------------------------------------------------------------------------------------ sampling
    def _acquire_round(self):
        """one free slab slice per sampling unit, or None if the slab has no room for a whole round (every row is
        with the learner: the sampler pauses, inference_worker.py:175-181)"""
        with self._cv:
            if len(self.buffer_mgr.traj_buffer_queue) < len(self.units):
                return None
            return [self.buffer_mgr.get_free_slice() for _ in self.units]

    def _rollout_all(self, policy_version: float, slices=None) -> List[slice]:
        """One sampling round: every unit rolls its env instances out into its slice of the slab, enqueued behind
        everything already on the current stream; returns with the current stream waiting for all of them."""
        n = self.env_info.num_agents
        if slices is None:
            slices = self._acquire_round()
            assert slices is not None, "no free trajectory rows"
        for iface in self._training_info_ifaces:  # curricula: batched_sampling.py:352-355
            set_training_info(iface, dict(approx_total_training_steps=int(self.learner.env_steps)))
        cur = torch.cuda.current_stream()
        for unit, sl in zip(self.units, slices):
            with self._cv:
                ev = self._free_events.pop(sl.start, None)
            if ev is not None:
                cur.wait_event(ev)  # the learner has finished reading these rows
            for j, e in enumerate(unit):
                rows = self.traj[sl.start + j * n: sl.start + (j + 1) * n]
                self.samplers[e].set_slab(rows, carry_from=self._prev_rows[e])
                self._prev_rows[e] = rows
        if self.split_streams is None:
            self.sampler.rollout(policy_version=policy_version)
        else:
            base, S = cur, len(self.split_streams)
            for sm in self.samplers:
                sm.begin_rollout(policy_version)  # (first call: env reset into slab obs[:, 0], on the base stream)
            self._ev_fork.record(base)
            for st in self.split_streams:
                st.wait_event(self._ev_fork)
            if all(sm.async_env for sm in self.samplers):
                # envs stepped by worker processes: software pipeline over the splits — while the workers step split A's
                # envs (begin(t) sent the actions) the GPU runs split B's inference (rollout_worker.py:96-117)
                for e, sm in enumerate(self.samplers):
                    with torch.cuda.stream(self.split_streams[e % S]):
                        sm.rollout_step_begin(0)
                for t in range(self.cfg.rollout):
                    for e, sm in enumerate(self.samplers):
                        with torch.cuda.stream(self.split_streams[e % S]):
                            sm.rollout_step_finish(t)
                            if t + 1 < self.cfg.rollout:
                                sm.rollout_step_begin(t + 1)
            else:
                for t in range(self.cfg.rollout):  # steps of the groups interleaved on the host, concurrent on the device
                    for e, sm in enumerate(self.samplers):
                        with torch.cuda.stream(self.split_streams[e % S]):
                            sm.rollout_step(t)
            for i, st in enumerate(self.split_streams):
                self._ev_join[i].record(st)
                base.wait_event(self._ev_join[i])
        ev = torch.cuda.Event()
        ev.record(cur)
        with self._cv:
            for sl in slices:
                self._round_events[sl.start] = ev
                self._ready += self.batcher.on_new_trajectories(sl)   # row ledger: adjacent slices -> datasets
            self.sampling_rounds += 1
            self._last_round_event = ev
            self._cv.notify_all()
        return slices

    def episode_stats(self):
        """episode statistics over all env instances"""
        if self.cfg.async_rl:
            self.rollout_stream.synchronize()
        tot = sum(sm.ep_stats.cpu() for sm in self.samplers)
        k = float(tot[2])
        return dict(episodes=k, mean_return=float(tot[0]) / k if k else 0.0, mean_len=float(tot[1]) / k if k else 0.0)

    # ------------------------------------------------------------------------------------------ learning
    def _train_dataset(self, ds: slice):
        """Learner.train on slab rows [ds) IN PLACE (the reference copies them into a training batch first,
        batcher.py:192-212), then hand the rows back: training slice -> row ledger -> free sampling slices"""
        main = torch.cuda.current_stream()
        unit = self.unit_rows
        with self._cv:
            evs = [self._round_events.get(start) for start in range(ds.start - ds.start % unit, ds.stop, unit)]
        for ev in evs:
            if ev is not None:
                main.wait_event(ev)
        stats = self.learner.train(self.traj[ds])
        ev = torch.cuda.Event()
        ev.record(main)
        with self._cv:
            for start in range(ds.start - ds.start % unit, ds.stop, unit):
                self._free_events[start] = ev
            self.batcher.on_training_batch_released(ds)
            self._ready += self.batcher.ready_batches()  # a dataset that was waiting for a free training batch
            self._cv.notify_all()
        if stats is not None:
            self.training_iteration_since_resume += 1
            if self.msg_handlers or len(self.policy_msg_handlers) > 1:  # learner report -> registered handlers
                self._process_msg({k: v for k, v in stats.items() if k != POLICY_ID_KEY} | {POLICY_ID_KEY: 0})
            self._observers_call("on_training_step", self, self.training_iteration_since_resume)
        return stats

    def iteration(self):
        """Synchronous APPO (rollout_worker.py:108-126): sampling rounds until a dataset is complete (one round when
        batch_size * num_batches_per_epoch == agents * rollout, k rounds when it is k times that), then the learner
        trains on every complete dataset (several per round when a dataset is a fraction of a round) while the
        sampler waits: zero policy lag at the first minibatch."""
        if self.cfg.async_rl:
            return self.iteration_async()
        while not self._ready:
            with self.timing.add_time("rollout"):
                self._rollout_all(float(self.learner.train_step))
        stats = None
        while self._ready:
            with self.timing.add_time("train"):
                stats = self._train_dataset(self._ready.pop(0)) or stats
        return stats

    def iteration_async(self):
        """Asynchronous APPO as stream-level overlap (the reference's async mode is process-level: batcher.py:214-218,
        inference_worker.py:175-181).  An iteration enqueues one sampling round on the rollout stream — into free slab
        slices, reading the most recently PUBLISHED weight snapshot (policy lag recorded in policy_version as the
        reference does) — and then trains, on the main stream, on the datasets that were complete BEFORE this round:
        rollout k+1 overlaps train(k).  The sampler pauses when the slab has no free slice (every row is with the
        learner or waiting in one of the num_batches_to_accumulate datasets).  Hand-offs are HIP events; the only host
        syncs are the learner's own (one per dataset, one per epoch)."""
        if self.threaded:
            return self._iteration_threaded()
        ac = self.learner.actor_critic
        main = torch.cuda.current_stream()
        todo, self._ready = self._ready, []
        slices = self._acquire_round()
        if slices is not None:
            with torch.cuda.stream(self.rollout_stream):
                if self.sampling_rounds >= 1:
                    self.rollout_stream.wait_event(self.ev_publish)    # the snapshot being read is complete
                ac.snap_read = self._pub_slot
                self._rollout_all(self.published_version, slices)
                self._slot_last_read[self._pub_slot] = self._last_round_event
        stats = None
        for ds in todo:
            stats = self._train_dataset(ds) or stats
        if todo:
            self._publish()
        elif slices is None:
            raise RuntimeError("async iteration made no progress: no free rows and no complete dataset "
                               "(slab smaller than one dataset?)")
        return stats

    def _publish(self) -> None:
        """K20 in async mode: copy the learner's weights into the snapshot slot no sampling round is reading (the one
        being enqueued right now reads _reading_slot; finished rounds are waited for through their events)"""
        main = torch.cuda.current_stream()
        with self._cv:
            slot = 1 - (self._reading_slot if self._reading_slot is not None else self._pub_slot)
            if self._slot_last_read[slot] is not None:
                main.wait_event(self._slot_last_read[slot])            # its last reader has finished
            self.learner.actor_critic.publish_weights(slot)
            ev = torch.cuda.Event()
            ev.record(main)
            self.ev_publish = ev
            self._pub_slot = slot
            self.published_version = float(self.learner.train_step)

    # ---- sampler thread (host envs in async mode)
    def _sampler_loop(self) -> None:
        try:
            torch.cuda.set_device(self.learner.device)
            ac = self.learner.actor_critic
            with torch.cuda.stream(self.rollout_stream):
                while True:
                    with self._cv:
                        while not self._stop and (slices := self._acquire_round()) is None:
                            self._service_ep_stats()
                            self._cv.wait(0.05)   # every row is with the learner: the sampler pauses
                        if self._stop:
                            return
                        slot, version, ev_pub = self._pub_slot, self.published_version, self.ev_publish
                        self._reading_slot = slot
                    if self.sampling_rounds >= 1 or self.learner.train_step > 0:
                        self.rollout_stream.wait_event(ev_pub)
                    ac.snap_read = slot
                    self._rollout_all(version, slices)
                    with self._cv:
                        self._slot_last_read[slot] = self._last_round_event
                        self._reading_slot = None
                        self._service_ep_stats()
                        self._cv.notify_all()
        except BaseException as e:  # surfaced by the learner thread
            with self._cv:
                self._thread_error = e
                self._cv.notify_all()

    def _service_ep_stats(self) -> None:
        """(sampler thread, lock held) the thread that owns the rollout stream reads + resets the episode statistics"""
        if self._ep_stats_request:
            self._ep_stats_result = self._read_ep_stats()
            self._ep_stats_request = False
            self._cv.notify_all()

    def close_envs(self) -> None:
        """stop the env worker processes (a no-op for in-process envs)"""
        if getattr(self, "parallel_envs", None) is not None:
            self.parallel_envs.close()
            self.parallel_envs = None

    def _start_sampler_thread(self) -> None:
        if self._thread is None:
            self._stop = False
            self._thread = threading.Thread(target=self._sampler_loop, name="sf-sampler", daemon=True)
            self._thread.start()

    def stop_sampler_thread(self) -> None:
        if self._thread is not None:
            with self._cv:
                self._stop = True
                self._cv.notify_all()
            self._thread.join()
            self._thread = None
            self.rollout_stream.synchronize()

    def _iteration_threaded(self):
        """learner side of the threaded async mode: take the next complete dataset (waiting for the sampler thread if
        there is none), train, publish"""
        self._start_sampler_thread()
        with self._cv:
            while not self._ready:
                if self._thread_error is not None:
                    raise RuntimeError("sampler thread died") from self._thread_error
                self._cv.wait(0.05)
            ds = self._ready.pop(0)
        stats = self._train_dataset(ds)
        self._publish()
        return stats

    # ------------------------------------------------------------------------------------------ reports / checkpoints
    def _report(self, t0: float) -> None:
        """the periodic console report of the reference (runner.py:314-346) + observers' extra_summaries"""
        self._emit_episodic_stats()
        fps = (self.learner.env_steps - self._env_steps0) / max(1e-9, time.time() - t0)
        if self.rank == 0:
            print(f"Fps is {fps:.1f}. Total num frames: {self.learner.env_steps}.")
            if "reward" in self.policy_avg_stats and len(self.policy_avg_stats["reward"][0]):
                avg = float(np.mean(self.policy_avg_stats["reward"][0]))
                print("Avg episode reward: %r" % [(0, f"{avg:.3f}")])
        self._write_summaries(fps)
        self._observers_call("extra_summaries", self, 0, int(self.learner.env_steps), None)

    def _write_summaries(self, fps: float) -> None:
        """Summary sink.  The reference writes tensorboard events under <experiment>/.summary/<policy_id>/
        (runner.py:298-346, learner.py:843-923); tensorboardX is not a dependency here, so the same scalars — perf/_fps,
        policy_stats/avg_<key>, train/* of the learner's last report — go to .summary/0/summaries.jsonl, one JSON object
        per report with env_steps as the x value (rank 0 only)."""
        if self.rank != 0:
            return
        d = os.path.join(self.cfg.train_dir, self.cfg.experiment, ".summary", "0")
        os.makedirs(d, exist_ok=True)
        rec = {"env_steps": int(self.learner.env_steps), "time": time.time(), "perf/_fps": float(fps),
               "train_step": int(self.learner.train_step)}
        for key, per_policy in self.policy_avg_stats.items():
            if len(per_policy[0]):
                rec[f"policy_stats/avg_{key}"] = float(np.mean(per_policy[0]))
        for k, v in dict(self.learner.last_summary).items():
            if isinstance(v, (int, float)):
                rec[f"train/{k}"] = float(v)
        with open(os.path.join(d, "summaries.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")

    def _save_policy(self) -> None:
        """runner.py:453-454 -> learner.save(); rank 0 only: replicas hold identical weights and share the directory"""
        if self.rank == 0:
            self.learner.save()

    def _save_milestone_policy(self) -> None:
        if self.rank == 0:
            self.learner.save_milestone()

    def _save_best_policy(self) -> None:
        """runner.py:459-475: once save_best_after env steps are in, save when the running average of
        cfg.save_best_metric improved (the threshold test is Learner.save_best's)"""
        metric = self.cfg.save_best_metric
        if self.rank != 0 or metric not in self.policy_avg_stats or self.learner.env_steps < self.cfg.save_best_after:
            return
        stats = self.policy_avg_stats[metric][0]
        if len(stats) > 0:
            self.learner.save_best(0, metric, float(np.mean(stats)))

    def run(self) -> int:
        cfg = self.cfg
        t0 = time.time()
        self._env_steps0 = self.learner.env_steps
        last = dict(report=t0, save=t0, best=t0, milestone=t0)
        self._observers_call("on_start", self)
        try:
            while self.learner.env_steps < cfg.train_for_env_steps and time.time() - t0 < cfg.train_for_seconds:
                self.iteration()
                now = time.time()
                if now - last["report"] >= self.report_interval_sec:
                    last["report"] = now
                    self._report(t0)
                # the reference's periodic timers (runner.py:170-176), checked between iterations
                if cfg.save_every_sec > 0 and now - last["save"] >= cfg.save_every_sec:
                    last["save"] = now
                    self._save_policy()
                if cfg.save_best_every_sec > 0 and now - last["best"] >= cfg.save_best_every_sec:
                    last["best"] = now
                    self._emit_episodic_stats()
                    self._save_best_policy()
                if cfg.save_milestones_sec > 0 and now - last["milestone"] >= cfg.save_milestones_sec:
                    last["milestone"] = now
                    self._save_milestone_policy()
        except KeyboardInterrupt:
            self.status = ExperimentStatus.INTERRUPTED
        self._emit_episodic_stats()
        self.stop_sampler_thread()
        torch.cuda.synchronize()
        self.close_envs()
        self.env_steps = self.learner.env_steps
        self.fps = (self.env_steps - self._env_steps0) / max(1e-9, time.time() - t0)
        self._write_summaries(self.fps)
        self._observers_call("on_stop", self)
        self._save_policy()        # runner.py:685-698 (_stop_training): final checkpoint + best check
        self._save_best_policy()
        if self.rank == 0:
            log.info(str(self.timing))
            print(f"Collected {{0: {self.env_steps}}}, FPS: {self.fps:.1f}")
        return self.status


def make_runner(cfg) -> Tuple[object, Runner]:
    """train.py:12-30 of the reference: with restart_behavior=resume (the default) the configuration saved in the
    experiment directory is the base and only flags given on the command line override it.  A cfg that did not come from
    parse_full_cfg (no `cli_args`: tests, bench) is used as it is."""
    if getattr(cfg, "restart_behavior", "resume") == "resume" and hasattr(cfg, "cli_args"):
        from sample_factory_amd.cfg.arguments import maybe_load_from_checkpoint
        cfg = maybe_load_from_checkpoint(cfg)
    return cfg, Runner(cfg)


def run_rl(cfg):
    cfg, runner = make_runner(cfg)
    status = runner.init()
    if status == ExperimentStatus.SUCCESS:
        status = runner.run()
    return status

"""Learner — the reference's `Learner` surface (sample_factory/algo/learning/learner.py:125-1067) on HIP kernels.

Same constructor, `init()`, `train(batch)`, `_prepare_batch`, `_get_minibatches`, `_calculate_losses`, `_train`,
`save()/load_from_checkpoint()` and checkpoint format; what changes is what runs underneath:

  reference op sequence                    here
  K13 valids / invalid count / sanitise                      sf_valid_mask           (1 launch)
  K9  bootstrap forward                                       sf_conv_fwd stack on slab[:, T] in place
  K10-K11 de-normalise, value bootstrap, GAE, returns         sf_gae_returns          (1 launch, LDS-staged scan)
  K12 returns normaliser                                      sf_moments + sf_rms_update + sf_rms_apply
  K7/K8 batcher copy + f32 obs materialisation                none: conv1 reads the u8 slab through (index, traj_T)
  K14 minibatch gather                                        sf_minibatch_indices; consumers read through the index
  K15 forward / K18 backward                                  fp32-MFMA implicit GEMM (sf_conv_fwd/wgrad/dgrad)
  K16 loss head fwd+bwd                                       sf_moments + sf_ppo_loss (analytic backward fused)
  K17 V-trace (CPU loop in the reference)                     sf_vtrace on device
  K18-K19 clip_grad_norm_ + Adam                              sf_grad_sumsq + sf_adam_step on the flat buffers
  C1  data-parallel replicas (new)                            torch.distributed all_reduce (RCCL) of the flat grad bucket
                                                              + 3-double moment buckets

Host synchronisation: one readback per dataset (num_invalids, as the reference) and one per epoch (actor losses for
the early-stop test / KL for the LR schedulers) instead of ~6 per minibatch.
"""
from __future__ import annotations

import glob
import time
import os
from os.path import join
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from sample_factory_amd import lib
from sample_factory_amd.algo.learning.dp import ReplicaGroup
from sample_factory_amd.algo.utils.tensor_dict import TensorDict
from sample_factory_amd.envs.spaces import (action_head_sizes, calc_num_action_parameters, calc_num_actions, is_box,
                                             is_discrete)
from sample_factory_amd.model.actor_critic import ActorCritic
from sample_factory_amd.model.model_factory import create_actor_critic
from sample_factory_amd.utils.attr_dict import AttrDict
from sample_factory_amd.utils.decay import LinearDecay

LEARNER_ENV_STEPS, POLICY_ID_KEY, STATS_KEY, TRAIN_STATS = "learner_env_steps", "policy_id", "stats", "train"


# ------------------------------------------------------------------------------------------------ LR schedulers
class LearningRateScheduler:
    """learner.py:35-113"""

    def update(self, current_lr, recent_kls):
        return current_lr

    def invoke_after_each_minibatch(self):
        return False

    def invoke_after_each_epoch(self):
        return False


class KlAdaptiveScheduler(LearningRateScheduler):
    def __init__(self, cfg, per_epoch: bool):
        self.threshold, self.min_lr, self.max_lr = cfg.lr_schedule_kl_threshold, cfg.lr_adaptive_min, cfg.lr_adaptive_max
        self.per_epoch = per_epoch
        self.n = cfg.num_batches_per_epoch if per_epoch else 1

    def update(self, current_lr, recent_kls):
        mean_kl = float(np.mean(recent_kls[-self.n:]))
        lr = current_lr
        if mean_kl > 2.0 * self.threshold:
            lr = max(current_lr / 1.5, self.min_lr)
        if mean_kl < 0.5 * self.threshold:
            lr = min(current_lr * 1.5, self.max_lr)
        return lr

    def invoke_after_each_minibatch(self):
        return not self.per_epoch

    def invoke_after_each_epoch(self):
        return self.per_epoch


class LinearDecayScheduler(LearningRateScheduler):
    def __init__(self, cfg):
        self.num_updates = cfg.train_for_env_steps // cfg.batch_size * cfg.num_epochs
        self.lr0 = cfg.learning_rate
        self.step = 0

    def invoke_after_each_minibatch(self):
        return True

    def update(self, current_lr, recent_kls):
        self.step += 1
        return self.lr0 * max(0.0, 1.0 - self.step / max(1, self.num_updates))


def get_lr_scheduler(cfg) -> LearningRateScheduler:
    if cfg.lr_schedule == "constant":
        return LearningRateScheduler()
    if cfg.lr_schedule == "kl_adaptive_minibatch":
        return KlAdaptiveScheduler(cfg, per_epoch=False)
    if cfg.lr_schedule == "kl_adaptive_epoch":
        return KlAdaptiveScheduler(cfg, per_epoch=True)
    if cfg.lr_schedule == "linear_decay":
        return LinearDecayScheduler(cfg)
    raise RuntimeError(f"Unknown scheduler {cfg.lr_schedule}")


def experiment_dir(cfg) -> str:
    d = join(cfg.train_dir, cfg.experiment)
    os.makedirs(d, exist_ok=True)
    return d


class _NullLock:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class ParameterServer:
    """model_sharing.py:17-43 — same process, same weights: publishing is bumping the version counter (K20)."""

    def __init__(self, policy_id, policy_versions: torch.Tensor, serial_mode: bool = True):
        self.policy_id = policy_id
        self.actor_critic = None
        self.policy_versions = policy_versions
        self.device = None
        self.policy_lock = _NullLock()

    def init(self, actor_critic, policy_version, device):
        self.actor_critic, self.device = actor_critic, device
        self.policy_versions[self.policy_id] = policy_version


class Learner:
    def __init__(self, cfg, env_info, policy_versions_tensor: torch.Tensor, policy_id: int, param_server,
                 process_group=None):
        self.cfg = cfg
        self.env_info = env_info
        self.policy_id = policy_id
        self.policy_versions_tensor = policy_versions_tensor
        self.param_server = param_server
        self.device: Optional[torch.device] = None
        self.actor_critic: Optional[ActorCritic] = None
        self.curr_lr: Optional[float] = None
        self.lr_scheduler: Optional[LearningRateScheduler] = None
        self.train_step = 0
        self.env_steps = 0
        self.best_performance = -1e9
        self.new_cfg: Optional[Dict] = None
        self.policy_to_load = None
        self.is_initialized = False
        self.last_summary: Dict = {}
        # data-parallel replicas (C1): one rank per GPU; inactive group = single GPU
        self.pg = process_group
        dp_on = process_group is not None or (getattr(cfg, "data_parallel", False) and torch.distributed.is_initialized())
        self.group = ReplicaGroup(process_group, bool(getattr(cfg, "dp_force_collectives", False)),
                                  bool(getattr(cfg, "dp_native_rccl", False)),
                                  oneshot_bytes=int(getattr(cfg, "dp_oneshot_bytes", 0) or 0)) if dp_on else None
        self.world = self.group.world if self.group is not None else 1
        self.dp = self.group is not None and self.group.on  # collectives are issued (world > 1, or forced for tests)
        self._grad_norms: List[float] = []

    # ------------------------------------------------------------------------------------------ init / checkpoints
    def _all_reduce(self, t: torch.Tensor) -> None:
        if self.dp:
            self.group.all_reduce_sum(t)

    def init(self):
        cfg = self.cfg
        if cfg.exploration_loss not in ("entropy", "symmetric_kl"):
            raise NotImplementedError(f"{cfg.exploration_loss} not supported!")
        if cfg.optimizer not in ("adam", "lamb"):
            raise RuntimeError(f"Unknown optimizer {cfg.optimizer}")  # learner.py:228-230
        if cfg.seed is not None:
            torch.manual_seed(cfg.seed)
            np.random.seed(cfg.seed)
        if not torch.cuda.is_available():
            raise lib.SfHipError("Learner.init(): no GPU visible. sample_factory_amd has no CPU path.")
        lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device())
        ar = self._all_reduce if self.dp else None
        cfg.dp_world = self.world
        # the native model, or a torch fallback around a user-registered module (model/model_factory.py)
        self.actor_critic = create_actor_critic(cfg, self.env_info.obs_space, self.env_info.action_space, self.device,
                                                all_reduce=ar)
        self.actor_critic.train()
        if self.dp:  # identical initial weights on every replica
            self.group.broadcast(self.actor_critic.flat_params, src=0)
            self.actor_critic.params_changed()
        P = self.actor_critic.num_flat
        self.exp_avg = torch.zeros(P, dtype=torch.float32, device=self.device)
        self.exp_avg_sq = torch.zeros(P, dtype=torch.float32, device=self.device)
        self.adam_step_count = 0
        self._dp_reduce_each_mb = True
        # C1 overlap: layers finish last -> first, so the tail of the flat gradient that starts at the largest layer
        # (Nature-CNN: the 3136x512 fc matrix + heads = 95 % of the bytes) is all-reduced on the collective's own
        # stream while conv3/conv2/conv1 are still being back-propagated; the small head of the buffer follows after the
        # backward pass.  Every element is still summed exactly once over the same replicas.
        self._dp_split = None
        ac_ = self.actor_critic
        if self.dp and getattr(cfg, "dp_overlap", True) and hasattr(ac_, "_segs") and ac_.rnn_kind is None:
            sizes = [L.K * L.N for L in ac_.layers]
            li = int(np.argmax(sizes))
            if li > 0 and sum(sizes[li:]) * 2 >= sum(sizes):
                self._dp_split = li
        self._lamb = None  # (segment ids, #segments, direction scratch, per-segment sums) — built on first use
        # small device scratch
        dev = self.device
        self._num_invalid = torch.zeros(1, dtype=torch.int32, device=dev)
        self._moments = torch.zeros(3, dtype=torch.float64, device=dev)
        self._sums = torch.zeros(8, dtype=torch.float64, device=dev)
        self._sumsq = torch.zeros(1, dtype=torch.float64, device=dev)
        self._scalars = None  # [num_epochs*num_batches, 16] per train() call
        A = calc_num_action_parameters(self.env_info.action_space)
        self.num_action_params = A
        self.num_actions = calc_num_actions(self.env_info.action_space)
        self._refresh_loss_cfg()
        self.load_from_checkpoint(self.policy_id)
        self.param_server.init(self.actor_critic, self.train_step, self.device)
        self.policy_versions_tensor[self.policy_id] = self.train_step
        self.lr_scheduler = get_lr_scheduler(cfg)
        self.curr_lr = cfg.learning_rate if self.curr_lr is None else self.curr_lr
        self.is_initialized = True
        state_dict = None if cfg.serial_mode else self.actor_critic.state_dict()
        return self.policy_id, state_dict, self.device, self.train_step

    @staticmethod
    def checkpoint_dir(cfg, policy_id):
        d = join(experiment_dir(cfg), f"checkpoint_p{policy_id}")
        os.makedirs(d, exist_ok=True)
        return d

    @staticmethod
    def get_checkpoints(checkpoints_dir, pattern="checkpoint_*"):
        return sorted(glob.glob(join(checkpoints_dir, pattern)))

    @staticmethod
    def load_checkpoint(checkpoints, device):
        if not checkpoints:
            return None
        return torch.load(checkpoints[-1], map_location="cpu", weights_only=False)

    def _get_checkpoint_dict(self):
        """learner.py:323-332: {train_step, env_steps, best_performance, model, optimizer, curr_lr}; the optimizer
        state is torch.optim.Adam's state_dict layout over the reference parameter order."""
        ac = self.actor_critic
        names = [n for n, _ in ac.ref_param_shapes()]
        m, v = ac.flat_to_ref(self.exp_avg), ac.flat_to_ref(self.exp_avg_sq)
        opt_state = {i: dict(step=torch.tensor(float(self.adam_step_count)), exp_avg=m[n], exp_avg_sq=v[n])
                     for i, n in enumerate(names)} if self.adam_step_count > 0 else {}
        opt = dict(state=opt_state, param_groups=[dict(
            lr=self.curr_lr, betas=(self.cfg.adam_beta1, self.cfg.adam_beta2), eps=self.cfg.adam_eps, weight_decay=0,
            amsgrad=False, maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
            decoupled_weight_decay=False, params=list(range(len(names))))])
        return dict(train_step=self.train_step, env_steps=self.env_steps, best_performance=self.best_performance,
                    model=ac.state_dict(), optimizer=opt, curr_lr=self.curr_lr)

    def _save_impl(self, name_prefix: str, name_suffix: str, keep_checkpoints: int) -> bool:
        """learner.py:334-360: atomic temp+rename, keep the `keep_checkpoints` newest files of this prefix"""
        if not self.is_initialized:
            return False
        d = self.checkpoint_dir(self.cfg, self.policy_id)
        tmp = join(d, f"{name_prefix}_temp")
        name = join(d, f"{name_prefix}_{self.train_step:09d}_{self.env_steps}{name_suffix}.pth")
        torch.save(self._get_checkpoint_dict(), tmp)
        os.rename(tmp, name)
        while len(cps := self.get_checkpoints(d, f"{name_prefix}_*")) > keep_checkpoints:
            if os.path.isfile(cps[0]):
                os.remove(cps[0])
        return True

    def save(self) -> bool:
        """learner.py:362-363"""
        return self._save_impl("checkpoint", "", self.cfg.keep_checkpoints)

    def save_milestone(self) -> None:
        """learner.py:365-374: a copy under checkpoint_p<id>/milestones/ that is never rotated out"""
        d = join(self.checkpoint_dir(self.cfg, self.policy_id), "milestones")
        os.makedirs(d, exist_ok=True)
        torch.save(self._get_checkpoint_dict(), join(d, f"checkpoint_{self.train_step:09d}_{self.env_steps}.pth"))

    def save_best(self, policy_id, metric, metric_value) -> bool:
        """learner.py:376-386: keep ONE best_* file, replaced when the metric improved by more than 1e-3"""
        if policy_id != self.policy_id:
            return False
        p = 3  # precision, number of significant digits
        if metric_value - self.best_performance > 1 / 10 ** p:
            self.best_performance = metric_value
            return self._save_impl("best", f"_{metric}_{metric_value:.{p}f}", 1)
        return False

    def _load_state(self, cp, load_progress=True):
        """learner.py:289-298"""
        if load_progress:
            self.train_step, self.env_steps = cp["train_step"], cp["env_steps"]
            self.best_performance = cp.get("best_performance", self.best_performance)
        ac = self.actor_critic
        ac.load_state_dict(cp["model"])
        st = cp["optimizer"].get("state", {})
        if st:
            names = [n for n, _ in ac.ref_param_shapes()]
            m = {n: st[i]["exp_avg"] for i, n in enumerate(names)}
            v = {n: st[i]["exp_avg_sq"] for i, n in enumerate(names)}
            self._ref_to_flat(m, self.exp_avg)
            self._ref_to_flat(v, self.exp_avg_sq)
            self.adam_step_count = int(float(st[0]["step"]))
        self.curr_lr = cp.get("curr_lr", self.cfg.learning_rate)

    def _ref_to_flat(self, ref: Dict[str, torch.Tensor], flat: torch.Tensor):
        ac = self.actor_critic
        keep_p = ac.flat_params.clone()
        sd = dict(ref)
        ac.load_state_dict(sd, strict=False)
        flat.copy_(ac.flat_params)
        ac.flat_params.copy_(keep_p)
        ac.params_changed()

    def load_from_checkpoint(self, policy_id, load_progress=True) -> None:
        """learner.py:300-310: the newest checkpoint_* (cfg.load_checkpoint_kind == "latest") or best_* ("best")"""
        prefix = dict(latest="checkpoint", best="best")[self.cfg.load_checkpoint_kind]
        cps = self.get_checkpoints(self.checkpoint_dir(self.cfg, policy_id), pattern=f"{prefix}_*")
        cp = self.load_checkpoint(cps, self.device)
        if cp is not None:
            self._load_state(cp, load_progress)

    # ------------------------------------------------------------------------------------------ PBT hooks
    def set_new_cfg(self, new_cfg: Dict) -> None:
        """learner.py:388-389 — population-based training hands over mutated hyper-parameters"""
        self.new_cfg = new_cfg

    def set_policy_to_load(self, policy_to_load) -> None:
        """learner.py:391-392"""
        self.policy_to_load = policy_to_load

    def _refresh_loss_cfg(self) -> None:
        cfg = self.cfg
        self.loss_cfg = lib.sf_loss_cfg(
            clip_ratio=cfg.ppo_clip_ratio, clip_value=cfg.ppo_clip_value, value_loss_coeff=cfg.value_loss_coeff,
            exploration_coeff=cfg.exploration_loss_coeff, kl_coeff=cfg.kl_loss_coeff,
            exploration_kind=0 if cfg.exploration_loss_coeff == 0.0 else (1 if cfg.exploration_loss == "entropy" else 2),
            action_kind=1 if is_box(self.env_info.action_space) else 0, dense_adv=int(bool(cfg.with_vtrace)))
        heads = action_head_sizes(self.env_info.action_space)
        self._head_sizes = heads
        if len(heads) > 1:  # Tuple of Discrete spaces: independent categorical heads
            self.loss_cfg.num_heads = len(heads)
            for i, nh in enumerate(heads):
                self.loss_cfg.head_n[i] = nh

    def _maybe_update_cfg(self) -> None:
        """learner.py:394-413: apply the new values; a PBT-optimised learning rate only with the constant schedule.
        Loss coefficients and Adam betas are read from cfg at every SGD step, so only the packed loss struct and
        curr_lr need refreshing."""
        if self.new_cfg is None:
            return
        for key, value in self.new_cfg.items():
            setattr(self.cfg, key, value)
        if self.cfg.lr_schedule == "constant" and self.curr_lr != self.cfg.learning_rate:
            self.curr_lr = self.cfg.learning_rate
        self._refresh_loss_cfg()
        self.new_cfg = None

    def _maybe_load_policy(self) -> None:
        """learner.py:415-428: take another policy's weights (not its progress) and invalidate in-flight experience"""
        if self.policy_to_load is None:
            return
        self.load_from_checkpoint(self.policy_to_load, load_progress=False)
        self.train_step += self.cfg.max_policy_lag + 1
        self.policy_versions_tensor[self.policy_id] = self.train_step
        self.policy_to_load = None

    # ------------------------------------------------------------------------------------------ batch preparation
    def _prepare_batch(self, batch: TensorDict) -> Tuple[AttrDict, int, int]:
        """learner.py:943-1034.  `batch` is the env-major trajectory slab slice on the GPU; it is mutated in place
        exactly where the reference mutates it (valids, values[:, -1], rewards under value_bootstrap, sanitised
        actions/log-probs).  Returns (buff, experience_size, num_invalids); buff holds FLAT [E*T] dataset arrays."""
        cfg, ac = self.cfg, self.actor_critic
        # several observation keys (torch fallback model, model/torch_policy.py): the model takes {key: slab view}
        multi = getattr(ac, "multi_key", False)
        obs = {k: batch["obs"][k] for k in ac.obs_keys} if multi else batch["obs"]["obs"]
        E, T = batch["rewards"].shape
        N = E * T
        # valids [E, T+1] (the batch is mutated as the reference mutates it) and, from the same launch, the flat [E*T]
        # dataset mask the minibatch consumers index (the reference's `[:, :-1]` + flatten is a copy)
        valids_flat = ac._buf(("prep", "valids_flat"), (N,), dtype=torch.bool)
        lib.valid_mask(batch["policy_id"], batch["policy_version"], batch["valids"], batch["actions"],
                       self.num_actions, batch["log_prob_actions"], self.policy_id, self.train_step,
                       cfg.max_policy_lag, self._num_invalid, valids_flat=valids_flat)
        if not ac.training:
            ac.train()
        if ac.obs_normalizer is not None:  # learner.py:957-961: statistics updated once per dataset, over all T+1 columns
            ac.obs_normalizer.update(obs, ac.obs_elems, E * (T + 1))
        # K9: bootstrap value of the T+1-th observation, read from the slab in place
        last = {k: v[:, T] for k, v in obs.items()} if multi else obs[:, T]
        rnn = dict(states=batch["rnn_states"][:, T]) if cfg.use_rnn else None
        heads = ac.forward_heads(last, E, sample_stride=0 if multi else obs.stride(0), tag="boot", rnn=rnn)[-1]  # learner weights
        lib.copy_rows(batch["values"][:, T], heads[:, 0])
        adv = torch.empty((E, T), dtype=torch.float32, device=self.device)
        ret = torch.empty((E, T), dtype=torch.float32, device=self.device)
        buff = AttrDict()
        if not cfg.with_vtrace:
            rms = ac.returns_normalizer.stats if cfg.normalize_returns else None
            lib.gae_returns(batch["rewards"], batch["dones"], batch["time_outs"], batch["values"], batch["valids"],
                            rms, cfg.gamma, cfg.gae_lambda, cfg.value_bootstrap, adv, ret)
            buff.advantages, buff.returns = adv.view(N), ret.view(N)
        elif cfg.value_bootstrap:
            # learner.py:980-990 runs before (and independently of) the advantage estimator: rewards += gamma * v(t) on
            # timed-out terminal steps.  The GAE launch performs exactly that in-place update; V-trace recomputes the
            # advantages per minibatch, so its adv/ret outputs are scratch here (normalize_returns is off with V-trace).
            lib.gae_returns(batch["rewards"], batch["dones"], batch["time_outs"], batch["values"], batch["valids"],
                            None, cfg.gamma, cfg.gae_lambda, True, adv, ret)
        # flat dataset views (index e*T + t).  The two [E, T+1] arrays are NOT compacted: the flat mask came out of the
        # sf_valid_mask launch, the old values are read in place through the loss kernel's row mapping (old_values_T)
        buff.obs = obs
        buff.actions = batch["actions"].view(N, self.num_actions)
        buff.action_logits = batch["action_logits"].view(N, self.num_action_params)
        buff.log_prob_actions = batch["log_prob_actions"].view(N)
        buff.rewards = batch["rewards"].view(N)
        buff.dones = batch["dones"].view(N)
        buff.policy_id, buff.policy_version = batch["policy_id"].view(N), batch["policy_version"].view(N)
        if cfg.use_rnn:
            buff.rnn_states = batch["rnn_states"]  # the slab [E, T+1, S], read in place by sf_rnn_chunk_setup
        buff["values"] = batch["values"]  # [E, T+1], row e*T+t at e*(T+1)+t; NB: item access, AttrDict.values is dict.values
        buff.valids = valids_flat
        buff.E, buff.T = E, T
        if cfg.normalize_returns and not cfg.with_vtrace:
            ac.returns_normalizer(buff.returns)  # in place: update (all-reduced moments under DP) + normalise
        num_invalids = int(self._num_invalid.item())  # the one host sync per dataset (reference: learner.py:1021)
        if self.dp:
            t = torch.tensor([num_invalids], dtype=torch.int64, device=self.device)
            self._all_reduce(t)
            self._global_invalids = int(t.item())
        else:
            self._global_invalids = num_invalids
        return buff, N, num_invalids

    # ------------------------------------------------------------------------------------------ minibatches
    def _get_minibatches(self, batch_size, experience_size):
        """learner.py:498-526.  Returns a list of (index_tensor|None, offset, n); nothing is gathered, the consumers
        read through the index.  shuffle_minibatches: the permutation of the recurrence-aligned chunk starts is the
        REFERENCE's — host np.random.permutation from the global stream seeded in init() as learner.py:199-204 does —
        uploaded (4 B per chunk) and expanded on the device (sf_minibatch_expand), so the index sets are equal to the
        reference's element for element.  cfg.device_shuffle=True keeps everything on the device instead (stateless
        Feistel permutation, sf_minibatch_indices: no host RNG, no upload; a different but equally valid shuffle)."""
        cfg = self.cfg
        assert cfg.rollout % cfg.recurrence == 0
        assert experience_size % batch_size == 0, f"experience size: {experience_size}, batch size: {batch_size}"
        n_mb = cfg.num_batches_per_epoch
        if n_mb == 1:
            return [(None, 0, experience_size)]
        # a dataset is exactly one training iteration (the Runner's Batcher hands over nothing else): an explicit n at
        # an offset past the end would be an out-of-bounds device read, a larger dataset would silently skip its tail
        assert n_mb * batch_size == experience_size, \
            f"dataset of {experience_size} samples != num_batches_per_epoch {n_mb} x batch_size {batch_size}"
        if cfg.shuffle_minibatches:
            idx = torch.empty(experience_size, dtype=torch.int32, device=self.device)
            if getattr(cfg, "device_shuffle", False):
                seed = (cfg.seed or 0) * 7919 + self.policy_id
                lib.minibatch_indices(idx, experience_size, cfg.recurrence, True, seed, self._shuffle_epoch)
                self._shuffle_epoch += 1
            else:
                starts = np.random.permutation(np.arange(0, experience_size, cfg.recurrence))  # learner.py:509-510
                starts_dev = torch.from_numpy(starts.astype(np.int32)).to(self.device, non_blocking=True)
                lib.minibatch_expand(starts_dev, idx, experience_size, cfg.recurrence)
            return [(idx[i * batch_size:(i + 1) * batch_size], 0, batch_size) for i in range(experience_size // batch_size)]
        return [(None, i * batch_size, batch_size) for i in range(n_mb)]

    _shuffle_epoch = 0

    # ------------------------------------------------------------------------------------------ losses
    def _calculate_losses(self, mb: AttrDict, num_invalids: int):
        """The reference's signature and return value (learner.py:537-669):
        `(action_distribution, policy_loss, exploration_loss, kl_old, kl_loss, value_loss, loss_summaries)` for the
        minibatch `mb` — a dataset as returned by `_prepare_batch` (the whole of it is the minibatch, as in
        tests/algo/test_learner.py:21-39 of the reference) or `(dataset, (index|None, offset, n))` for a part of it.
        The losses are 0-dim device tensors produced by the fused HIP loss kernel.  `action_distribution` is the
        reference's object type for the action space (algo/utils/action_distributions.py: Categorical / Tuple /
        Continuous over the network's action parameters, + `.values`), `kl_old` the per-sample KL(new || old) over the
        VALID samples (learner.py:460-470: `masked_select`), whose mean the kernel's `kl` scalar is.  The training loop
        itself uses `_losses_native` (same kernels, no tensor unpacking, gradient at the loss heads kept)."""
        from sample_factory_amd.algo.utils.action_distributions import get_action_distribution
        buff, part = (mb if isinstance(mb, tuple) else (mb, None))
        if part is None:
            part = (None, 0, buff.E * buff.T)
        acts, g_heads, sc = self._losses_native(buff, part, num_invalids)
        index, offset, n = part
        heads = acts[-1]
        A = self.num_action_params
        space = self.env_info.action_space
        dist = get_action_distribution(space, heads[:n, 1:1 + A])
        dist.raw_logits, dist.values = heads[:n, 1:1 + A], heads[:n, 0]
        rows = index.long() if index is not None else torch.arange(offset, offset + n, device=self.device)
        kl_old = dist.kl_divergence(get_action_distribution(space, buff.action_logits[rows]))[buff.valids[rows]]
        summaries = AttrDict(adv_mean=sc[6], adv_std=sc[7], num_valid=sc[8], entropy=sc[9], kl_divergence_max=sc[5],
                             kl_old_mean=sc[4].clone(), ratio=self._ratio[:n], values=heads[:n, 0], g_heads=g_heads)
        return dist, sc[0].clone(), sc[1].clone(), kl_old, sc[2].clone(), sc[3].clone(), summaries

    def _losses_native(self, buff: AttrDict, mb, num_invalids: int, scalars_out: Optional[torch.Tensor] = None,
                       moments: Optional[torch.Tensor] = None):
        """learner.py:537-669 for one minibatch mb=(index, offset, n): forward, (v-trace), advantage moments,
        fused loss forward+backward.  Returns (acts, g_heads, scalars[16] device tensor) — scalars follow
        sf_loss_scalars: policy, exploration, kl, value losses, kl mean/max, adv mean/std, n_valid, entropy."""
        cfg, ac = self.cfg, self.actor_critic
        index, offset, n = mb
        A = self.num_action_params
        rnn = None
        if cfg.use_rnn:  # learner.py:557-569: chunk-start states, done-or-invalid boundaries (masked-loop BPTT)
            R, Cn = cfg.recurrence, n // cfg.recurrence
            keep_tm = ac._buf(("rnn", "keep_tm"), (R, Cn))
            h0 = ac._buf(("rnn", "h0"), (Cn, buff.rnn_states.shape[-1]))
            lib.rnn_chunk_setup(buff.dones, buff.valids, buff.rnn_states, index, offset, Cn, R, keep_tm, h0, traj_T=buff.T)
            rnn = dict(R=R, h0=h0, keep_tm=keep_tm)
        acts = ac.forward_heads(buff.obs, n, sample_stride=ac.obs_elems, index=index, offset=offset,
                                traj_T=buff.T, tag="train", rnn=rnn)
        heads = acts[-1]
        ld = ac.heads_ld  # 1 + A padded to a multiple of 4; padding columns of g_heads stay zero
        params, values = heads[:, 1:], heads[:, 0]
        g_heads = ac._zbuf(("g", "heads"), (n, ld))
        if cfg.with_vtrace:
            vs = ac._buf(("vt", "vs"), (n,))
            adv = ac._buf(("vt", "adv"), (n,))
            lib.vtrace(params, ld, values, ld, buff.actions, buff.log_prob_actions, buff.rewards, buff.dones, index,
                       offset, n, A, self.loss_cfg.action_kind, cfg.recurrence, cfg.gamma, cfg.vtrace_rho,
                       cfg.vtrace_c, vs, adv, head_sizes=self._head_sizes)
            lib.moments(adv, buff.valids, index, n, self._moments, offset=offset, dense_x=True)
            adv_arr, tgt_arr = adv, vs
        elif moments is None:
            lib.moments(buff.advantages, buff.valids, index, n, self._moments, offset=offset)
            adv_arr, tgt_arr = buff.advantages, buff.returns
        else:
            adv_arr, tgt_arr = buff.advantages, buff.returns
        if moments is None:
            moments = self._moments
            self._all_reduce(moments)  # global per-minibatch advantage statistics under DP
        # else: `moments` is this minibatch's row of the per-epoch table (_epoch_moments): GAE advantages do not depend on
        # the weights, so every minibatch's {sum, sumsq, n} was computed and exchanged ONCE, before the epoch's first
        # forward pass -- no collective sits between this forward pass and its loss
        self._ratio = ac._buf(("loss", "ratio"), (n,))
        self.loss_cfg.old_values_T = buff.T  # buff["values"] is the slab's [E, T+1] array
        lib.ppo_loss(params, ld, values, ld, buff.actions, buff.log_prob_actions, buff.action_logits, buff["values"],
                     adv_arr, tgt_arr, buff.valids, index, offset, n, A, self.loss_cfg, moments, self._sums,
                     g_heads[:, 1:], g_heads[:, 0], ratio_out=self._ratio)
        self._last_mb = (index, offset, n, values, adv_arr)  # for _record_summaries
        if self.dp and self._dp_reduce_each_mb:  # only the per-minibatch KL-adaptive LR needs global values NOW
            self.group.loss_sums(self._sums)
        out = scalars_out if scalars_out is not None else torch.zeros(16, dtype=torch.float32, device=self.device)
        lib.loss_scalars(self._sums, moments, self.loss_cfg, out)
        return acts, g_heads, out

    def _epoch_moments(self, buff: AttrDict, minibatches) -> Optional[torch.Tensor]:
        """Data-parallel replicas, GAE advantages (no V-trace): {sum(adv), sum(adv^2), n_valid} of EVERY minibatch of the
        epoch in one [num_minibatches, 3] f64 table and ONE all-reduce of it, issued before the epoch's first forward
        pass (the advantages are inputs of the epoch — `_prepare_batch` wrote them — and the index sets are known as
        soon as `_get_minibatches` returns).  Replaces one 24-byte collective per SGD step that sat on the critical
        path between the forward pass and the loss (learner.py:646-647 needs the GLOBAL minibatch statistics).  Same
        kernel, same per-minibatch sums, same reduction over the ranks: the values are those of the per-step exchange.
        V-trace advantages depend on the current weights (learner.py:571-640) and keep the per-step exchange."""
        if not self.dp or self.cfg.with_vtrace or not getattr(self.cfg, "dp_epoch_moments", True):
            return None
        tab = self.actor_critic._buf(("loss", "epoch_moments"), (len(minibatches), 3), dtype=torch.float64)
        for i, (index, offset, n) in enumerate(minibatches):
            lib.moments(buff.advantages, buff.valids, index, n, tab[i], offset=offset)
        self._all_reduce(tab)
        return tab

    # ------------------------------------------------------------------------------------------ SGD
    def _train(self, buff: AttrDict, batch_size: int, experience_size: int, num_invalids: int) -> Optional[AttrDict]:
        """learner.py:671-841"""
        cfg, ac = self.cfg, self.actor_critic
        early_stopping_tolerance = 1e-6
        prev_epoch_actor_loss = 1e9
        recent_kls: List[float] = []
        num_sgd_steps = 0
        n_mb = cfg.num_batches_per_epoch
        self._scalars = ac._buf(("loss", "scalars"), (cfg.num_epochs * n_mb, 16))  # every row used is written in full
        global_size = experience_size * self.world
        need_kl_each_mb = self.lr_scheduler.invoke_after_each_minibatch() and isinstance(self.lr_scheduler, KlAdaptiveScheduler)
        self._dp_reduce_each_mb = need_kl_each_mb
        # per-minibatch KL-adaptive schedule (learner.py:46-85) WITHOUT a read-back per SGD step: the learning rate lives
        # in device memory, sf_lr_kl_adaptive updates it from the step's mean KL (scalars row slot 4; slot 10 = new rate)
        # and sf_adam_step_dlr reads it.  Lamb keeps the host form (its trust ratios need the rate on the host anyway).
        lr_on_device = need_kl_each_mb and cfg.optimizer == "adam"
        if lr_on_device:
            if not hasattr(self, "_lr_dev"):
                self._lr_dev = torch.zeros(1, dtype=torch.float32, device=self.device)
            self._lr_dev.fill_(float(self.curr_lr))
        self._grad_norms = []
        # fused recurrent passes: sticky abort word, cleared once per call; the optimiser kernels skip their update
        # while it is set, and it is read back with every epoch's scalars (before any further epoch is trained)
        skip = ac.rnn_abort_word() if hasattr(ac, "rnn_abort_word") else None
        if skip is not None:
            ac.rnn_abort_clear()
        for epoch in range(cfg.num_epochs):
            minibatches = self._get_minibatches(batch_size, experience_size)
            mom_tab = self._epoch_moments(buff, minibatches)
            for batch_num, mb in enumerate(minibatches):
                row = self._scalars[epoch * n_mb + batch_num]
                acts, g_heads, _ = self._losses_native(buff, mb, num_invalids, row,
                                                       moments=None if mom_tab is None else mom_tab[batch_num])
                index, offset, n = mb
                # C1: every replica's gradient already carries the GLOBAL 1/n_valid -> SUM over replicas
                if self._dp_split is not None:
                    cut = ac._segs[self._dp_split][0]
                    pending = []
                    ac.backward(acts, g_heads, buff.obs, n, sample_stride=ac.obs_elems, index=index, offset=offset,
                                traj_T=buff.T,
                                on_layer_done=lambda li: pending.append(self.group.all_reduce_grads_async(
                                    ac.flat_grads[cut:])) if li == self._dp_split else None)
                    self.group.all_reduce_grads(ac.flat_grads[:cut])
                    for work in pending:
                        work.wait()
                else:
                    ac.backward(acts, g_heads, buff.obs, n, sample_stride=ac.obs_elems, index=index, offset=offset,
                                traj_T=buff.T)
                    if self.dp:
                        self.group.all_reduce_grads(ac.flat_grads)
                actual_lr = self.curr_lr
                valid_frac = 1.0
                if self._global_invalids > 0:  # learner.py:788-794
                    valid_frac = (global_size - self._global_invalids) / global_size
                    actual_lr = self.curr_lr * valid_frac
                self.adam_step_count += 1
                use_clip = cfg.max_grad_norm > 0.0
                if use_clip or getattr(cfg, "record_grad_norm", False):
                    lib.grad_sumsq(ac.flat_grads, self._sumsq)
                    if getattr(cfg, "record_grad_norm", False):
                        self._grad_norms.append(float(self._sumsq.sqrt().item()))
                if cfg.optimizer == "lamb":  # optimizers.py:14-189 (its own defaults: weight_decay 1e-4, min_trust 0.01)
                    if self._lamb is None:
                        seg, nseg = ac.tensor_segment_ids()
                        self._lamb = (seg, nseg, torch.empty_like(ac.flat_params),
                                      torch.zeros(128, dtype=torch.float64, device=self.device))
                    seg, nseg, scratch, seg_sums = self._lamb
                    lib.lamb_step(ac.flat_params, ac.flat_grads, self.exp_avg, self.exp_avg_sq, scratch, seg, seg_sums,
                                  nseg, self.adam_step_count, actual_lr, cfg.adam_beta1, cfg.adam_beta2, cfg.adam_eps,
                                  1e-4, 0.01, cfg.max_grad_norm if use_clip else 0.0, self._sumsq if use_clip else None,
                                  skip_flag=skip)
                elif lr_on_device:
                    lib.adam_step_dlr(ac.flat_params, ac.flat_grads, self.exp_avg, self.exp_avg_sq, self.adam_step_count,
                                      self._lr_dev, valid_frac, cfg.adam_beta1, cfg.adam_beta2, cfg.adam_eps,
                                      cfg.max_grad_norm if use_clip else 0.0, self._sumsq if use_clip else None,
                                      skip_flag=skip)
                else:
                    lib.adam_step(ac.flat_params, ac.flat_grads, self.exp_avg, self.exp_avg_sq, self.adam_step_count,
                                  actual_lr, cfg.adam_beta1, cfg.adam_beta2, cfg.adam_eps,
                                  cfg.max_grad_norm if use_clip else 0.0, self._sumsq if use_clip else None,
                                  skip_flag=skip)
                ac.params_changed()
                num_sgd_steps += 1
                self.train_step += 1
                if lr_on_device:
                    sch = self.lr_scheduler
                    lib.lr_kl_adaptive(row[4:5], self._lr_dev, sch.threshold, sch.min_lr, sch.max_lr, lr_out=row[10:11])
                elif need_kl_each_mb:
                    recent_kls.append(float(row[4].item()))
                    self.curr_lr = self.lr_scheduler.update(self.curr_lr, recent_kls)
                elif self.lr_scheduler.invoke_after_each_minibatch():
                    self.curr_lr = self.lr_scheduler.update(self.curr_lr, recent_kls)
                # K20: same-process inference reads the same flat buffer in stream order; publishing = version bump
                self.policy_versions_tensor[self.policy_id] = self.train_step
            # ---- end of epoch: ONE readback (actor losses for early stopping, KLs for the per-epoch scheduler)
            blk = self._scalars[epoch * n_mb:(epoch + 1) * n_mb]
            if self.dp and not need_kl_each_mb:
                # data-parallel: every replica divided its LOCAL sums by the GLOBAL n, so the loss / KL / entropy means
                # add up across ranks (max KL: MAX) — ONE packed exchange per epoch (ReplicaGroup.reduce_sum_max)
                # instead of two collectives per SGD step; the per-rank columns (adv mean / std, n) stay as they are
                # the abort word of the fused recurrent passes is rank-local: it rides along as a MAX column so that
                # every replica raises in the same epoch (one rank raising alone would leave the others in a collective)
                flag = (skip.to(torch.float32) if skip is not None else blk.new_zeros(1)).expand(blk.shape[0], 1)
                pack = torch.cat([blk[:, [0, 1, 2, 3, 4, 9, 5]], flag], dim=1).contiguous()
                self.group.reduce_sum_max(pack, (6, 7))
                blk[:, [0, 1, 2, 3, 4, 9, 5]] = pack[:, :7]
                aborted = skip is not None and bool(pack[0, 7].item() != 0)
            elif self.dp and skip is not None:
                aborted = bool(self.group.all_reduce_max(skip.to(torch.float32)).item() != 0)
            else:
                aborted = skip is not None and ac.rnn_pass_aborted()
            rows = blk.cpu()
            if aborted:
                raise lib.SfHipError(
                    "a fused recurrent sequence pass was aborted (a work-group never arrived: is the GPU shared with "
                    "another process?); the optimiser steps after it were skipped, the weights are those of the last "
                    "good SGD step; set SF_LSTM_SEQ=0 to use the per-step kernels")
            actor_losses = (rows[:, 0] + rows[:, 1] + rows[:, 2]).double().numpy()
            if lr_on_device:  # the rate the device schedule arrived at, read back with the epoch's scalars
                self.curr_lr = float(rows[-1, 10])
                actual_lr = self.curr_lr * valid_frac
            if not need_kl_each_mb or lr_on_device:
                recent_kls.extend(rows[:, 4].double().tolist())
            if self.lr_scheduler.invoke_after_each_epoch():
                self.curr_lr = self.lr_scheduler.update(self.curr_lr, recent_kls)
            new_epoch_actor_loss = float(np.mean(actor_losses))
            if abs(prev_epoch_actor_loss - new_epoch_actor_loss) < early_stopping_tolerance:
                break
            prev_epoch_actor_loss = new_epoch_actor_loss
        last = rows[-1]
        stats = AttrDict(lr=self.curr_lr, actual_lr=actual_lr, env_steps=self.env_steps,
                         policy_loss=float(last[0]), exploration_loss=float(last[1]),
                         kl_loss=float(last[2]), value_loss=float(last[3]), kl_divergence=float(last[4]),
                         kl_divergence_max=float(last[5]), adv_mean=float(last[6]), adv_std=float(last[7]),
                         entropy=float(last[9]), loss=float(last[0] + last[1] + last[2] + last[3]),
                         num_sgd_steps=num_sgd_steps)
        if self._should_save_summaries():  # learner.py:312-317: every 2 s early on, decaying to every 2 min
            stats.update(self._record_summaries(buff))
            self._last_summary_time = time.time()
        self.last_summary = stats
        return stats

    _last_summary_time = 0.0
    _summary_rate_decay = LinearDecay([(0, 2.0), (100000, 60.0), (1000000, 120.0)])  # seconds between summaries over train_step (learner.py:164)

    def _should_save_summaries(self) -> bool:
        every = self._summary_rate_decay.at(float(self.train_step))
        return time.time() - self._last_summary_time >= every or getattr(self.cfg, "summaries_every_train", False)

    def _record_summaries(self, buff: AttrDict) -> Dict[str, float]:
        """The rest of learner.py:843-923 for the LAST minibatch of the call: ratio / clipping / value-delta statistics,
        action / advantage / logit ranges, policy-lag (version_diff_*), gradient norm and Adam's largest second moment —
        ONE pass over the minibatch on the device (sf_train_summaries) and ONE readback of 24 doubles (the reference
        does ~25 torch reductions with an `.item()` each)."""
        index, offset, n, values, adv_arr = self._last_mb
        if not hasattr(self, "_summ"):
            self._summ = torch.empty(24, dtype=torch.float64, device=self.device)
        ld = values.stride(0) if values.dim() == 1 else 1
        lib.train_summaries(buff.valids, self._ratio, values, ld, buff["values"], buff.T, buff.actions, self.num_actions,
                            adv_arr, bool(self.cfg.with_vtrace), buff.policy_id, buff.policy_version, buff.action_logits,
                            self.num_action_params, index, offset, n, self.policy_id, self.train_step,
                            self.loss_cfg.clip_ratio, self.exp_avg_sq, self._summ)
        o = self._summ.cpu().tolist()
        rows, nv, ns = o[0], o[1], o[2]
        has_v, has_s = nv > 0, ns > 0
        return dict(
            valids_fraction=nv / rows, same_policy_fraction=ns / rows, value=o[3] / rows,
            ratio_mean=o[4] / nv if has_v else 0.0, ratio_min=o[8] if has_v else 1.0, ratio_max=o[9] if has_v else 1.0,
            fraction_clipped=o[5] / nv if has_v else 0.0, value_delta=o[6] / rows, value_delta_max=o[10],
            act_min=o[11], act_max=o[12], adv_min=o[13], adv_max=o[14], max_abs_logprob=o[15],
            version_diff_avg=o[7] / ns if has_s else 0.0, version_diff_min=o[16] if has_s else 0.0,
            version_diff_max=o[17] if has_s else 0.0, grad_norm=float(self._sumsq.sqrt().item()),
            adam_max_second_moment=o[18])

    def train(self, batch: TensorDict) -> Optional[Dict]:
        """learner.py:1036-1067"""
        self._maybe_update_cfg()
        self._maybe_load_policy()
        buff, experience_size, num_invalids = self._prepare_batch(batch)
        if self._global_invalids >= experience_size * self.world:
            return None
        train_stats = self._train(buff, self.cfg.batch_size, experience_size, num_invalids)
        frameskip = self.env_info.frameskip if self.cfg.summaries_use_frameskip else 1
        self.env_steps += experience_size * self.world * frameskip
        stats = {LEARNER_ENV_STEPS: self.env_steps, POLICY_ID_KEY: self.policy_id}
        if train_stats is not None:
            stats[TRAIN_STATS] = train_stats
        return stats

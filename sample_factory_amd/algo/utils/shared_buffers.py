"""Device-resident trajectory slab — the layout of sample_factory/algo/utils/shared_buffers.py:79-117.

Env-major [num_traj, rollout(+1), ...]; every policy output is f32 (shared_buffers.py:100-103); sentinel fills as in
the reference (:45-49,107-115) so reads of unwritten slots are obvious.  One slab lives in HBM for the whole run; the
rollout kernels write into it, the learner kernels read it in place (no batcher copy, batcher.py:192-212).
"""
from __future__ import annotations

import numpy as np
import torch

from sample_factory_amd.algo.utils.tensor_dict import TensorDict
from sample_factory_amd.envs.spaces import calc_num_action_parameters, calc_num_actions

MAGIC_FLOAT = -4242.42
MAGIC_INT = 43

_NP2T = {np.dtype(np.uint8): torch.uint8, np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float32,
         np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64, np.dtype(np.bool_): torch.bool}


def policy_output_shapes(num_actions, num_action_distribution_parameters):
    """shared_buffers.py:67-76"""
    return [("actions", [num_actions]), ("action_logits", [num_action_distribution_parameters]),
            ("log_prob_actions", []), ("values", []), ("policy_version", [])]


def _init(shape, dtype, device):
    t = torch.zeros(shape, dtype=dtype, device=device)
    if t.is_floating_point():
        t.fill_(MAGIC_FLOAT)
    elif dtype in (torch.int32, torch.int64, torch.uint8):
        t.fill_(MAGIC_INT)
    return t


def alloc_trajectory_tensors(env_info, num_traj, rollout, rnn_size, device, share=False) -> TensorDict:
    obs_space = env_info.obs_space
    if not hasattr(obs_space, "spaces"):
        raise Exception("Only Dict observations spaces are supported")
    t = TensorDict()
    t["obs"] = TensorDict()
    for name, space in obs_space.spaces.items():
        t["obs"][name] = _init([num_traj, rollout + 1] + list(space.shape), _NP2T[np.dtype(space.dtype)], device)
    t["rnn_states"] = _init([num_traj, rollout + 1, rnn_size], torch.float32, device)
    na, nap = calc_num_actions(env_info.action_space), calc_num_action_parameters(env_info.action_space)
    for name, shape in policy_output_shapes(na, nap):
        rl = rollout + 1 if name == "values" else rollout
        t[name] = _init([num_traj, rl] + shape, torch.float32, device)
    t["rewards"] = torch.full([num_traj, rollout], -42.42, dtype=torch.float32, device=device)
    t["dones"] = torch.ones([num_traj, rollout], dtype=torch.bool, device=device)
    t["time_outs"] = torch.zeros([num_traj, rollout], dtype=torch.bool, device=device)
    t["policy_id"] = torch.full([num_traj, rollout], -1, dtype=torch.int32, device=device)
    t["valids"] = torch.zeros([num_traj, rollout + 1], dtype=torch.bool, device=device)
    return t


def trajectories_per_minibatch(cfg) -> int:
    return cfg.batch_size // cfg.rollout


def trajectories_per_training_iteration(cfg) -> int:
    return cfg.num_batches_per_epoch * trajectories_per_minibatch(cfg)


class BufferMgr:
    """Slab bookkeeping of shared_buffers.py:152-239 for ONE process per GPU: how many trajectory rows the device slab
    needs (agents x envs per worker, x2 when rollouts overlap training or several policies share the sampler, never
    fewer than the learner needs to accumulate `max_batches_to_accumulate` datasets), the queue of free row slices of
    `sampling_trajectories_per_iteration` rows handed to the sampler, and the policy-version tensor.  The slab itself
    is the device-resident TensorDict of alloc_trajectory_tensors; the learner trains on row slices IN PLACE (no
    batcher copy), so a slice goes sampler -> learner -> back to this queue."""

    def __init__(self, cfg, env_info, device, allocate: bool = True):
        import math
        from collections import deque
        self.cfg, self.env_info, self.device = cfg, env_info, device
        num_buffers = sum(env_info.num_agents * cfg.num_envs_per_worker for _ in range(cfg.num_workers))
        self.trajectories_per_training_iteration = trajectories_per_training_iteration(cfg)
        if cfg.batched_sampling:
            per_iter = (env_info.num_agents * cfg.num_envs_per_worker) // cfg.worker_num_splits
            assert math.gcd(self.trajectories_per_training_iteration, per_iter) == min(
                self.trajectories_per_training_iteration, per_iter), \
                f"worker_traj_per_iteration={per_iter} should divide {self.trajectories_per_training_iteration} or vice versa"
            self.sampling_trajectories_per_iteration = per_iter
        else:
            self.sampling_trajectories_per_iteration = -1
        if cfg.async_rl or cfg.num_policies > 1:
            num_buffers *= 2  # one set of buffers to sample into, one to learn from
        self.max_batches_to_accumulate = cfg.num_batches_to_accumulate if cfg.async_rl else 1
        self.buffers_per_device = {str(device): num_buffers}
        # at the very least enough rows to feed the learner (shared_buffers.py:206-211)
        num_buffers = max(num_buffers, self.max_batches_to_accumulate * self.trajectories_per_training_iteration *
                          cfg.num_policies)
        self.num_buffers = num_buffers
        self.traj_buffer_queue = deque()
        if cfg.batched_sampling:
            for i in range(0, num_buffers, self.sampling_trajectories_per_iteration):
                self.traj_buffer_queue.append(slice(i, i + self.sampling_trajectories_per_iteration))
        else:
            for i in range(num_buffers):
                self.traj_buffer_queue.append(i)
        self.policy_versions = torch.zeros([cfg.num_policies], dtype=torch.int32)
        self.traj_tensors = None
        if allocate:
            from sample_factory_amd.model.actor_critic import get_rnn_size
            self.traj_tensors = alloc_trajectory_tensors(env_info, num_buffers, cfg.rollout, get_rnn_size(cfg), device)

    def get_free_slice(self):
        """next free row slice for the sampler, or None when every slab row is in flight (sampler must pause:
        inference_worker.py:175-181)"""
        return self.traj_buffer_queue.popleft() if self.traj_buffer_queue else None

    def release(self, s) -> None:
        self.traj_buffer_queue.append(s)

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

"""Device visibility per process — sample_factory/utils/gpu_utils.py:12-89 for ROCm.

The reference narrows `CUDA_VISIBLE_DEVICES` per learner / inference / rollout process.  On MI355X the variable the HIP
runtime reads is `HIP_VISIBLE_DEVICES` (`CUDA_VISIBLE_DEVICES` is honoured by PyTorch-ROCm as an alias; when a user set
it and not the HIP one, it seeds the HIP one so both agree).  This engine runs one process per GPU (torchrun sets
LOCAL_RANK), so the per-policy assignment collapses to "process index -> device index", same arithmetic as the reference.
`CUDA_ENVVAR` keeps its name because user code imports it.
"""
from __future__ import annotations

import os
from typing import List, Optional

from sample_factory_amd.utils.get_available_gpus import get_gpus_without_triggering_pytorch_cuda_initialization
from sample_factory_amd.utils.utils import log

CUDA_ENVVAR = "HIP_VISIBLE_DEVICES"
_ALIAS = "CUDA_VISIBLE_DEVICES"


def set_global_cuda_envvars(cfg) -> None:
    """once per experiment, before any HIP call: make the set of usable devices explicit (none for --device=cpu)"""
    if CUDA_ENVVAR not in os.environ:
        if getattr(cfg, "device", "gpu") == "cpu":
            os.environ[CUDA_ENVVAR] = ""
        elif _ALIAS in os.environ:
            os.environ[CUDA_ENVVAR] = os.environ[_ALIAS]
        else:
            os.environ[CUDA_ENVVAR] = get_gpus_without_triggering_pytorch_cuda_initialization(os.environ)
    log.info("Environment var %s is %s", CUDA_ENVVAR, os.environ[CUDA_ENVVAR])


def get_available_gpus() -> List[int]:
    """device indices listed in HIP_VISIBLE_DEVICES"""
    return [int(g) for g in os.environ.get(CUDA_ENVVAR, "").split(",") if g.strip()]


def gpus_for_process(process_idx: int, num_gpus_per_process: int, gpu_mask: Optional[List[int]] = None) -> List[int]:
    """indices (relative to the visible set) process `process_idx` should use: consecutive blocks, wrapping around"""
    available = get_available_gpus()
    if gpu_mask is not None:
        assert len(available) >= len(gpu_mask), \
            f"Number of available GPUs ({len(available)}) is less than number of GPUs in mask ({len(gpu_mask)})"
        available = [available[g] for g in gpu_mask]
    if not available:
        return []
    first = process_idx * num_gpus_per_process
    return [(first + i) % len(available) for i in range(num_gpus_per_process)]


def set_gpus_for_process(process_idx, num_gpus_per_process, process_type, gpu_mask=None) -> List[int]:
    """narrow HIP_VISIBLE_DEVICES of THIS process to its share (call before the first HIP call of the process)"""
    use = gpus_for_process(process_idx, num_gpus_per_process, gpu_mask)
    if not use:
        os.environ[CUDA_ENVVAR] = ""
        log.debug("Not using GPUs for %s process %d", process_type, process_idx)
        return use
    available = get_available_gpus()
    if gpu_mask is not None:
        available = [available[g] for g in gpu_mask]
    os.environ[CUDA_ENVVAR] = ",".join(str(available[g]) for g in use)
    log.info("Set environment var %s to %r (GPU indices %r) for %s process %d", CUDA_ENVVAR, os.environ[CUDA_ENVVAR], use,
             process_type, process_idx)
    return use


def cuda_envvars_for_policy(policy_id, process_type):
    set_gpus_for_process(policy_id, 1, process_type)

"""Torch helpers of the plugin surface under the reference's module path (sample_factory/algo/utils/torch_utils.py:12-68):
what user model code imports (`calc_num_elements` to size the layer after a conv stack, `to_scalar`, `masked_select`)
plus the runtime switches the reference's workers call.  Own implementation."""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch


def init_torch_runtime(cfg, max_num_threads: Optional[int] = 1) -> None:
    """per-process torch settings.  `cudnn.benchmark` is MIOpen's find mode on ROCm: only user torch modules go through
    MIOpen here (the native network kernels do not), and their shapes are fixed, so the search pays for itself."""
    if max_num_threads is not None:
        torch.set_num_threads(max_num_threads)
    if getattr(cfg, "device", "gpu") == "gpu":
        torch.backends.cudnn.benchmark = True


def inference_context(is_serial: bool):
    """serial mode shares tensors between sampler and learner, so inference tensors are not allowed there"""
    return torch.no_grad() if is_serial else torch.inference_mode()


def to_torch_dtype(numpy_dtype) -> torch.dtype:
    return torch.from_numpy(np.zeros(1, dtype=numpy_dtype)).dtype


def calc_num_elements(module, module_input_shape) -> int:
    """number of output elements of `module` for ONE sample of shape `module_input_shape` (a probe forward)"""
    with torch.no_grad():
        return int(module(torch.rand((1,) + tuple(module_input_shape))).numel())


def to_scalar(value):
    return value.item() if isinstance(value, torch.Tensor) else value


def masked_select(x: torch.Tensor, mask: torch.Tensor, num_non_mask: int) -> torch.Tensor:
    """x[mask] as a flat tensor, skipping the gather when nothing is masked out"""
    return x if num_non_mask == 0 else torch.masked_select(x, mask)


def synchronize(cfg, device) -> None:
    if getattr(cfg, "serial_mode", False):
        return
    device = torch.device(device) if isinstance(device, str) else device
    if device.type == "cuda":
        torch.cuda.synchronize(device)

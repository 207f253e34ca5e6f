"""ObservationNormalizer with running statistics on the GPU — sample_factory/utils/normalize.py:24-70 +
algo/utils/running_mean_std.py:22-136 (RunningMeanStdDictInPlace over the single "obs" key, full-shape statistics).

Only instantiated when cfg.normalize_input=True.  (With normalize_input=False — the north-star preset — the u8 frames
are consumed in place by the first conv layer and nothing here runs.)  Unlike the fused path this one materialises the
normalised f32 minibatch, as the reference does (learner.py:925-941); it is a correctness feature, not the fast path.
"""
from __future__ import annotations

import numpy as np
import torch

from sample_factory_amd import lib


class ObservationNormalizer:
    def __init__(self, cfg, obs_shape, obs_u8: bool, device, all_reduce=None, world: int = 1):
        self.obs_shape = tuple(obs_shape)
        self.D = int(np.prod(obs_shape))
        self.obs_u8 = obs_u8
        self.is_image = len(self.obs_shape) == 3
        self.C = self.obs_shape[0] if self.is_image else 0
        self.HW = self.obs_shape[1] * self.obs_shape[2] if self.is_image else 0
        self.sub_mean = float(cfg.obs_subtract_mean) if abs(cfg.obs_subtract_mean) > 1e-5 else 0.0
        self.inv_scale = float(np.float32(1.0 / cfg.obs_scale)) if abs(cfg.obs_scale - 1.0) > 1e-5 else 1.0
        dev = torch.device(device)
        self.device = dev
        self.mean = torch.zeros(self.D, dtype=torch.float64, device=dev)
        self.var = torch.ones(self.D, dtype=torch.float64, device=dev)
        self.count = torch.ones(1, dtype=torch.float64, device=dev)
        self._count2 = torch.ones(1, dtype=torch.float64, device=dev)
        self._sum = torch.zeros(self.D, dtype=torch.float64, device=dev)
        self._sumsq = torch.zeros(self.D, dtype=torch.float64, device=dev)
        self.mu_tab = torch.zeros(self.D, dtype=torch.float32, device=dev)
        self.rstd_tab = torch.ones(self.D, dtype=torch.float32, device=dev)
        self._all_reduce, self.world = all_reduce, world
        self.refresh_tables()

    def refresh_tables(self) -> None:
        lib.obsnorm_update(self.mean, self.var, self.count, self._count2, None, None, 0, self.D, self.mu_tab,
                           self.rstd_tab)

    def update(self, obs: torch.Tensor, stride: int, n: int, index=None, offset: int = 0, traj_T: int = 0) -> None:
        """training-mode statistics update over n observations (running_mean_std.py:64-77)"""
        lib.obsnorm_moments(obs, self.obs_u8, stride, index, offset, traj_T, n, self.D, self.sub_mean, self.inv_scale,
                            self._sum, self._sumsq)
        if self._all_reduce is not None:  # data-parallel replicas: global batch moments
            self._all_reduce(self._sum)
            self._all_reduce(self._sumsq)
        lib.obsnorm_update(self.mean, self.var, self.count, self._count2, self._sum, self._sumsq, n * self.world,
                           self.D, self.mu_tab, self.rstd_tab)
        self.count, self._count2 = self._count2, self.count

    def apply(self, obs: torch.Tensor, stride: int, n: int, out: torch.Tensor, index=None, offset: int = 0,
              traj_T: int = 0, tabs=None) -> None:
        mu, rstd = tabs if tabs is not None else (self.mu_tab, self.rstd_tab)  # tabs: published snapshot (async mode)
        lib.obsnorm_apply(obs, self.obs_u8, stride, index, offset, traj_T, n, self.D, self.C, self.HW, self.sub_mean,
                          self.inv_scale, mu, rstd, out)

    def state_dict(self, prefix="obs_normalizer.running_mean_std.running_mean_std.obs."):
        return {prefix + "running_mean": self.mean.detach().cpu().view(self.obs_shape).clone(),
                prefix + "running_var": self.var.detach().cpu().view(self.obs_shape).clone(),
                prefix + "count": self.count.detach().cpu().clone()}

    def load_state_dict(self, sd, prefix="obs_normalizer.running_mean_std.running_mean_std.obs."):
        self.mean.copy_(torch.as_tensor(sd[prefix + "running_mean"], dtype=torch.float64).reshape(-1))
        self.var.copy_(torch.as_tensor(sd[prefix + "running_var"], dtype=torch.float64).reshape(-1))
        self.count.copy_(torch.as_tensor(sd[prefix + "count"], dtype=torch.float64).reshape(-1))
        self.refresh_tables()

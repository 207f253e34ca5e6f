"""RunningMeanStdInPlace (scalar statistics) on the GPU — sample_factory/algo/utils/running_mean_std.py:22-110.

State is one device tensor of 3 doubles {running_mean, running_var, count} (count starts at 1, as in the reference);
`state_dict()` exposes it under the reference's buffer names.
"""
from __future__ import annotations

import torch

from sample_factory_amd import lib

_NORM_EPS = 1e-5
_DEFAULT_CLIP = 5.0


class RunningMeanStdInPlace:
    def __init__(self, input_shape, device, all_reduce=None):
        assert tuple(input_shape) == (1,), "only scalar statistics (returns normaliser) are implemented natively"
        self.input_shape = tuple(input_shape)
        self.device = torch.device(device)
        self.stats = torch.tensor([0.0, 1.0, 1.0], dtype=torch.float64, device=self.device)
        self._stats_new = torch.empty_like(self.stats)
        self._moments = torch.zeros(3, dtype=torch.float64, device=self.device)
        self.training = True
        self._all_reduce = all_reduce  # data-parallel learner: sums the batch moments over ranks

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def __call__(self, x: torch.Tensor, denormalize: bool = False) -> None:
        """Normalizes (or de-normalizes) IN PLACE, updating the statistics first when training."""
        if self.training and not denormalize:
            lib.moments(x, None, None, x.numel(), self._moments)
            if self._all_reduce is not None:
                self._all_reduce(self._moments)
            lib.rms_update(self.stats, self._moments, self._stats_new)
            self.stats, self._stats_new = self._stats_new, self.stats
        lib.rms_apply(x, self.stats, denormalize)

    def state_dict(self, prefix=""):
        s = self.stats.detach().cpu()
        return {prefix + "running_mean": s[0:1].clone(), prefix + "running_var": s[1:2].clone(),
                prefix + "count": s[2:3].clone()}

    def load_state_dict(self, sd, prefix=""):
        vals = [float(sd[prefix + k].reshape(-1)[0]) for k in ("running_mean", "running_var", "count")]
        self.stats.copy_(torch.tensor(vals, dtype=torch.float64))

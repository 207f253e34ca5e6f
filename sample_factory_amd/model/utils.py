"""Layer initialisers a custom model applies with `module.apply(...)` (sample_factory/model/utils.py:4-17;
sf_examples/nethack/models/scaled.py).  Both touch Linear / Conv2d only, zero the bias and return the module."""
from __future__ import annotations

import torch.nn as nn


def _init(module, weight_init):
    if isinstance(module, (nn.Linear, nn.Conv2d)):
        weight_init(module.weight)
        if module.bias is not None:
            module.bias.data.zero_()
    return module


def orthogonal_init(module, gain=1.0):
    return _init(module, lambda w: nn.init.orthogonal_(w, gain=gain))


def he_normal_init(module):
    return _init(module, lambda w: nn.init.kaiming_normal_(w, mode="fan_in", nonlinearity="relu"))

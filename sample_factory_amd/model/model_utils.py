"""Model-building helpers of the plugin surface — sample_factory/model/model_utils.py:11-72 under its reference path:
`nonlinearity(cfg)`, `fc_layer`, `create_mlp`, the `ModelModule` base of Encoder / ModelCore / Decoder, `model_device`,
`get_rnn_size`.  User encoders are written against these (sf_examples/train_custom_env_custom_model.py:99-117)."""
from __future__ import annotations

from typing import List, Optional

import torch
from torch import nn

from sample_factory_amd.cfg.configurable import Configurable
from sample_factory_amd.model.actor_critic import ACT_KIND, get_rnn_size  # noqa: F401


def nonlinearity(cfg, inplace: bool = False) -> nn.Module:
    """the activation cfg.nonlinearity names (elu | relu | tanh)"""
    kind = cfg.nonlinearity
    if kind == "elu":
        return nn.ELU(inplace=inplace)
    if kind == "relu":
        return nn.ReLU(inplace=inplace)
    if kind == "tanh":
        return nn.Tanh()
    raise Exception(f"Unknown {cfg.nonlinearity=}")


def fc_layer(in_features: int, out_features: int, bias=True, spec_norm=False) -> nn.Module:
    layer = nn.Linear(in_features, out_features, bias)
    return nn.utils.spectral_norm(layer) if spec_norm else layer


def create_mlp(layer_sizes: List[int], input_size: int, activation: nn.Module) -> nn.Module:
    """Linear + activation per entry of layer_sizes (Sequential indices 0, 2, 4, ... hold the Linear layers); Identity
    for an empty list"""
    layers: List[nn.Module] = []
    for size in layer_sizes:
        layers += [fc_layer(input_size, int(size)), activation]
        input_size = int(size)
    return nn.Sequential(*layers) if layers else nn.Identity()


class ModelModule(nn.Module, Configurable):
    def __init__(self, cfg):
        nn.Module.__init__(self)
        Configurable.__init__(self, cfg)

    def get_out_size(self):
        raise NotImplementedError()


def model_device(model: nn.Module) -> Optional[torch.device]:
    """device of the first parameter; None for a parameter-free module"""
    for p in model.parameters():
        return p.device
    return None

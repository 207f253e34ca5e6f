"""Model-core plugin surface — sample_factory/model/core.py:10-88 under its reference path.

`ModelCore` is the base class of a user-registered core (`register_model_core_factory(f)`, `f(cfg, core_input_size)`);
`ModelCoreRNN` is the reference's GRU / LSTM core with cfg.rnn_num_layers stacked layers as a torch module (parameter
paths `core.core.weight_ih_l<k>` ...), `ModelCoreIdentity` the feed-forward no-op.  State layout per sample, as the
trajectory buffer keeps it: [layer 0 | layer 1 | ...], a layer's block being h (GRU) or [h | c] (LSTM).  The one-layer
default core does not use this class: it runs on the fused sequence kernels (`csrc/sf_rnn.hip`).
"""
from __future__ import annotations

from abc import ABC

import torch
from torch import nn

from sample_factory_amd.model.model_utils import ModelModule


class ModelCore(ModelModule, ABC):
    def __init__(self, cfg):
        super().__init__(cfg)
        self.core_output_size = -1  # set by the derived class

    def get_out_size(self) -> int:
        return self.core_output_size


class ModelCoreRNN(ModelCore):
    """forward(head_output, rnn_states): head_output [n, F] (one step) or a PackedSequence; rnn_states [n, S]"""

    def __init__(self, cfg, input_size: int):
        super().__init__(cfg)
        if cfg.rnn_type not in ("gru", "lstm"):
            raise RuntimeError(f"Unknown RNN type {cfg.rnn_type}")
        self.is_gru = cfg.rnn_type == "gru"
        self.H, self.rnn_num_layers = int(cfg.rnn_size), int(cfg.rnn_num_layers)
        self.core = (nn.GRU if self.is_gru else nn.LSTM)(int(input_size), self.H, self.rnn_num_layers)
        self.core_output_size = self.H

    def forward(self, head_output, rnn_states):
        one_step = torch.is_tensor(head_output)
        x = head_output.unsqueeze(0) if one_step else head_output
        n = rnn_states.shape[0]
        per_layer = rnn_states.reshape(n, self.rnn_num_layers, -1).transpose(0, 1)      # [L, n, H] or [L, n, 2H]
        if self.is_gru:
            out, new = self.core(x, per_layer.contiguous())
        else:
            h, c = per_layer[..., :self.H], per_layer[..., self.H:]
            out, (h, c) = self.core(x, (h.contiguous(), c.contiguous()))
            new = torch.cat((h, c), dim=2)
        if one_step:
            out = out.squeeze(0)
        return out, new.transpose(0, 1).reshape(n, -1)


class ModelCoreIdentity(ModelCore):
    """no recurrence: features and (fake) states pass through"""

    def __init__(self, cfg, input_size: int):
        super().__init__(cfg)
        self.core_output_size = int(input_size)

    def forward(self, head_output, fake_rnn_states):
        return head_output, fake_rnn_states


def default_make_core_func(cfg, core_input_size: int) -> ModelCore:
    return (ModelCoreRNN if cfg.use_rnn else ModelCoreIdentity)(cfg, core_input_size)

"""Decoder plugin surface — sample_factory/model/decoder.py:11-35 under its reference path: the layers between the core
and the value / action heads.  `Decoder` is the base of a user-registered decoder (`register_decoder_factory(f)`,
`f(cfg, core_output_size)`); `MlpDecoder` is the default (cfg.decoder_mlp_layers, parameters under `.mlp`)."""
from __future__ import annotations

from abc import ABC
from typing import List

from sample_factory_amd.model.model_utils import ModelModule, create_mlp, nonlinearity


class Decoder(ModelModule, ABC):
    pass


class MlpDecoder(Decoder):
    def __init__(self, cfg, decoder_input_size: int):
        super().__init__(cfg)
        self.core_input_size = int(decoder_input_size)
        sizes: List[int] = list(getattr(cfg, "decoder_mlp_layers", []) or [])
        self.mlp = create_mlp(sizes, self.core_input_size, nonlinearity(cfg))
        self.decoder_out_size = int(sizes[-1]) if sizes else self.core_input_size

    def forward(self, core_output):
        return self.mlp(core_output)

    def get_out_size(self) -> int:
        return self.decoder_out_size


def default_make_decoder_func(cfg, core_input_size: int) -> Decoder:
    return MlpDecoder(cfg, core_input_size)

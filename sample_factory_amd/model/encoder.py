"""Encoder plugin surface — sample_factory/model/encoder.py:15-240 under its reference path.

`Encoder` is the base class a user's custom encoder derives from and registers through
`global_model_factory().register_encoder_factory(f)`, `f(cfg, obs_space) -> Encoder`
(sf_examples/train_custom_env_custom_model.py:99-136).  The default encoders below are the reference's architectures
written as torch modules with the reference's parameter paths (`encoders.<key>.mlp_head.<i>.*`,
`encoders.<key>.enc.conv_head.<i>.*`, `encoders.<key>.enc.mlp_layers.<i>.*`), so checkpoints move both ways.  They run
on the torch-module path of this engine (`model/torch_policy.py`): observation dicts with several keys, stacked RNN
layers, separate actor / critic weights and any user-registered part.  The single-key default model does NOT go through
these classes — it runs on the hand-written HIP kernels (`model/actor_critic.py`).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
from torch import Tensor, nn

from sample_factory_amd.algo.utils.torch_utils import calc_num_elements
from sample_factory_amd.model.model_utils import ModelModule, create_mlp, model_device, nonlinearity
from sample_factory_amd.utils.utils import log


class Encoder(ModelModule):
    """observations (dict of normalised tensors) -> feature vector [n, get_out_size()]"""

    def __init__(self, cfg):
        super().__init__(cfg)

    def get_out_size(self) -> int:
        raise NotImplementedError()

    def model_to_device(self, device) -> None:
        """where the parameters live; override to keep parts of the model elsewhere"""
        self.to(device)

    def device_for_input_tensor(self, input_tensor_name: str) -> Optional[torch.device]:
        """device the rollout code should place observation `input_tensor_name` on (None: parameter-free encoder)"""
        return model_device(self)

    def type_for_input_tensor(self, input_tensor_name: str) -> torch.dtype:
        return torch.float32


class MlpEncoder(Encoder):
    """vector observation -> cfg.encoder_mlp_layers"""

    def __init__(self, cfg, obs_space):
        super().__init__(cfg)
        sizes: List[int] = list(cfg.encoder_mlp_layers)
        self.mlp_head = create_mlp(sizes, int(obs_space.shape[0]), nonlinearity(cfg))
        self.encoder_out_size = int(sizes[-1]) if sizes else int(obs_space.shape[0])

    def forward(self, obs: Tensor) -> Tensor:
        return self.mlp_head(obs)

    def get_out_size(self) -> int:
        return self.encoder_out_size


# cfg.encoder_conv_architecture -> [(out_channels, kernel, stride)]
CONV_FILTERS = dict(convnet_simple=((32, 8, 4), (64, 4, 2), (128, 3, 2)),
                    convnet_impala=((16, 8, 4), (32, 4, 2)),
                    convnet_atari=((32, 8, 4), (64, 4, 2), (64, 3, 1)))


class ConvEncoderImpl(nn.Module):
    """conv stack (`conv_head`) + cfg.encoder_conv_mlp_layers (`mlp_layers`)"""

    def __init__(self, obs_shape, conv_filters, extra_mlp_layers: List[int], activation: nn.Module):
        super().__init__()
        layers: List[nn.Module] = []
        for spec in conv_filters:
            if spec == "maxpool_2x2":
                layers.append(nn.MaxPool2d((2, 2)))
                continue
            if not isinstance(spec, (list, tuple)):
                raise NotImplementedError(f"Layer {spec} not supported!")
            cin, cout, k, stride = spec
            layers += [nn.Conv2d(cin, cout, k, stride=stride), activation]
        self.conv_head = nn.Sequential(*layers)
        self.conv_head_out_size = calc_num_elements(self.conv_head, tuple(obs_shape))
        self.mlp_layers = create_mlp(list(extra_mlp_layers), self.conv_head_out_size, activation)
        self.out_size = int(extra_mlp_layers[-1]) if len(extra_mlp_layers) else self.conv_head_out_size

    def forward(self, obs: Tensor) -> Tensor:
        x = self.conv_head(obs)
        return self.mlp_layers(x.contiguous().view(-1, self.conv_head_out_size))


class ConvEncoder(Encoder):
    def __init__(self, cfg, obs_space):
        super().__init__(cfg)
        arch = cfg.encoder_conv_architecture
        if arch not in CONV_FILTERS:
            raise NotImplementedError(f"Unknown encoder architecture {arch}")
        c = int(obs_space.shape[0])
        filters = []
        for cout, k, st in CONV_FILTERS[arch]:
            filters.append((c, cout, k, st))
            c = cout
        self.enc = ConvEncoderImpl(obs_space.shape, filters, list(cfg.encoder_conv_mlp_layers), nonlinearity(cfg))
        self.encoder_out_size = self.enc.out_size
        log.debug("Conv encoder output size: %d", self.encoder_out_size)

    def forward(self, obs: Tensor) -> Tensor:
        return self.enc(obs)

    def get_out_size(self) -> int:
        return self.encoder_out_size


class ResBlock(nn.Module):
    """x + conv3x3(act(conv3x3(act(x)))), 'same' padding"""

    def __init__(self, cfg, input_ch: int, output_ch: int):
        super().__init__()
        self.res_block_core = nn.Sequential(nonlinearity(cfg), nn.Conv2d(input_ch, output_ch, 3, stride=1, padding=1),
                                            nonlinearity(cfg), nn.Conv2d(output_ch, output_ch, 3, stride=1, padding=1))

    def forward(self, x: Tensor) -> Tensor:
        return x + self.res_block_core(x)


class ResnetEncoder(Encoder):
    """cfg.encoder_conv_architecture = resnet_impala: three (conv3x3, maxpool/2, 2 residual blocks) stages of 16 / 32 / 32
    channels, then cfg.encoder_conv_mlp_layers"""

    def __init__(self, cfg, obs_space):
        super().__init__(cfg)
        if cfg.encoder_conv_architecture != "resnet_impala":
            raise NotImplementedError(f"Unknown resnet architecture {cfg.encoder_conv_architecture}")
        c, layers = int(obs_space.shape[0]), []
        for cout, blocks in ((16, 2), (32, 2), (32, 2)):
            layers += [nn.Conv2d(c, cout, 3, stride=1, padding=1), nn.MaxPool2d(3, stride=2, padding=1)]
            layers += [ResBlock(cfg, cout, cout) for _ in range(blocks)]
            c = cout
        act = nonlinearity(cfg)
        layers.append(act)
        self.conv_head = nn.Sequential(*layers)
        self.conv_head_out_size = calc_num_elements(self.conv_head, tuple(obs_space.shape))
        extra = list(cfg.encoder_conv_mlp_layers)
        self.mlp_layers = create_mlp(extra, self.conv_head_out_size, act)
        self.encoder_out_size = int(extra[-1]) if extra else self.conv_head_out_size

    def forward(self, obs: Tensor) -> Tensor:
        x = self.conv_head(obs)
        return self.mlp_layers(x.contiguous().view(-1, self.conv_head_out_size))

    def get_out_size(self) -> int:
        return self.encoder_out_size


def make_img_encoder(cfg, obs_space) -> Encoder:
    arch = cfg.encoder_conv_architecture
    if arch.startswith("convnet"):
        return ConvEncoder(cfg, obs_space)
    if arch.startswith("resnet"):
        return ResnetEncoder(cfg, obs_space)
    raise NotImplementedError(f"Unknown convolutional architecture {arch}")


class MultiInputEncoder(Encoder):
    """one encoder per observation key (sorted: vectors -> MlpEncoder, images -> make_img_encoder), outputs concatenated
    — the default encoder of the reference for any observation dict, one key included"""

    def __init__(self, cfg, obs_space):
        super().__init__(cfg)
        self.obs_keys = sorted(k for k in obs_space.keys() if k != "action_mask")
        self.encoders = nn.ModuleDict()
        total = 0
        for key in self.obs_keys:
            shape = obs_space[key].shape
            if len(shape) == 1:
                enc: Encoder = MlpEncoder(cfg, obs_space[key])
            elif len(shape) > 1:
                enc = make_img_encoder(cfg, obs_space[key])
            else:
                raise NotImplementedError(f"Unsupported observation space {obs_space}")
            self.encoders[key] = enc
            total += enc.get_out_size()
        self.encoder_out_size = total

    def forward(self, obs_dict: Dict[str, Tensor]) -> Tensor:
        if len(self.obs_keys) == 1:
            return self.encoders[self.obs_keys[0]](obs_dict[self.obs_keys[0]])
        return torch.cat([self.encoders[k](obs_dict[k]) for k in self.obs_keys], 1)

    def get_out_size(self) -> int:
        return self.encoder_out_size


def default_make_encoder_func(cfg, obs_space) -> Encoder:
    """what the model factory uses when no encoder factory was registered"""
    return MultiInputEncoder(cfg, obs_space)

"""Native actor-critic: encoder (conv head + MLP | MLP) -> identity core -> decoder MLP -> {critic, action params}.

Architecture, initialisation and parameter NAMES follow the reference (sample_factory/model/actor_critic.py:136-195,
model/encoder.py:72-150, model/decoder.py:15-31, model/action_parameterization.py:20-39) so `state_dict()` round-trips
with Sample Factory checkpoints; the compute is libsf_hip.so's fp32-MFMA implicit-GEMM kernels:

 * all parameters live in ONE flat fp32 buffer (segments padded to 256 B) with matching flat grad / Adam-moment buffers
   -> one launch for grad-norm, one for Adam, one bucket for the data-parallel all-reduce;
 * activations are NHWC, weights K-major [K, Cout]; conversion from/to the reference's OIHW / [out,in] layouts happens
   only in load_state_dict()/state_dict();
 * the u8 -> f32 observation normalisation (utils/normalize.py:51-70) is fused into the first layer's loader;
 * critic_linear and distribution_linear are one fused [F, 1+A] GEMM (column 0 = value).

Native: conv or MLP encoder (relu/tanh/elu), optional one-layer GRU / LSTM core (per-step cell kernels; width-512 BPTT
passes as ONE persistent launch each, csrc/sf_rnn.hip), MLP decoder, input running-mean-std (normalize_input), Discrete /
Tuple-of-Discrete / Box action heads.  Separate actor/critic weights and multi-layer RNNs raise; user-registered torch
modules and multi-key observation dicts go through model/torch_policy.py (the network under autograd, everything around
it native) — there is no silent fallback.
"""
from __future__ import annotations

import math
import threading
from typing import Dict, List, Optional

import numpy as np
import torch

from sample_factory_amd import lib
from sample_factory_amd.algo.utils.running_mean_std import RunningMeanStdInPlace
from sample_factory_amd.envs.spaces import is_box, calc_num_action_parameters

import os

_LSTM_SEQ = os.environ.get("SF_LSTM_SEQ", "1") != "0"  # A/B switch: 0 = per-step launches instead of the fused passes
_CONV1_NORM = os.environ.get("SF_CONV1_NORM", "1") != "0"  # A/B switch: 0 = normalised f32 copy of the frames + f32 conv1
_MLP2 = os.environ.get("SF_MLP2", "1") != "0"          # A/B switch: 0 = layer-by-layer encoder in the rollout as well

ACT_KIND = {"relu": 1, "tanh": 2, "elu": 3}  # model/model_utils.py:27-35; fused into the GEMM epilogues

CONV_ARCHS = {  # model/encoder.py:126-134: [out_channels, kernel, stride]
    "convnet_simple": [[32, 8, 4], [64, 4, 2], [128, 3, 2]],
    "convnet_impala": [[16, 8, 4], [32, 4, 2]],
    "convnet_atari": [[32, 8, 4], [64, 4, 2], [64, 3, 1]],
}


def get_rnn_size(cfg) -> int:
    """model/model_utils.py:11-24"""
    size = cfg.rnn_size * cfg.rnn_num_layers if cfg.use_rnn else 1
    if cfg.rnn_type == "lstm":
        size *= 2
    if not cfg.actor_critic_share_weights:
        size *= 2
    return size


def _pad(n: int, q: int = 64) -> int:
    return (n + q - 1) // q * q


class _Layer:
    """One implicit-GEMM layer: conv (spatial) or linear (1x1 on a 1x1 image)."""

    def __init__(self, name, desc: lib.sf_conv_desc, ref_w_shape, kind, first_fc_chw=None, wname=None, bname=None,
                 role="chain"):
        self.name = name          # reference parameter prefix, e.g. encoder.encoders.obs.enc.conv_head.0
        self.wname = wname or name + ".weight"
        self.bname = bname or name + ".bias"
        self.role = role          # "chain" | "rnn_ih" (x projection, followed by the cell) | "rnn_hh" (side branch)
        self.in_act_kind = 0      # activation kind of the tensor feeding this layer (fused into its dgrad epilogue)
        self.rnn_l = 0            # index of the stacked recurrent layer a "rnn_ih" / "rnn_hh" projection belongs to
        self.desc = desc
        self.kind = kind          # "conv_u8" | "conv" | "linear" | "linear_after_conv" | "heads"
        self.ref_w_shape = ref_w_shape
        self.first_fc_chw = first_fc_chw
        self.K = desc.KH * desc.KW * desc.Cin
        self.N = desc.Cout
        self.w = self.b = self.gw = self.gb = self.wt = None  # views into the flat buffers (wt: [N, K] copy)

    @property
    def out_pixels(self):
        return self.desc.OH * self.desc.OW

    # ---- layout conversion (reference <-> native)
    def w_from_ref(self, w_ref: torch.Tensor) -> torch.Tensor:
        d = self.desc
        if self.kind == "conv_u8":      # k = (c*KH + kh)*KW + kw
            return w_ref.reshape(d.Cout, self.K).t().contiguous()
        if self.kind == "conv":         # k = (kh*KW + kw)*Cin + c
            return w_ref.permute(2, 3, 1, 0).reshape(self.K, d.Cout).contiguous()
        if self.kind == "linear_after_conv":  # reference flattens NCHW, we flatten NHWC
            C, OH, OW = self.first_fc_chw
            return w_ref.view(self.N, C, OH, OW).permute(2, 3, 1, 0).reshape(self.K, self.N).contiguous()
        return w_ref.t().contiguous()   # linear: [out,in] -> [in,out]

    def w_to_ref(self, w: torch.Tensor) -> torch.Tensor:
        d = self.desc
        if self.kind == "conv_u8":
            return w.t().reshape(d.Cout, d.Cin, d.KH, d.KW).contiguous()
        if self.kind == "conv":
            return w.reshape(d.KH, d.KW, d.Cin, d.Cout).permute(3, 2, 0, 1).contiguous()
        if self.kind == "linear_after_conv":
            C, OH, OW = self.first_fc_chw
            return w.reshape(OH, OW, C, self.N).permute(3, 2, 0, 1).reshape(self.N, self.K).contiguous()
        return w.t().contiguous()


def _linear_desc(K, N, relu) -> lib.sf_conv_desc:
    return lib.sf_conv_desc(Cin=K, H=1, W=1, Cout=N, KH=1, KW=1, stride=1, OH=1, OW=1, in_u8=0, relu=int(relu),
                            traj_T=0, sub_mean=0.0, inv_scale=1.0)


class ActorCritic:
    """Shared-weights feed-forward actor-critic (reference: ActorCriticSharedWeights)."""

    def __init__(self, cfg, obs_space, action_space, device="cuda", all_reduce=None, obs_key: str = "obs", part: str = "full"):
        """part: "full" = the whole shared-weights model on the single key "obs"; "encoder" = only the encoder of observation
        key `obs_key` (no core / decoder / heads: forward_heads ends at the encoder output, backward starts from the gradient
        wrt its last layer's pre-activation); "trunk" = core + decoder + heads on a dense f32 feature batch (obs_space["obs"] =
        Box(F): the concatenated encoder outputs, already activated), whose backward also produces d(loss)/d(features)
        (`g_input`).  The two partial forms are the towers of model/actor_critic_multikey.py."""
        assert part in ("full", "encoder", "trunk")
        self.cfg = cfg
        self.obs_space = obs_space
        self.action_space = action_space
        self.device = torch.device(device)
        self.training = True
        self.part, self.obs_key = part, obs_key
        self.headless, self.features_in = part == "encoder", part == "trunk"
        self.g_input = None
        if cfg.use_rnn and (cfg.rnn_num_layers < 1 or cfg.rnn_type not in ("gru", "lstm")):
            raise NotImplementedError("native recurrent core: GRU or LSTM, rnn_num_layers >= 1")
        if not cfg.actor_critic_share_weights:
            raise NotImplementedError("separate actor/critic weights are outside the hot-path scope (SURVEY.md §2.1)")
        if cfg.nonlinearity not in ACT_KIND:
            raise NotImplementedError(f"Unknown nonlinearity {cfg.nonlinearity}")
        act = ACT_KIND[cfg.nonlinearity]
        # running input statistics exist for the keys in cfg.normalize_input_keys (None / []: all keys,
        # running_mean_std.py:113-131); this model has the single key "obs"
        keys_ = getattr(cfg, "normalize_input_keys", None)
        norm_input = bool(cfg.normalize_input) and (not keys_ or obs_key in keys_) and not self.features_in
        # ActionParameterizationContinuousNonAdaptiveStddev (action_parameterization.py:42-78): the network outputs the
        # means only, log-stddev is one learned vector.  In the fused heads GEMM the log-stddev columns keep ZERO weights
        # (their weight gradient is discarded) and their BIAS is the learned vector, so params = [means | log_std] comes
        # out of the same launch and the bias gradient (column sums) is exactly d loss / d learned_stddev.
        self.nonadaptive_std = is_box(action_space) and not cfg.adaptive_stddev and not self.headless
        # continuous_tanh_scale > 0 (non-adaptive case only, as in the reference): means = tanh(x / s) * s, applied in
        # place to the mean columns of the heads matrix (sf_tanh_scale_fwd / _bwd)
        self.tanh_scale = float(cfg.continuous_tanh_scale) if self.nonadaptive_std else 0.0
        keys = sorted(k for k in obs_space.spaces.keys() if k != "action_mask")  # obs_space_without_action_mask
        if part != "encoder" and keys != ["obs"]:
            raise NotImplementedError(f"single 'obs' key (+ optional 'action_mask') only, got {keys}")
        space = obs_space[obs_key]
        self.obs_shape = tuple(space.shape)
        self.obs_u8 = np.dtype(space.dtype) == np.uint8
        self.num_action_params = calc_num_action_parameters(action_space)
        self.layers: List[_Layer] = []
        sub_mean = float(cfg.obs_subtract_mean)
        inv_scale = float(np.float32(1.0 / cfg.obs_scale)) if abs(cfg.obs_scale - 1.0) > 1e-5 else 1.0
        if abs(sub_mean) <= 1e-5:
            sub_mean = 0.0
        if obs_key != "obs" or self.features_in:  # normalize.py:38-47: mean shift / scale belong to the key named "obs" only
            sub_mean, inv_scale = 0.0, 1.0

        self._fused_norm = False
        if len(self.obs_shape) == 3:
            C, H, W = self.obs_shape
            if not self.obs_u8:
                raise NotImplementedError("image observations must be uint8 CHW (pixel_format=CHW)")
            pfx = f"encoder.encoders.{obs_key}.enc."
            cin, h, w = C, H, W
            for i, (cout, k, s) in enumerate(CONV_ARCHS[cfg.encoder_conv_architecture]):
                oh, ow = (h - k) // s + 1, (w - k) // s + 1
                # normalize_input=True: the Nature-CNN conv1 geometry normalises INSIDE conv1's loader (sf_conv_fwd_norm /
                # sf_conv_wgrad_norm: frames stay u8, first layer keeps its raw-frame form); any other first layer reads a
                # materialised normalised f32 NHWC batch (utils/normalize.py)
                if i == 0 and norm_input and _CONV1_NORM:
                    probe = lib.sf_conv_desc(Cin=cin, H=h, W=w, Cout=cout, KH=k, KW=k, stride=s, OH=oh, OW=ow, in_u8=1,
                                             relu=act, traj_T=0, sub_mean=sub_mean, inv_scale=inv_scale)
                    self._fused_norm = bool(lib.conv_norm_supported(1, probe))
                first = i == 0 and (not norm_input or self._fused_norm)
                desc = lib.sf_conv_desc(Cin=cin, H=h, W=w, Cout=cout, KH=k, KW=k, stride=s, OH=oh, OW=ow,
                                        in_u8=int(first), relu=act, traj_T=0,
                                        sub_mean=sub_mean if first else 0.0, inv_scale=inv_scale if first else 1.0)
                self.layers.append(_Layer(f"{pfx}conv_head.{2 * i}", desc, (cout, cin, k, k),
                                          "conv_u8" if first else "conv"))
                cin, h, w = cout, oh, ow
            feat, chw = cin * h * w, (cin, h, w)
            for j, size in enumerate(cfg.encoder_conv_mlp_layers):
                self.layers.append(_Layer(f"{pfx}mlp_layers.{2 * j}", _linear_desc(feat, size, act), (size, feat),
                                          "linear_after_conv" if j == 0 else "linear", first_fc_chw=chw))
                feat = size
        elif len(self.obs_shape) == 1:
            if self.obs_u8 or ((sub_mean != 0.0 or inv_scale != 1.0) and not norm_input):
                raise NotImplementedError("vector observations must be f32; obs_scale/obs_subtract_mean on vectors "
                                          "need normalize_input=True")
            feat = self.obs_shape[0]
            for j, size in enumerate([] if self.features_in else cfg.encoder_mlp_layers):
                self.layers.append(_Layer(f"encoder.encoders.{obs_key}.mlp_head.{2 * j}", _linear_desc(feat, size, act),
                                          (size, feat), "linear"))
                feat = size
        else:
            raise NotImplementedError(f"Unsupported observation shape {self.obs_shape}")
        self.rnn_kind, self.rnn_H, self.rnn_S = None, 0, get_rnn_size(cfg)
        use_rnn = bool(cfg.use_rnn) and not self.headless
        self.rnn_L = int(cfg.rnn_num_layers) if use_rnn else 0
        self.rnn_SL = self.rnn_S // max(1, self.rnn_L)  # state columns of ONE recurrent layer: H (GRU) or 2 H (LSTM: [h | c])
        if use_rnn and not self.layers and not self.features_in:
            raise NotImplementedError("a recurrent core needs at least one encoder layer in front of it")
        if self.headless and not self.layers:
            raise NotImplementedError(f"encoder of key '{obs_key}': at least one layer needed")
        # an image encoder WITHOUT fully connected layers ends in a conv layer: its output rows are [pixel][channel] (NHWC)
        # while the reference flattens [channel][pixel] (encoder.py:117: view(-1, conv_head_out_size) of an NCHW tensor) —
        # the composite re-orders the features when it concatenates them (out_chw = (C, OH, OW))
        self.out_chw = chw if (self.headless and len(self.obs_shape) == 3 and self.layers[-1].kind in ("conv", "conv_u8")) else None
        if use_rnn:  # model/core.py:19-64: nn.GRU / nn.LSTM(input=feat, hidden=rnn_size), torch gate order
            Hs = cfg.rnn_size
            G = 3 if cfg.rnn_type == "gru" else 4
            self.rnn_kind, self.rnn_H = (0 if cfg.rnn_type == "gru" else 1), Hs
            # cfg.rnn_num_layers stacked layers (nn.GRU / nn.LSTM(input, hidden, num_layers), model/core.py:27-30): layer l > 0
            # reads layer l - 1's output; every layer is one (input projection, recurrent projection) pair on the same
            # cell / sequence kernels; its state is columns [l * SL, (l + 1) * SL) of a sample's state row (core.py:42-58)
            for l in range(int(cfg.rnn_num_layers)):
                for nm, role, kin in (("ih", "rnn_ih", feat), ("hh", "rnn_hh", Hs)):
                    Lr = _Layer(f"core.core.{nm}{l}", _linear_desc(kin, G * Hs, 0), (G * Hs, kin), "linear",
                                wname=f"core.core.weight_{nm}_l{l}", bname=f"core.core.bias_{nm}_l{l}", role=role)
                    Lr.rnn_l = l
                    self.layers.append(Lr)
                feat = Hs
        for j, size in enumerate([] if self.headless else cfg.decoder_mlp_layers):
            self.layers.append(_Layer(f"decoder.mlp.{2 * j}", _linear_desc(feat, size, act), (size, feat), "linear"))
            feat = size
        self.feat = feat
        A = self.num_action_params
        # fused heads [F, 1+A] padded to a multiple of 4 columns (zero weights, zero gradients) so that every operand
        # of every layer takes the 16-byte vector loaders; column 0 = value, columns 1..A = action parameters
        self.heads_ld = (1 + A + 3) // 4 * 4
        if not self.headless:
            self.layers.append(_Layer("heads", _linear_desc(feat, self.heads_ld, 0), (self.heads_ld, feat), "heads"))
        self.act_kind = act
        # what produced the input of each chain layer: obs (0), an activated layer (act), the RNN cell (0); a trunk's input
        # is the encoders' activated output
        prev_kind = act if self.features_in else 0
        for L in self.layers:
            if L.role == "rnn_hh":
                continue
            L.in_act_kind = prev_kind
            prev_kind = 0 if L.role == "rnn_ih" else L.desc.relu
        self.obs_elems = int(np.prod(self.obs_shape))
        # vector observations through a two-layer MLP encoder: the rollout can take the fused inference kernel
        self._mlp2_ok = (len(self.obs_shape) == 1 and len(self.layers) >= (2 if self.headless else 3) and not self.features_in and
                         all(L.kind == "linear" and L.role == "chain" and L.desc.relu == act for L in self.layers[:2]) and
                         len(cfg.encoder_mlp_layers) == 2 and
                         lib.mlp2_supported(self.obs_shape[0], self.layers[0].N, self.layers[1].N))

        # ---- flat parameter / gradient / Adam buffers
        off = 0
        self._segs = []
        for L in self.layers:
            self._segs.append((off, off + _pad(L.K * L.N)))
            off = self._segs[-1][1] + _pad(L.N)
        self.num_flat = off
        self.seat_flat(torch.zeros(off, dtype=torch.float32, device=self.device),
                       torch.zeros(off, dtype=torch.float32, device=self.device),
                       torch.zeros(off, dtype=torch.float32, device=self.device))
        self.obs_normalizer = None
        if norm_input:
            from sample_factory_amd.utils.normalize import ObservationNormalizer
            self.obs_normalizer = ObservationNormalizer(cfg, self.obs_shape, self.obs_u8, self.device,
                                                        all_reduce=all_reduce, world=getattr(cfg, "dp_world", 1))
            if obs_key != "obs":  # mean shift / scale: the key named "obs" only (normalize.py:38-47)
                self.obs_normalizer.sub_mean, self.obs_normalizer.inv_scale = 0.0, 1.0
        self._norm_prefix = f"obs_normalizer.running_mean_std.running_mean_std.{obs_key}."
        self._xn: Dict = {}
        self.returns_normalizer: Optional[RunningMeanStdInPlace] = None
        if cfg.normalize_returns and not self.headless:
            self.returns_normalizer = RunningMeanStdInPlace((1,), self.device, all_reduce=all_reduce)
        self._bufs: Dict = {}
        self._layout_gen = 0  # bumped whenever a buffer, workspace or parameter view is (re)allocated: launch_key()
        self._wss: Dict = {}
        self._tls = threading.local()  # per-thread "which scratch am I using": a sampler thread may run beside the learner
        self._snap = None
        # per-tag results of the last forward: a sampler thread ("inf*" tags) and the learner thread ("boot", "train")
        # call forward_heads concurrently, so nothing a forward leaves behind may live in an untagged attribute
        self._ctx: Dict = {}
        self._rnn_out: Dict = {}   # tag -> [(h_out, c_out | None) per recurrent layer] of the last ONE-STEP forward under that tag
        self._rnn_saved_l: Dict = {}  # layer index -> what the last training pass of that recurrent layer left for its BPTT
        self.initialize_weights()

    def seat_flat(self, flat_params: torch.Tensor, flat_grads: torch.Tensor, flat_params_t: torch.Tensor) -> None:
        """(re)build every layer's views on the given flat buffers [num_flat] — parameters, gradients and the Cout-major
        copies [N, K] of the weights of the layers the gfx950 LDS-DMA forward can take (sf_conv_fwd_t; refreshed by
        params_changed() after every parameter update).  A model composed of several of these (separate actor / critic
        weights) seats its towers on slices of ONE buffer, so that the optimiser, the gradient exchange and the snapshots
        see a single flat parameter vector.  Current values are carried over."""
        old = getattr(self, "flat_params", None)
        if old is not None:
            flat_params.copy_(old)
            flat_params_t.copy_(self.flat_params_t)
        self.flat_params, self.flat_grads, self.flat_params_t = flat_params, flat_grads, flat_params_t
        self._layout_gen = getattr(self, "_layout_gen", 0) + 1
        for L, (o, ob) in zip(self.layers, self._segs):
            L.w = flat_params[o:o + L.K * L.N].view(L.K, L.N)
            L.b = flat_params[ob:ob + L.N]
            L.gw = flat_grads[o:o + L.K * L.N].view(L.K, L.N)
            L.gb = flat_grads[ob:ob + L.N]
            # (incl. the recurrent projections W_ih / W_hh; the narrow heads matrix takes sf_conv_fwd_t's one-wave-per-16-rows kernel)
            ok = not L.desc.in_u8 and ((L.desc.Cin % 32 == 0 and L.N >= 32) or (L.kind == "heads" and L.K % 16 == 0))
            L.wt = flat_params_t[o:o + L.K * L.N].view(L.N, L.K) if ok else None

    # ------------------------------------------------------------------------------------------ reference surface
    def num_params(self) -> int:
        n = sum(L.K * L.N + L.N for L in self._body())
        return n if self.headless else n + (self.feat + 1) * (1 + self.num_action_params)

    def _body(self):
        """every layer but the fused heads (an encoder tower has none)"""
        return self.layers if self.headless else self.layers[:-1]

    # ---- what a composite (model/actor_critic_separate.py) needs from a tower, whatever the tower is made of
    @property
    def heads_layer(self):
        return self.layers[-1]

    def share_normalizers_from(self, other) -> None:
        """input / return normalisers belong to the MODEL (actor_critic.py:44-61): a second tower uses the first one's"""
        self.obs_normalizer = other.obs_normalizer
        self.returns_normalizer = None

    def share_seq_sync_from(self, other) -> None:
        self._bufs[("rnn", "seq_sync")] = other._seq_sync_buf()

    def share_snapshot_tables_from(self, other) -> None:
        self._snap_tabs = other._snap_tabs

    def normalizer_state(self) -> Dict[str, torch.Tensor]:
        return self.obs_normalizer.state_dict(self._norm_prefix) if self.obs_normalizer is not None else {}

    def load_normalizer_state(self, sd) -> None:
        if self.obs_normalizer is not None and self._norm_prefix + "count" in sd:
            self.obs_normalizer.load_state_dict(sd, self._norm_prefix)

    def train(self, mode=True):
        self.training = mode
        if self.returns_normalizer is not None:
            self.returns_normalizer.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def model_to_device(self, device):
        assert torch.device(device).type == "cuda", "the native model only lives on the GPU"

    def normalize_obs(self, obs):
        """Identity: normalisation is fused into the first layer (the f32 copy is never materialised)."""
        return obs

    def initialize_weights(self):
        """actor_critic.py:73-96: orthogonal (gain) / xavier_uniform / torch_default on the REFERENCE layout; bias 0."""
        cfg = self.cfg
        gain = cfg.policy_init_gain
        sd = {}
        for name, shape in self.ref_param_shapes():
            t = torch.empty(shape, dtype=torch.float32)
            if name.endswith("learned_stddev"):  # action_parameterization.py:58-60
                t.fill_(math.log(cfg.initial_stddev))
            elif name.startswith("core.core."):  # nn.GRU/nn.LSTM keep torch's default init (initialize_weights skips them)
                bound = 1.0 / math.sqrt(self.rnn_H)
                t.uniform_(-bound, bound)
            elif name.endswith(".bias"):
                if cfg.policy_initialization == "torch_default":
                    fan_in = int(np.prod(dict(self.ref_param_shapes())[name[:-4] + "weight"][1:]))
                    bound = 1 / math.sqrt(fan_in)
                    t.uniform_(-bound, bound)
                else:
                    t.zero_()
            elif cfg.policy_initialization == "orthogonal":
                torch.nn.init.orthogonal_(t, gain=gain)
            elif cfg.policy_initialization == "xavier_uniform":
                torch.nn.init.xavier_uniform_(t, gain=gain)
            else:
                torch.nn.init.kaiming_uniform_(t, a=math.sqrt(5))
            sd[name] = t
        self.load_state_dict(sd, strict=False)

    def ref_param_shapes(self):
        """(name, shape) of every trainable parameter under the reference's names, in the reference's order."""
        out = []
        for L in self._body():
            if L.role == "rnn_hh":
                continue
            if L.role == "rnn_ih":  # torch order: weight_ih, weight_hh, bias_ih, bias_hh
                Lh = self.layers[self.layers.index(L) + 1]
                out += [(L.wname, tuple(L.ref_w_shape)), (Lh.wname, tuple(Lh.ref_w_shape)), (L.bname, (L.N,)),
                        (Lh.bname, (Lh.N,))]
                continue
            out.append((L.wname, tuple(L.ref_w_shape)))
            out.append((L.bname, (L.N,)))
        if self.headless:
            return out
        A, F = self.num_action_params, self.feat
        out += [("critic_linear.weight", (1, F)), ("critic_linear.bias", (1,))]
        if self.nonadaptive_std:  # torch lists a module's own parameters before its children's
            out += [("action_parameterization.learned_stddev", (A // 2,)),
                    ("action_parameterization.distribution_linear.weight", (A // 2, F)),
                    ("action_parameterization.distribution_linear.bias", (A // 2,))]
        else:
            out += [("action_parameterization.distribution_linear.weight", (A, F)),
                    ("action_parameterization.distribution_linear.bias", (A,))]
        return out

    def state_dict(self) -> Dict[str, torch.Tensor]:
        sd = {}
        if self.obs_normalizer is not None:
            sd.update(self.obs_normalizer.state_dict(self._norm_prefix))
        if self.returns_normalizer is not None:
            sd.update(self.returns_normalizer.state_dict("returns_normalizer."))
        for L in self._body():
            sd[L.wname] = L.w_to_ref(L.w.detach()).cpu()
            sd[L.bname] = L.b.detach().cpu().clone()
        if self.headless:
            return sd
        H, A = self.layers[-1], self.num_action_params
        w = H.w.detach().cpu()
        sd["critic_linear.weight"] = w[:, 0:1].t().contiguous()
        sd["critic_linear.bias"] = H.b.detach().cpu()[0:1].clone()
        Aw = A // 2 if self.nonadaptive_std else A  # columns the network really produces
        sd["action_parameterization.distribution_linear.weight"] = w[:, 1:1 + Aw].t().contiguous()
        sd["action_parameterization.distribution_linear.bias"] = H.b.detach().cpu()[1:1 + Aw].clone()
        if self.nonadaptive_std:
            sd["action_parameterization.learned_stddev"] = H.b.detach().cpu()[1 + Aw:1 + A].clone()
        return sd

    def load_state_dict(self, sd, strict=True):
        with torch.no_grad():
            for L in self._body():
                L.w.copy_(L.w_from_ref(torch.as_tensor(sd[L.wname], dtype=torch.float32)))
                L.b.copy_(torch.as_tensor(sd[L.bname], dtype=torch.float32))
            if self.headless:
                self.params_changed()
                if self.obs_normalizer is not None and self._norm_prefix + "count" in sd:
                    self.obs_normalizer.load_state_dict(sd, self._norm_prefix)
                return
            H, A = self.layers[-1], self.num_action_params
            cw = torch.as_tensor(sd["critic_linear.weight"], dtype=torch.float32)
            aw = torch.as_tensor(sd["action_parameterization.distribution_linear.weight"], dtype=torch.float32)
            pad = self.heads_ld - 1 - A
            ab = torch.as_tensor(sd["action_parameterization.distribution_linear.bias"], dtype=torch.float32).reshape(-1)
            if self.nonadaptive_std:  # log-stddev columns: zero weights, bias = the learned vector
                aw = torch.cat([aw, torch.zeros((A // 2, self.feat))], dim=0)
                ab = torch.cat([ab, torch.as_tensor(sd["action_parameterization.learned_stddev"],
                                                    dtype=torch.float32).reshape(-1)])
            H.w.copy_(torch.cat([cw, aw, torch.zeros((pad, self.feat))], dim=0).t().contiguous())
            H.b.copy_(torch.cat([torch.as_tensor(sd["critic_linear.bias"], dtype=torch.float32).reshape(1), ab,
                                 torch.zeros(pad)]))
            self.params_changed()
            if self.obs_normalizer is not None and self._norm_prefix + "count" in sd:
                self.obs_normalizer.load_state_dict(sd, self._norm_prefix)
            if self.returns_normalizer is not None and "returns_normalizer.running_mean" in sd:
                self.returns_normalizer.load_state_dict(sd, "returns_normalizer.")
            elif strict and self.returns_normalizer is not None:
                raise KeyError("returns_normalizer.* missing from state dict")

    # flat <-> reference order (for optimizer state interop and parity tests)
    def flat_to_ref(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Interpret a flat buffer (params, grads or Adam moments) under the reference's names/layouts."""
        out = {}
        for L, (o, ob) in zip(self._body(), self._segs):
            out[L.wname] = L.w_to_ref(flat[o:o + L.K * L.N].view(L.K, L.N)).cpu()
            out[L.bname] = flat[ob:ob + L.N].cpu().clone()
        if self.headless:
            return out
        H, (o, ob), A = self.layers[-1], self._segs[-1], self.num_action_params
        w = flat[o:o + H.K * H.N].view(H.K, H.N).cpu()
        b = flat[ob:ob + H.N].cpu()
        out["critic_linear.weight"] = w[:, 0:1].t().contiguous()
        out["critic_linear.bias"] = b[0:1].clone()
        Aw = A // 2 if self.nonadaptive_std else A
        out["action_parameterization.distribution_linear.weight"] = w[:, 1:1 + Aw].t().contiguous()
        out["action_parameterization.distribution_linear.bias"] = b[1:1 + Aw].clone()
        if self.nonadaptive_std:
            out["action_parameterization.learned_stddev"] = b[1 + Aw:1 + A].clone()
        return out

    # ------------------------------------------------------------------------------------------ compute
    def _buf(self, key, shape, dtype=torch.float32):
        t = self._bufs.get(key)
        if t is None or t.shape != torch.Size(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._bufs[key] = t
            self._layout_gen += 1
        return t

    def launch_key(self, tag: str = "inf"):
        """identity of every address a one-step forward under `tag` hands the library (weights or the published snapshot
        it reads, activation buffers, workspaces): a recorded launch program (lib.LaunchProgram) of that forward is valid
        exactly as long as this value does not change"""
        return (self._layout_gen, self.snap_read if self._snap is not None else -1)

    def _aligned_frames(self, tag, x, stride, idx, off, tT, n):
        """the n frames a launch would read — dataset row d = idx[i] | off + i, slab row d + d // traj_T (sf_common.h
        sample_base) — gathered into a dense, allocator-aligned u8 [n, obs_elems] buffer"""
        d = idx.long() if idx is not None else torch.arange(off, off + n, device=x.device)
        pos = d + torch.div(d, int(tT), rounding_mode="floor") if tT else d
        avail = (x.untyped_storage().nbytes() - x.storage_offset() * x.element_size()) // x.element_size()
        rows = (avail - self.obs_elems) // int(stride) + 1
        flat = x.as_strided((rows, self.obs_elems), (int(stride), 1))
        out = self._buf((tag, "frames_aligned"), (n, self.obs_elems), dtype=x.dtype)
        lib.recording_unsafe("frames gathered by a torch op")
        torch.index_select(flat, 0, pos, out=out)
        return out, self.obs_elems, None, 0, 0

    def _zbuf(self, key, shape):
        """like _buf but zero-filled on creation (buffers with never-written padding columns)"""
        t = self._bufs.get(key)
        if t is None or t.shape != torch.Size(shape):
            t = torch.zeros(shape, dtype=torch.float32, device=self.device)
            self._bufs[key] = t
            self._layout_gen += 1
        return t

    def tensor_segment_ids(self):
        """(seg_id u8 [num_flat], num_segments): which REFERENCE parameter tensor every flat element belongs to
        (per-tensor statistics of Lamb, optimizers.py:108-135); 255 = padding.  The fused heads matrix [feat, 1+A+pad]
        holds two reference tensors column-wise: critic_linear (column 0) and distribution_linear (columns 1..A)."""
        seg = torch.full((self.num_flat,), 255, dtype=torch.uint8)
        nseg = 0
        for L, (o, ob) in zip(self._body(), self._segs):
            seg[o:o + L.K * L.N] = nseg
            seg[ob:ob + L.N] = nseg + 1
            nseg += 2
        if self.headless:
            return seg.to(self.device), nseg
        H, (o, ob), A = self.layers[-1], self._segs[-1], self.num_action_params
        hw = torch.full((H.K, H.N), 255, dtype=torch.uint8)
        hw[:, 0] = nseg          # critic_linear.weight
        Aw = A // 2 if self.nonadaptive_std else A
        hw[:, 1:1 + Aw] = nseg + 2  # distribution_linear.weight (the frozen zero columns of learned_stddev stay 255)
        seg[o:o + H.K * H.N] = hw.reshape(-1)
        hb = torch.full((H.N,), 255, dtype=torch.uint8)
        hb[0] = nseg + 1          # critic_linear.bias
        hb[1:1 + Aw] = nseg + 3   # distribution_linear.bias
        if self.nonadaptive_std:
            hb[1 + Aw:1 + A] = nseg + 4  # learned_stddev
        seg[ob:ob + H.N] = hb
        return seg.to(self.device), nseg + (5 if self.nonadaptive_std else 4)

    def params_changed(self) -> None:
        """call after ANY write to flat_params (optimiser step, load_state_dict, broadcast): refresh derived copies"""
        for L in self.layers:
            if L.wt is not None:
                lib.transpose(L.w, L.wt, L.K, L.N)

    def _workspace(self, nbytes: int) -> torch.Tensor:
        """split-K / wgrad scratch; one per stream role so that the rollout and learner streams never share it"""
        key = getattr(self._tls, "role", "learner")
        ws = self._wss.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
            self._wss[key] = ws
            self._layout_gen += 1
        return ws

    # ---- async mode (cfg.async_rl): inference reads a published SNAPSHOT of the weights (K20).  The reference copies
    # the state_dict into every inference worker under policy_lock (model_sharing.py:128-143); here the learner
    # ping-pongs two device copies so a rollout never sees a half-updated buffer and nothing blocks.
    def enable_weight_snapshots(self) -> None:
        self._snap = [self.flat_params.clone(), self.flat_params.clone()]
        self._snap_t = [self.flat_params_t.clone(), self.flat_params_t.clone()]
        self._snap_views = []
        for buf, buft in zip(self._snap, self._snap_t):
            self._snap_views.append([(buf[o:o + L.K * L.N].view(L.K, L.N), buf[ob:ob + L.N],
                                      buft[o:o + L.K * L.N].view(L.N, L.K) if L.wt is not None else None)
                                     for L, (o, ob) in zip(self.layers, self._segs)])
        self.snap_read = 0
        self._layout_gen += 1
        on = self.obs_normalizer
        self._snap_tabs = [(on.mu_tab.clone(), on.rstd_tab.clone()) for _ in range(2)] if on is not None else None

    def publish_weights(self, slot: int) -> None:
        self._snap[slot].copy_(self.flat_params)
        self._snap_t[slot].copy_(self.flat_params_t)
        if self._snap_tabs is not None:
            self._snap_tabs[slot][0].copy_(self.obs_normalizer.mu_tab)
            self._snap_tabs[slot][1].copy_(self.obs_normalizer.rstd_tab)

    def _wb(self, li, tag):
        if tag.startswith("inf") and self._snap is not None:
            return self._snap_views[self.snap_read][li]
        L = self.layers[li]
        return L.w, L.b, L.wt

    def _gemm(self, li, x, stride, index, offset, traj_T, out, n, tag):
        L = self.layers[li]
        d = L.desc
        if traj_T:
            d = lib.sf_conv_desc.from_buffer_copy(L.desc)
            d.traj_T = int(traj_T)
        w, b, wt = self._wb(li, tag)
        if wt is not None and index is None and not traj_T and lib.conv_fwd_t_supported(n, d):
            wsb = lib.conv_fwd_t_workspace(n, d)  # non-zero: wide layer, few rows -> split along K
            lib.conv_fwd_t(x, stride, wt, b, out, n, d, self._workspace(wsb) if wsb else None)  # LDS-DMA kernel
            return
        wsb = lib.conv_fwd_workspace(n, d) if index is None and not traj_T else 0  # split-K for chip-starving launches
        lib.conv_fwd_raw(x, stride, index, offset, w, b, out, n, d, self._workspace(wsb) if wsb else None)

    def forward_heads(self, obs: torch.Tensor, n: int, *, sample_stride: int, index=None, offset: int = 0,
                      traj_T: int = 0, tag="inf", rnn=None) -> List[torch.Tensor]:
        """Run the whole stack on `n` samples; returns the list of layer outputs (last = heads [n, heads_ld]).

        obs: any (possibly strided) view whose data_ptr is sample 0; logical sample i lives at row
        (index[i] | offset+i) [-> slab row if traj_T] * sample_stride elements.
        rnn (recurrent models): {"states": [n, S] view} for one inference step (new state -> new_rnn_parts_of(tag)) or
        {"R": recurrence, "h0": [n/R, S], "keep_tm": [R, n/R]} for a training pass over recurrence-length chunks.
        """
        acts: List[Optional[torch.Tensor]] = [None] * len(self.layers)
        inputs: List[Optional[torch.Tensor]] = [None] * len(self.layers)
        relu_mask0 = None
        self._tls.role = "rollout" + tag[3:] if tag.startswith("inf") else "learner"  # "inf", "inf1", ...: env groups
        x, stride, idx, off, tT = obs, sample_stride, index, offset, traj_T
        first_layer = 0
        if self._mlp2_ok and tag.startswith("inf") and index is None and not traj_T and _MLP2:
            # rollout on vector observations: normalisation + the two encoder layers in ONE launch (sf_mlp2_fwd); the
            # learner keeps the layer kernels (it needs the intermediate activations for the backward pass)
            on = self.obs_normalizer
            tabs = None
            if on is not None:
                tabs = self._snap_tabs[self.snap_read] if self._snap is not None else (on.mu_tab, on.rstd_tab)
            (w1, b1, _), (w2, b2, _) = self._wb(0, tag), self._wb(1, tag)
            out = self._buf((tag, 1), (n, self.layers[1].N))
            lib.mlp2_fwd(obs, sample_stride, n, self.obs_elems, on.sub_mean if on is not None else 0.0,
                         on.inv_scale if on is not None else 1.0, tabs[0] if tabs else None, tabs[1] if tabs else None,
                         w1, b1, w2, b2, self.act_kind, out)
            acts[1] = out
            x, stride, idx, off, tT = out, self.layers[1].N, None, 0, 0
            first_layer = 2
        norm_tabs = None
        if self.obs_normalizer is not None and first_layer == 0:  # normalize_input=True
            on = self.obs_normalizer
            tabs = self._snap_tabs[self.snap_read] if (tag.startswith("inf") and self._snap is not None) else None
            if self._fused_norm:
                # image frames: (x - mu) * rstd, clamped, happens in conv1's loader (sf_conv_fwd_norm) — the frames stay
                # u8 in the slab and no normalised f32 copy is written or read
                norm_tabs = tabs if tabs is not None else (on.mu_tab, on.rstd_tab)
                if x.data_ptr() % 4 or stride % 4:
                    # the loader fetches the bytes as 32-bit words: a frame view at an odd address (a custom slab offset)
                    # degrades to one aligned u8 copy of the batch's frames instead of failing the launch
                    x, stride, idx, off, tT = self._aligned_frames(tag, x, stride, idx, off, tT, n)
            else:  # any other shape: materialise the normalised f32 batch (NHWC), as the reference does
                xn = self._buf((tag, "obsn"), (n, self.obs_elems))
                on.apply(obs, sample_stride, n, xn, index=index, offset=offset, traj_T=traj_T, tabs=tabs)
                x, stride, idx, off, tT = xn, self.obs_elems, None, 0, 0
        first_in = (x, stride, idx, off, tT)
        seq = rnn is not None and "R" in rnn
        for li, L in enumerate(self.layers):
            if L.role == "rnn_hh" or li < first_layer:
                continue
            if L.role == "rnn_ih" and seq:  # BPTT pass: the recurrent block works time-major ([R, C, .])
                R, Cn = rnn["R"], n // rnn["R"]
                xt = self._buf((tag, "x_tm", li), (n, L.K))
                xt.view(R, Cn, L.K).copy_(x.view(Cn, R, L.K).transpose(0, 1))
                x = xt
            inputs[li] = x
            out = self._buf((tag, li), (n * L.out_pixels, L.N))
            mask = None
            if tag == "train" and li == 0 and norm_tabs is None and li + 1 < len(self.layers) and self.layers[1].role == "chain":
                d0 = L.desc
                if tT:
                    d0 = lib.sf_conv_desc.from_buffer_copy(L.desc)
                    d0.traj_T = int(tT)
                w_, b_, _ = self._wb(li, tag)
                aligned = x.data_ptr() % 4 == 0 and stride % 4 == 0 and w_.data_ptr() % 16 == 0 and out.data_ptr() % 16 == 0
                if aligned and lib.conv_relu_mask_supported(n, d0):  # (the launch fails hard on operands it cannot take)
                    # conv1 on raw frames: the forward also records ONE sign bit per output element; the backward pass
                    # masks with these bits inside conv1's weight-gradient kernel, so conv2's data gradient never
                    # re-reads this activation
                    mask = self._buf((tag, "relu_mask0"), (n * L.out_pixels,), dtype=torch.int32)
                    lib.conv_fwd_relu_mask(x, stride, idx, off, w_, b_, out, mask, n, d0)
            fuse_x = False
            if L.role == "rnn_ih" and seq and _LSTM_SEQ and L.wt is not None and idx is None and not tT:
                # BPTT pass whose fused sequence kernel can also do the input projection (sf_*_seq_fwd_x): no GEMM launch,
                # no [n, G*H] round trip through memory for gx
                fuse_x = lib.lstm_seq_supported(n // rnn["R"], self.rnn_H) and lib.seq_fwd_x_supported(n // rnn["R"], self.rnn_H, L.K)
            fuse_dual = False
            if L.role == "rnn_ih" and rnn is not None and not seq and idx is None and not tT:
                # one recurrent inference step: x W_ih^T and h W_hh^T in ONE launch (sf_linear_fwd_dual) — the 64-deep input
                # projection was a launch of its own that is all fill and drain (GRU: the candidate gate's parts stay apart)
                fuse_dual = (self._wb(li, tag)[2] is not None and self._wb(li + 1, tag)[2] is not None and
                             self.rnn_H % 64 == 0 and lib.linear_fwd_dual_supported(n, 4 * self.rnn_H, L.K, self.rnn_H))
            if li == 0 and norm_tabs is not None:
                w_, b_, _ = self._wb(li, tag)
                dn = L.desc
                if tT:
                    dn = lib.sf_conv_desc.from_buffer_copy(L.desc)
                    dn.traj_T = int(tT)
                lib.conv_fwd_norm(x, stride, idx, off, norm_tabs[0], norm_tabs[1], w_, b_, out, n, dn)
            elif mask is None and not fuse_x and not fuse_dual:
                self._gemm(li, x, stride, idx, off, tT, out, n, tag)
            if li == 0:
                relu_mask0 = mask  # kept in this forward's OWN context (below): nothing a forward leaves behind is untagged
            acts[li] = out
            x = out
            if L.role == "rnn_ih":
                if fuse_x:
                    acts[li] = None  # gx is never materialised
                    x = self._rnn_sequence_fwd(li, None, n, rnn, tag, x_tm=inputs[li])
                elif fuse_dual:
                    acts[li] = None
                    x = self._rnn_step(li, None, n, rnn, tag, x_in=inputs[li], x_stride=stride)
                else:
                    x = self._rnn_sequence_fwd(li, out, n, rnn, tag) if seq else self._rnn_step(li, out, n, rnn, tag)
            stride, idx, off, tT = x.numel() // n, None, 0, 0  # elements per sample of the activation just produced
        if self.tanh_scale > 0:  # action_parameterization.py:62-66 (col 0 = value, then the means)
            lib.tanh_scale_fwd(acts[-1], self.heads_ld, n, 1, self.num_action_params // 2, self.tanh_scale)
        self._ctx[tag] = dict(acts=acts, inputs=inputs, first_in=first_in, rnn=rnn, relu_mask0=relu_mask0,
                              norm_tabs=norm_tabs)
        return acts

    # ------------------------------------------------------------------------------------------ recurrent core
    def _rnn_step(self, li, gx, n, rnn, tag, x_in=None, x_stride=0):
        """one inference step: (gx, h W_hh^T + b_hh) -> cell -> new state (model/core.py:37-64)"""
        Lh, H, S, kind = self.layers[li + 1], self.rnn_H, self.rnn_S, self.rnn_kind
        l, SL = self.layers[li].rnn_l, self.rnn_SL
        assert rnn["states"].shape == (n, S) and rnn["states"].stride(1) == 1
        st = rnn["states"][:, l * SL:(l + 1) * SL]  # this layer's [h | c] columns of the state rows (strided view)
        gh = self._buf((tag, "gh", l), (n, Lh.N))
        w_hh, b_hh, wt_hh = self._wb(li + 1, tag)
        if x_in is not None:  # both projections and both biases in one launch; the cell reads the [n, 4H] pre-activations
            _, b_ih, wt_ih = self._wb(li, tag)
            gpre = self._buf((tag, "gpre", l), (n, 4 * H))
            lib.linear_fwd_dual(x_in, x_stride, wt_ih, b_ih, st, st.stride(0), wt_hh, b_hh, gpre, n, gru_H=H if kind == 0 else 0)
            gx, gh = gpre, None
        elif wt_hh is not None and lib.conv_fwd_t_supported(n, Lh.desc):  # LDS-DMA GEMM (2048 envs x 512 x 2048 fills the chip)
            wsb = lib.conv_fwd_t_workspace(n, Lh.desc)
            lib.conv_fwd_t(st, st.stride(0), wt_hh, b_hh, gh, n, Lh.desc, self._workspace(wsb) if wsb else None)
        else:
            lib.conv_fwd_raw(st, st.stride(0), None, 0, w_hh, b_hh, gh, n, Lh.desc)
        h_out = self._buf((tag, "h_out", l), (n, H))
        c_out = self._buf((tag, "c_out", l), (n, H)) if kind == 1 else None
        lib.rnn_cell_fwd(kind, gx, gh, st, st.stride(0), st[:, H:] if kind == 1 else None, st.stride(0), None, n, H,
                         None, h_out, c_out, None, None)
        # keyed by tag: the rollout runner masks + stores ITS OWN step's state in one launch (sf_rnn_store_state) even
        # while the learner thread runs the bootstrap forward (tag "boot") through the same model object
        if l == 0:
            self._rnn_out[tag] = []
        self._rnn_out[tag].append((h_out, c_out))
        return h_out

    def _rnn_sequence_fwd(self, li, GX, n, rnn, tag, x_tm=None):
        """training pass: masked time loop over recurrence-length chunks (state zeroed after done/invalid steps —
        the loop form of rnn_utils.py:114-158, see tests/algo/test_rnn.py in the reference)"""
        Lh, H, kind = self.layers[li + 1], self.rnn_H, self.rnn_kind
        R, Cn = rnn["R"], n // rnn["R"]
        l, SL = self.layers[li].rnn_l, self.rnn_SL
        h0, keep = rnn["h0"][:, l * SL:(l + 1) * SL], rnn["keep_tm"]  # this layer's columns of the chunk-start states
        GH = Lh.N
        gates = self._buf((tag, "gates", li), (R, Cn, 4 * H))
        Hprev = self._buf((tag, "Hprev", li), (R + 1, Cn, H))
        Cprev = self._buf((tag, "Cprev", li), (R + 1, Cn, H)) if kind == 1 else None
        Cout = self._buf((tag, "Cout", li), (R, Cn, H)) if kind == 1 else None
        lib.copy_rows(Hprev[0], h0[:, :H])  # (the library's own row-copy kernel: [h | c] columns of the chunk-start states)
        if kind == 1:
            lib.copy_rows(Cprev[0], h0[:, H:])
        out = self._buf((tag, "core_out", li), (n, H))
        fused = _LSTM_SEQ and lib.lstm_seq_supported(Cn, H)
        if fused:  # ONE persistent launch for the whole time loop (csrc/sf_rnn.hip): W_hh slices resident in LDS; the
            # core output is written in the minibatch's own row order (chunk-major), no transpose copy
            sync = self._seq_sync_buf()
            Li = self.layers[li]
            if x_tm is not None and kind == 1:   # ... and the input projection x W_ih^T + b_ih inside the same launch
                lib.lstm_seq_fwd_x(x_tm, Li.wt, Li.b, Lh.w, Lh.b, keep, gates, Hprev, out, Cprev, Cout, sync, R, Cn, H, env_major=True)
            elif x_tm is not None:
                lib.gru_seq_fwd_x(x_tm, Li.wt, Li.b, Lh.w, Lh.b, keep, gates, Hprev, out, sync, R, Cn, H, env_major=True)
            elif kind == 1:
                lib.lstm_seq_fwd(GX, Lh.w, Lh.b, keep, gates, Hprev, out, Cprev, Cout, sync, R, Cn, H, env_major=True)
            else:
                lib.gru_seq_fwd(GX, Lh.w, Lh.b, keep, gates, Hprev, out, sync, R, Cn, H, env_major=True)
            self._rnn_saved_l[li] = self._rnn_saved = dict(gates=gates, Hprev=Hprev, Cprev=Cprev, Cout=Cout, keep=keep, R=R, Cn=Cn, fused=True)
            return out
        Hout = self._buf((tag, "Hout", li), (R, Cn, H))
        gh = self._buf((tag, "gh_seq", li), (Cn, GH))
        GXv = GX.view(R, Cn, GH)
        for t in range(R):
            lib.conv_fwd_raw(Hprev[t], H, None, 0, Lh.w, Lh.b, gh, Cn, Lh.desc)
            lib.rnn_cell_fwd(kind, GXv[t], gh, Hprev[t], H, Cprev[t] if kind == 1 else None, H, keep[t], Cn, H,
                             gates[t], Hout[t], Cout[t] if kind == 1 else None, Hprev[t + 1],
                             Cprev[t + 1] if kind == 1 else None)
        out.view(Cn, R, H).copy_(Hout.transpose(0, 1))
        self._rnn_saved_l[li] = self._rnn_saved = dict(gates=gates, Hprev=Hprev, Cprev=Cprev, Cout=Cout, keep=keep, R=R, Cn=Cn, fused=False)
        return out

    def _rnn_sequence_bwd(self, li, d_core, n):
        """BPTT: returns dL/d(gx) time-major [R*C, G*H]; accumulates the W_hh / b_hh gradients"""
        Lh, H, kind = self.layers[li + 1], self.rnn_H, self.rnn_kind
        sv = self._rnn_saved_l[li]
        R, Cn, keep = sv["R"], sv["Cn"], sv["keep"]
        GH = Lh.N
        dGX = self._buf(("g", "dGX", li), (R, Cn, GH))
        if sv.get("fused"):  # the whole backward time loop in one persistent launch (cell backward + W_hh^T product + carries)
            sync = self._seq_sync_buf()
            dGH = dGX
            dOut = d_core if d_core.is_contiguous() else d_core.contiguous()  # read in the minibatch's row order
            if kind == 1:
                lib.lstm_seq_bwd(dOut, sv["gates"], sv["Cprev"], sv["Cout"], keep, Lh.w, dGX, sync, R, Cn, H, env_major=True)
            else:  # GRU: the candidate gate's recurrent part is scaled by r -> W_hh sees its own gate gradients
                dGH = self._buf(("g", "dGH", li), (R, Cn, GH))
                lib.gru_seq_bwd(dOut, sv["gates"], sv["Hprev"], keep, Lh.w, dGX, dGH, sync, R, Cn, H, env_major=True)
            ws = self._workspace(lib.conv_wgrad_workspace(n, Lh.desc))
            lib.conv_wgrad_raw(sv["Hprev"][:R].reshape(n, H), H, None, 0, dGH.view(n, GH), Lh.gw, Lh.gb, n, Lh.desc, ws)
            return dGX.view(n, GH)
        dOut = self._buf(("g", "dOut_tm", li), (R, Cn, H))
        dOut.copy_(d_core.view(Cn, R, H).transpose(0, 1))
        dGH = self._buf(("g", "dGH", li), (R, Cn, GH)) if kind == 0 else dGX
        dh = self._buf(("g", "dh", li), (Cn, H))
        dh_direct = self._buf(("g", "dh_direct", li), (Cn, H)) if kind == 0 else None
        dhW = self._buf(("g", "dhW", li), (Cn, H))
        carry_h = self._buf(("g", "carry_h", li), (Cn, H))
        carry_c = self._buf(("g", "carry_c", li), (Cn, H)) if kind == 1 else None
        dc_prev = self._buf(("g", "dc_prev", li), (Cn, H)) if kind == 1 else None
        for t in range(R - 1, -1, -1):
            last = t == R - 1
            lib.rows_add_scale(dOut[t], None if last else carry_h, None, Cn, H, dh)
            lib.rnn_cell_bwd(kind, dh, None if (last or kind == 0) else carry_c, sv["gates"][t], sv["Hprev"][t], H,
                             sv["Cprev"][t] if kind == 1 else None, H, sv["Cout"][t] if kind == 1 else None, Cn, H,
                             dGX[t], dGH[t] if kind == 0 else None, dh_direct, dc_prev)
            if t > 0:  # gradient wrt the state entering step t, masked by keep[t-1] (state was zeroed after a done)
                lib.conv_dgrad(dGH[t], Lh.w, None, dhW, Cn, Lh.desc)
                lib.rows_add_scale(dhW, dh_direct, keep[t - 1], Cn, H, carry_h)
                if kind == 1:
                    lib.rows_add_scale(dc_prev, None, keep[t - 1], Cn, H, carry_c)
        ws = self._workspace(lib.conv_wgrad_workspace(n, Lh.desc))
        lib.conv_wgrad_raw(sv["Hprev"][:R].reshape(n, H), H, None, 0, dGH.view(n, GH), Lh.gw, Lh.gb, n, Lh.desc, ws)
        return dGX.view(n, GH)

    def new_rnn_parts_of(self, tag: str = "inf"):
        """[(h, c | None) per recurrent layer] produced by the last one-step forward issued under `tag` (None if there was
        none); layer l's parts belong into columns [l * rnn_SL, (l + 1) * rnn_SL) of a state row"""
        return self._rnn_out.get(tag)

    def new_rnn_states_of(self, tag: str = "inf") -> torch.Tensor:
        """[B, S] state after the last one-step forward under `tag` (reference output key `new_rnn_states`; stacked layers:
        [h0 | c0 | h1 | c1 ...], model/core.py:54-58)"""
        return torch.cat([t for h, c in self._rnn_out[tag] for t in ((h,) if c is None else (h, c))], dim=1)

    @property
    def new_rnn_states(self) -> torch.Tensor:
        """the default inference tag's state (ActorCritic.forward())"""
        return self.new_rnn_states_of("inf")

    def _seq_sync_buf(self) -> torch.Tensor:
        """hand-off counters (words 0..127, zeroed by every launch) + the STICKY abort word (word 128) of the fused
        sequence passes; one buffer for the forward and the backward pass (they run back to back on one stream)"""
        t = self._bufs.get(("rnn", "seq_sync"))
        if t is None:
            t = self._bufs[("rnn", "seq_sync")] = torch.zeros(192, dtype=torch.int32, device=self.device)
        return t

    def rnn_abort_word(self) -> Optional[torch.Tensor]:
        """int32 [1] view of the sticky abort word (None: this model has no fused recurrent passes).  The optimiser
        kernels take it as their skip flag: once a pass has aborted, no later SGD step of the call touches the weights."""
        return self._seq_sync_buf()[128:129] if (self.rnn_kind is not None and _LSTM_SEQ) else None

    def rnn_abort_clear(self) -> None:
        if self.rnn_kind is not None and _LSTM_SEQ:
            self._seq_sync_buf()[128:129].zero_()

    def rnn_pass_aborted(self) -> bool:
        """True if a fused sequence pass gave up waiting for a work-group since the last rnn_abort_clear() (GPU shared
        with another process): its results are garbage.  One 4-byte readback; the Learner calls it where it reads the
        epoch's loss scalars back anyway."""
        t = self._bufs.get(("rnn", "seq_sync"))
        return t is not None and bool(int(t[128].item()))

    def forward(self, normalized_obs_dict, rnn_states=None, values_only: bool = False, action_mask=None):
        """Inference-style forward on a dense obs batch [B, ...]; returns dict(values, action_logits, new_rnn_states)
        (GPU tensors).  Sampling is a separate fused kernel (sf_sample_write_step) driven by the rollout runner."""
        obs = normalized_obs_dict["obs"] if isinstance(normalized_obs_dict, dict) else normalized_obs_dict
        # action masks only affect SAMPLING (actor_critic.py:169-181); the rollout runner hands them to the sampler kernel
        B = obs.shape[0]
        rnn = dict(states=rnn_states) if self.rnn_kind is not None else None
        heads = self.forward_heads(obs, B, sample_stride=self.obs_elems if obs.is_contiguous() else obs.stride(0),
                                   rnn=rnn)[-1]
        res = dict(values=heads[:, 0])
        if not values_only:
            res["action_logits"] = heads[:, 1:1 + self.num_action_params]
        res["new_rnn_states"] = self.new_rnn_states if self.rnn_kind is not None else rnn_states
        return res

    def backward(self, acts: List[torch.Tensor], g_heads: torch.Tensor, obs: torch.Tensor, n: int, *,
                 sample_stride: int, index=None, offset: int = 0, traj_T: int = 0, on_layer_done=None) -> None:
        """Back-propagate d(loss)/d(heads) [n, heads_ld] through the stack of the last "train" forward into
        self.flat_grads (overwritten).  on_layer_done(li): called once the launches that produce layer li's weight and
        bias gradient are enqueued (layers are finished last -> first: the data-parallel learner starts the all-reduce
        of a finished tail of the flat gradient while the earlier layers are still being back-propagated)."""
        ctx = self._ctx["train"]
        self._tls.role = "learner"
        inputs = ctx["inputs"]
        x0, stride0, idx0, off0, tT0 = ctx["first_in"]
        g = g_heads
        if self.tanh_scale > 0:  # d tanh(x/s)*s / dx = 1 - (y/s)^2 on the mean columns
            lib.tanh_scale_bwd(g_heads, ctx["acts"][-1], self.heads_ld, n, 1, self.num_action_params // 2, self.tanh_scale)
        chain = [li for li, L in enumerate(self.layers) if L.role != "rnn_hh"]
        mask0 = ctx.get("relu_mask0")  # sign bits recorded by THIS train forward (None: conv2's dgrad reads the activation)
        for pos in range(len(chain) - 1, -1, -1):
            li = chain[pos]
            L = self.layers[li]
            d = L.desc
            if L.role == "rnn_ih":
                g = self._rnn_sequence_bwd(li, g, n)  # dL/d(core_out) [n,H] -> dL/d(gx) time-major
            if pos == 0 and not self.features_in:
                d0 = lib.sf_conv_desc.from_buffer_copy(d)
                d0.traj_T = int(tT0)
                ws = self._workspace(lib.conv_wgrad_workspace(n, d0))
                if ctx.get("norm_tabs") is not None:  # the forward normalised in conv1's loader: so does the gradient
                    mu_, rstd_ = ctx["norm_tabs"]
                    lib.conv_wgrad_norm(x0, stride0, idx0, off0, mu_, rstd_, g, L.gw, L.gb, n, d0, ws)
                elif mask0 is not None:  # g is the gradient wrt conv1's ReLU output, unmasked: the kernel applies the bits
                    lib.conv_wgrad_relu_mask(x0, stride0, idx0, off0, g, mask0, L.gw, L.gb, n, d0, ws)
                else:
                    lib.conv_wgrad_raw(x0, stride0, idx0, off0, g, L.gw, L.gb, n, d0, ws)
            else:
                x = inputs[li]
                ws = self._workspace(lib.conv_wgrad_workspace(n, d))
                lib.conv_wgrad_raw(x, d.H * d.W * d.Cin, None, 0, g, L.gw, L.gb, n, d, ws)
                gin = self._buf(("g", li - 1), tuple(x.shape))
                dd = lib.sf_conv_desc.from_buffer_copy(d)
                dd.relu = L.in_act_kind  # derivative of the activation that produced x, fused into the epilogue
                if pos == 1 and mask0 is not None:  # conv1's sign bits are applied by its weight-gradient kernel instead
                    dd.relu = 0
                    lib.conv_dgrad(g, L.w, None, gin, n, dd)
                else:
                    lib.conv_dgrad(g, L.w, x if L.in_act_kind else None, gin, n, dd)
                g = gin
                if L.role == "rnn_ih":  # back to sample-major for the encoder
                    R, Cn = self._rnn_saved["R"], self._rnn_saved["Cn"]
                    gs = self._buf(("g", "x_sm", li), (n, L.K))
                    gs.view(Cn, R, L.K).copy_(g.view(R, Cn, L.K).transpose(0, 1))
                    g = gs
                if pos == 0:  # a trunk: d(loss) / d(pre-activation of the encoders' last layers), [n, F]
                    self.g_input = g
            if self.nonadaptive_std and li == len(self.layers) - 1:
                A = self.num_action_params  # the log-stddev columns have no weights in the reference: keep them at zero
                L.gw[:, 1 + A // 2:1 + A].zero_()
            if on_layer_done is not None:
                on_layer_done(li)

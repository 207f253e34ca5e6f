"""Fallback for USER-REGISTERED models (model_factory.py): a torch nn.Module behind the interface the native Learner /
rollout runner use (forward_heads / backward / flat_params / flat_grads / ...).

Only the network itself runs through torch (autograd on the GPU); the parameters are re-seated as views into ONE flat
fp32 buffer and the gradients into a second one, so the native gradient-norm / Adam / Lamb kernels, the data-parallel
all-reduce and the checkpoint code work unchanged.  The module must follow the reference's ActorCritic calling
convention (model/actor_critic.py:23-133):

    module(normalized_obs_dict, rnn_states, values_only=False) -> dict with "values" [n] and "action_logits" [n, A]

Sampling stays native (sf_sample_write_step), so "actions"/"log_prob_actions" in the returned dict are ignored.
Recurrent models run their core step by step (forward_head / forward_core / forward_tail, one-layer default GRU/LSTM);
async weight snapshots are not supported on this path (NotImplementedError).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch
from torch import nn

from sample_factory_amd.algo.utils.running_mean_std import RunningMeanStdInPlace
from sample_factory_amd.envs.spaces import calc_num_action_parameters
from sample_factory_amd.model.core import ModelCoreRNN
from sample_factory_amd.model.decoder import default_make_decoder_func
from sample_factory_amd.model.encoder import MultiInputEncoder, default_make_encoder_func


class TorchObsNormalizer:
    """utils/normalize.py:24-70 + running_mean_std.py:22-136 in torch (the native normaliser writes NHWC for the native
    conv stack; user modules expect the reference's NCHW float observations)."""

    def __init__(self, cfg, obs_shape, device, all_reduce=None, world: int = 1, key: str = "obs"):
        # obs_subtract_mean / obs_scale touch the "obs" key only (utils/normalize.py:59-65); the running statistics
        # exist per key, for the keys in cfg.normalize_input_keys (None: all) — running_mean_std.py:113-131
        self.key = key
        self.sub_mean = float(cfg.obs_subtract_mean) if (key == "obs" and abs(cfg.obs_subtract_mean) > 1e-5) else 0.0
        self.scale = float(cfg.obs_scale) if key == "obs" else 1.0
        keys = getattr(cfg, "normalize_input_keys", None)
        self.running = bool(cfg.normalize_input) and (not keys or key in keys)
        self.mean = torch.zeros(obs_shape, dtype=torch.float64, device=device)
        self.var = torch.ones(obs_shape, dtype=torch.float64, device=device)
        self.count = torch.ones(1, dtype=torch.float64, device=device)
        self._all_reduce, self.world = all_reduce, world

    def _scale(self, x: torch.Tensor) -> torch.Tensor:
        x = x.float()
        if self.sub_mean != 0.0:
            x = x - self.sub_mean
        if abs(self.scale - 1.0) > 1e-5:
            x = x * (1.0 / self.scale)
        return x

    def update(self, obs: torch.Tensor, stride: int, n: int, **_):
        """training-mode statistics update over the whole dataset (learner.py:957-961)"""
        if not self.running:
            return
        rows = obs.reshape((-1,) + tuple(self.mean.shape))[:n]
        s, ss = torch.zeros_like(self.mean), torch.zeros_like(self.mean)
        per = max(1, (64 << 20) // max(1, self.mean.numel() * 8))  # <= 64 MiB of float64 temporaries per chunk
        for i in range(0, n, per):  # image keys: the whole dataset as float64 would be tens of GB
            x = self._scale(rows[i:i + per]).double()
            s += x.sum(0)
            ss += (x * x).sum(0)
        if self._all_reduce is not None:
            self._all_reduce(s)
            self._all_reduce(ss)
        bn = float(n * self.world)
        bmean = s / bn
        bvar = (ss - s * bmean) / (bn - 1.0)  # unbiased, as torch.var
        delta = bmean - self.mean
        tot = self.count + bn
        m_a, m_b = self.var * self.count, bvar * bn
        self.mean = self.mean + delta * bn / tot
        self.var = (m_a + m_b + delta * delta * self.count * bn / tot) / tot
        self.count = tot

    def __call__(self, x: torch.Tensor, stats=None) -> torch.Tensor:
        """stats = (mean, var) of a published snapshot (async inference); None = the learner's current moments"""
        x = self._scale(x)
        if self.running:
            mean, var = stats if stats is not None else (self.mean, self.var)
            x = ((x - mean.float()) / torch.sqrt(var.float() + 1e-5)).clamp(-5.0, 5.0)
        return x

    def state_dict(self, prefix=None):
        prefix = prefix or f"obs_normalizer.running_mean_std.running_mean_std.{self.key}."
        if not self.running:
            return {}
        return {prefix + "running_mean": self.mean.cpu().clone(), prefix + "running_var": self.var.cpu().clone(),
                prefix + "count": self.count.cpu().clone()}

    def load_state_dict(self, sd, prefix=None):
        prefix = prefix or f"obs_normalizer.running_mean_std.running_mean_std.{self.key}."
        if self.running and prefix + "count" in sd:
            self.mean.copy_(torch.as_tensor(sd[prefix + "running_mean"], dtype=torch.float64))
            self.var.copy_(torch.as_tensor(sd[prefix + "running_var"], dtype=torch.float64))
            self.count.copy_(torch.as_tensor(sd[prefix + "count"], dtype=torch.float64).reshape(-1))


class _DictObsNormalizer:
    """what the Learner updates once per dataset when the observation is a dict of several keys"""

    def __init__(self, norms: Dict[str, "TorchObsNormalizer"]):
        self.norms = norms

    def update(self, obs, stride: int, n: int, **_):
        for k, nm in self.norms.items():
            nm.update(obs[k], 0, n)


def _make_decoder(cfg, size, factory):
    if factory.make_model_decoder_func is not None:
        return factory.make_model_decoder_func(cfg, size)
    return default_make_decoder_func(cfg, size)


def _make_core(cfg, size, factory):
    """a registered core factory decides; otherwise the reference's default: ModelCoreRNN iff cfg.use_rnn (None = no core)"""
    if factory.make_model_core_func is not None:
        return factory.make_model_core_func(cfg, size)
    return ModelCoreRNN(cfg, size) if cfg.use_rnn else None


def _initialize_weights(module: nn.Module, cfg) -> None:
    """model/actor_critic.py:73-96 (ActorCritic.initialize_weights applied to every layer, user-registered ones included):
    every `.bias` that is a Parameter is zeroed whatever the scheme (LayerNorm, user layers, ...; GRU / LSTM name theirs
    bias_ih_l<k> and keep torch's own initialisation); orthogonal / xavier_uniform touch the weights of exactly
    nn.Linear and nn.Conv2d (subclasses keep theirs, as `type(layer) is ...` in the reference)."""
    gain = cfg.policy_init_gain
    for m in module.modules():
        if isinstance(getattr(m, "bias", None), nn.Parameter):
            m.bias.data.fill_(0)
        if type(m) in (nn.Linear, nn.Conv2d):
            if cfg.policy_initialization == "orthogonal":
                nn.init.orthogonal_(m.weight.data, gain=gain)
            elif cfg.policy_initialization == "xavier_uniform":
                nn.init.xavier_uniform_(m.weight.data, gain=gain)


class _SeparateTorchActorCritic(nn.Module):
    """model/actor_critic.py:198-334 (ActorCriticSeparateWeights, cfg.actor_critic_share_weights=False): the actor and the
    critic own an encoder, a core and a decoder each.  forward_head concatenates the two encodings, the recurrent state is
    [actor state | critic state] (model_utils.py:20-22 doubles its width), forward_tail reads the value from the critic's
    half and the action parameters from the actor's.  Parameter paths equal the reference's (actor_encoder.*, actor_core.*,
    critic_encoder.*, critic_core.*, actor_decoder.mlp.*, critic_decoder.mlp.*, critic_linear.*,
    action_parameterization.distribution_linear.*)."""

    def __init__(self, cfg, obs_space, action_space, factory):
        super().__init__()

        def make_encoder():
            if factory.make_model_encoder_func is not None:
                return factory.make_model_encoder_func(cfg, obs_space)
            return default_make_encoder_func(cfg, obs_space)

        def make_core(size):
            return _make_core(cfg, size, factory)

        self.actor_encoder = make_encoder()
        self.actor_core = make_core(int(self.actor_encoder.get_out_size()))
        self.critic_encoder = make_encoder()
        self.critic_core = make_core(int(self.critic_encoder.get_out_size()))
        a_size = int(self.actor_core.get_out_size()) if self.actor_core is not None else int(self.actor_encoder.get_out_size())
        c_size = int(self.critic_core.get_out_size()) if self.critic_core is not None else int(self.critic_encoder.get_out_size())
        self.actor_decoder = _make_decoder(cfg, a_size, factory)
        self.critic_decoder = _make_decoder(cfg, c_size, factory)
        self.critic_linear = nn.Linear(int(self.critic_decoder.get_out_size()), 1)
        self.action_parameterization = nn.Module()
        # (the reference sizes the action head by the CRITIC decoder's width, actor_critic.py:222; the two are equal)
        self.action_parameterization.distribution_linear = nn.Linear(int(self.critic_decoder.get_out_size()),
                                                                     calc_num_action_parameters(action_space))
        _initialize_weights(self, cfg)

    def forward_head(self, normalized_obs_dict):
        return torch.cat([self.actor_encoder(normalized_obs_dict), self.critic_encoder(normalized_obs_dict)], dim=1)

    def forward_core(self, head_output, rnn_states):
        if self.actor_core is None:
            return head_output, rnn_states
        heads, states = head_output.chunk(2, dim=1), rnn_states.chunk(2, dim=1)
        a_out, a_new = self.actor_core(heads[0], states[0])
        c_out, c_new = self.critic_core(heads[1], states[1])
        return torch.cat([a_out, c_out], dim=1), torch.cat([a_new, c_new], dim=1)

    def forward_tail(self, core_output, values_only: bool = False):
        a_feat, c_feat = core_output.chunk(2, dim=1)
        res = dict(values=self.critic_linear(self.critic_decoder(c_feat)).squeeze(-1))
        if not values_only:
            res["action_logits"] = self.action_parameterization.distribution_linear(self.actor_decoder(a_feat))
        return res

    def forward(self, normalized_obs_dict, rnn_states=None, values_only: bool = False):
        x, new_rnn = self.forward_core(self.forward_head(normalized_obs_dict), rnn_states)
        res = self.forward_tail(x, values_only)
        res["new_rnn_states"] = new_rnn
        return res


class _DefaultTorchTail(nn.Module):
    """encoder -> [core] -> [decoder] -> critic_linear / distribution_linear, for a user-registered ENCODER (or core /
    decoder) with the remaining parts in their default form (model/actor_critic.py:136-195, decoder.py:15-31)."""

    def __init__(self, cfg, obs_space, action_space, factory, encoder=None):
        super().__init__()
        self.encoder = encoder if encoder is not None else (
            factory.make_model_encoder_func(cfg, obs_space) if factory.make_model_encoder_func else None)
        if self.encoder is None:
            raise NotImplementedError("register an encoder (or a whole actor-critic) when customising core/decoder")
        size = int(self.encoder.get_out_size())
        self.core = _make_core(cfg, size, factory)
        if self.core is not None:
            size = int(self.core.get_out_size())
        self.decoder = _make_decoder(cfg, size, factory)
        size = int(self.decoder.get_out_size())
        self.critic_linear = nn.Linear(size, 1)
        # same parameter path as the reference (model/action_parameterization.py:20-37)
        self.action_parameterization = nn.Module()
        self.action_parameterization.distribution_linear = nn.Linear(size, calc_num_action_parameters(action_space))
        _initialize_weights(self, cfg)

    # head / core / tail as in model/actor_critic.py:160-195 (the recurrent training pass runs the core step by step)
    def forward_head(self, normalized_obs_dict):
        return self.encoder(normalized_obs_dict)

    def forward_core(self, x, rnn_states):
        if self.core is None:
            return x, rnn_states
        return self.core(x, rnn_states)

    def forward_tail(self, x, values_only: bool = False):
        x = self.decoder(x)
        res = dict(values=self.critic_linear(x).squeeze(-1))
        if not values_only:
            res["action_logits"] = self.action_parameterization.distribution_linear(x)
        return res

    def forward(self, normalized_obs_dict, rnn_states=None, values_only: bool = False):
        x, new_rnn = self.forward_core(self.forward_head(normalized_obs_dict), rnn_states)
        res = self.forward_tail(x, values_only)
        res["new_rnn_states"] = new_rnn
        return res


def obs_keys_of(obs_space) -> List[str]:
    keys = list(obs_space.spaces.keys()) if hasattr(obs_space, "spaces") else ["obs"]
    return sorted(k for k in keys if k != "action_mask")


def build_torch_actor_critic(cfg, obs_space, action_space, factory) -> nn.Module:
    """the default actor-critic in torch around whatever the user registered; with no encoder registered (observation
    dicts of several keys and stacked recurrent layers land here) the reference's MultiInputEncoder (model/encoder.py:33-69:
    it is the default encoder for one key as well)"""
    if not cfg.actor_critic_share_weights:  # ActorCriticSeparateWeights (model/actor_critic.py:198-334)
        return _SeparateTorchActorCritic(cfg, obs_space, action_space, factory)
    enc = None
    if factory.make_model_encoder_func is None:
        enc = default_make_encoder_func(cfg, obs_space)
    return _DefaultTorchTail(cfg, obs_space, action_space, factory, encoder=enc)


class TorchPolicyAdapter:
    def __init__(self, cfg, obs_space, action_space, device, module: nn.Module, all_reduce=None):
        self.cfg, self.device = cfg, torch.device(device)
        if cfg.use_rnn and not all(hasattr(module, m) for m in ("forward_head", "forward_core", "forward_tail")):
            raise NotImplementedError("a recurrent user model must expose forward_head / forward_core / forward_tail "
                                      "(model/actor_critic.py:23-133)")
        self.module = module.to(self.device).float()
        self.obs_keys = obs_keys_of(obs_space)
        self.multi_key = len(self.obs_keys) > 1  # the Learner / rollout runner then pass {key: slab view} dicts
        self.obs_shapes = {k: tuple(obs_space[k].shape) for k in self.obs_keys} if hasattr(obs_space, "spaces") else \
            {"obs": tuple(obs_space.shape)}
        main = "obs" if "obs" in self.obs_shapes else self.obs_keys[0]
        self.obs_shape = self.obs_shapes[main]
        self.obs_elems = int(np.prod(self.obs_shape))
        self.num_action_params = int(calc_num_action_parameters(action_space))
        self.heads_ld = (1 + self.num_action_params + 3) // 4 * 4
        self.rnn_kind = (0 if cfg.rnn_type == "gru" else 1) if cfg.use_rnn else None  # the rollout runner's switch
        self._rnn_out: Dict = {}  # tag -> new state of the last one-step forward under that tag (sampler || learner threads)
        self.training = True
        # ---- re-seat parameters / gradients as views into flat buffers (16-byte aligned segments)
        params = [p for p in self.module.parameters() if p.requires_grad]
        self._names = [n for n, p in self.module.named_parameters() if p.requires_grad]
        offs, off = [], 0
        for p in params:
            offs.append(off)
            off += (p.numel() + 3) // 4 * 4
        self.num_flat = off
        self.flat_params = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.flat_grads = torch.zeros_like(self.flat_params)
        self._params, self._offs = params, offs
        with torch.no_grad():
            for p, o in zip(params, offs):
                self.flat_params[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = self.flat_params[o:o + p.numel()].view(p.shape)
                p.grad = self.flat_grads[o:o + p.numel()].view(p.shape)
        world = int(getattr(cfg, "dp_world", 1) or 1)
        self._norms = {k: TorchObsNormalizer(cfg, self.obs_shapes[k], self.device, all_reduce, world, key=k)
                       for k in self.obs_shapes}
        self._norm = self._norms[main]  # scale/shift always applies
        # what the Learner updates once per dataset
        if not cfg.normalize_input:
            self.obs_normalizer = None
        else:
            self.obs_normalizer = _DictObsNormalizer(self._norms) if self.multi_key else self._norm
        self.returns_normalizer: Optional[RunningMeanStdInPlace] = None
        if cfg.normalize_returns:
            self.returns_normalizer = RunningMeanStdInPlace((1,), self.device, all_reduce=all_reduce)
        self._bufs: Dict = {}
        self._train_heads = None
        self._snap = None

    def num_params(self) -> int:
        return sum(p.numel() for p in self._params)

    def ref_param_shapes(self):
        return [(n, tuple(p.shape)) for n, p in zip(self._names, self._params)]

    def train(self, mode=True):
        self.training = mode
        self.module.train(mode)
        if self.returns_normalizer is not None:
            self.returns_normalizer.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def params_changed(self) -> None:
        pass  # the module's parameters ARE views of flat_params

    def enable_weight_snapshots(self) -> None:
        """async_rl=True (the reference's default): inference reads PUBLISHED weights while the learner updates its own.
        Two snapshot slots, as on the native path (model/actor_critic.py): a slot is a deep copy of the module whose
        parameters are re-seated as views into one flat snapshot buffer; `publish_weights(slot)` is one device copy of
        the flat parameters (+ the module's buffers and the observation normaliser's moments); forwards tagged "inf*"
        run the module of slot `snap_read`."""
        import copy
        self._snap, self._snap_modules, self._snap_norms, self._snap_frozen = [], [], [], []
        self._frozen = [p for p in self.module.parameters() if not p.requires_grad]  # not in flat_params: copied on publish
        for _ in range(2):
            buf = self.flat_params.clone()
            m = copy.deepcopy(self.module)
            mine = [p for p in m.parameters() if p.requires_grad]
            self._snap_frozen.append([p for p in m.parameters() if not p.requires_grad])
            assert len(mine) == len(self._params) and len(self._snap_frozen[-1]) == len(self._frozen)
            for p, o in zip(mine, self._offs):
                p.grad = None
                p.requires_grad_(False)
                p.data = buf[o:o + p.numel()].view(p.shape)
            m.eval()
            self._snap.append(buf)
            self._snap_modules.append(m)
            self._snap_norms.append({k: (nm.mean, nm.var) for k, nm in self._norms.items()})
        self.snap_read = 0

    def publish_weights(self, slot: int) -> None:
        self._snap[slot].copy_(self.flat_params)
        with torch.no_grad():
            for dst, src in zip(self._snap_modules[slot].buffers(), self.module.buffers()):
                dst.copy_(src)
            for dst, src in zip(self._snap_frozen[slot], self._frozen):  # frozen (requires_grad=False) parameters a user
                dst.copy_(src)                                           # callback may still have changed
        # TorchObsNormalizer.update REBINDS mean / var to new tensors, so holding the current ones is a snapshot
        self._snap_norms[slot] = {k: (nm.mean, nm.var) for k, nm in self._norms.items()}

    def _buf(self, key, shape, dtype=torch.float32):
        t = self._bufs.get(key)
        if t is None or t.shape != torch.Size(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._bufs[key] = t
        return t

    def _zbuf(self, key, shape):
        t = self._bufs.get(key)
        if t is None or t.shape != torch.Size(shape):
            t = torch.zeros(shape, dtype=torch.float32, device=self.device)
            self._bufs[key] = t
        return t

    def tensor_segment_ids(self):
        seg = torch.full((self.num_flat,), 255, dtype=torch.uint8)
        for i, (p, o) in enumerate(zip(self._params, self._offs)):
            seg[o:o + p.numel()] = i
        if len(self._params) > 64:
            raise NotImplementedError("Lamb with more than 64 parameter tensors")
        return seg.to(self.device), len(self._params)

    # ---- gather the logical samples exactly as the native loaders address them
    def _gather(self, obs, n, index, offset, traj_T, shape=None):
        shape = self.obs_shape if shape is None else shape
        if traj_T:
            flat = obs.reshape((-1,) + shape)
            d = index.long() if index is not None else torch.arange(offset, offset + n, device=self.device)
            return flat[d + d // traj_T]                     # dataset index e*T+t -> slab row e*(T+1)+t
        x = obs.reshape((-1,) + shape) if obs.is_contiguous() else obs
        if index is not None:
            return x[index.long()]
        return x[offset:offset + n]

    def forward_heads(self, obs, n, *, sample_stride, index=None, offset=0, traj_T=0, tag="inf", rnn=None) -> List[torch.Tensor]:
        module, nstats = self.module, {}
        if self._snap is not None and tag.startswith("inf"):  # published snapshot (async mode)
            module, nstats = self._snap_modules[self.snap_read], self._snap_norms[self.snap_read]
        if isinstance(obs, dict):  # several observation keys: every key gathered / normalised on its own
            xd = {k: self._norms[k](self._gather(obs[k], n, index, offset, traj_T, self.obs_shapes[k]), nstats.get(k))
                  for k in self.obs_keys}
        else:
            main = self._norm.key
            xd = {"obs": self._norm(self._gather(obs, n, index, offset, traj_T), nstats.get(main))}
        train = tag == "train"
        # rollout forwards ("inf*") run in eval mode on BOTH paths, as the reference's inference worker does with its own
        # copy (inference_worker.py: actor_critic.eval()): the snapshot modules are eval() copies; in sync mode the learner's
        # module (kept in train mode, learner.py:228) is switched for the duration of the forward, so Dropout / BatchNorm
        # layers of a user encoder behave the same with async_rl on and off
        flip = tag.startswith("inf") and module is self.module and module.training
        if flip:
            module.eval()
        try:
            return self._forward_heads(module, xd, n, tag, rnn, train)
        finally:
            if flip:
                module.train()

    def _forward_heads(self, module, xd, n, tag, rnn, train) -> List[torch.Tensor]:
        with torch.set_grad_enabled(train):
            if rnn is None:
                res = module(xd, None, values_only=False)
            elif "R" not in rnn:  # one inference step on the stored state (rollout, bootstrap value)
                x, new_states = module.forward_core(module.forward_head(xd), rnn["states"])
                res = module.forward_tail(x, values_only=False)
                self._rnn_out[tag] = new_states.detach()
            else:
                # BPTT over recurrence-length chunks as a masked time loop: rows are chunk-major (Cn chunks x R steps),
                # the state is zeroed after a done / invalid step — the loop form of rnn_utils.py:114-158 that the
                # reference's tests/algo/test_rnn.py proves equal to its PackedSequence path
                R, Cn = rnn["R"], n // rnn["R"]
                feats = module.forward_head(xd).reshape(Cn, R, -1)
                h, keep, outs = rnn["h0"], rnn["keep_tm"], []
                for t in range(R):
                    out, h = module.forward_core(feats[:, t], h)
                    outs.append(out)
                    h = h * keep[t].unsqueeze(1)
                res = module.forward_tail(torch.stack(outs, 1).reshape(n, -1), values_only=False)
            heads = torch.cat([res["values"].reshape(n, 1), res["action_logits"].reshape(n, self.num_action_params),
                               torch.zeros((n, self.heads_ld - 1 - self.num_action_params), device=self.device)], dim=1)
        if train:
            self._train_heads = heads
        return [heads.detach()]

    def new_rnn_parts_of(self, tag: str = "inf"):
        """the native model hands (h, c) to sf_rnn_store_state; a torch module's state is one tensor -> None"""
        return None

    def new_rnn_states_of(self, tag: str = "inf") -> torch.Tensor:
        return self._rnn_out[tag]

    @property
    def new_rnn_states(self):
        return self._rnn_out.get("inf")

    def forward(self, normalized_obs_dict, rnn_states=None, values_only: bool = False, action_mask=None):
        if self.multi_key:
            obs = normalized_obs_dict
            nrows = next(iter(obs.values())).shape[0]
        else:
            obs = normalized_obs_dict["obs"] if isinstance(normalized_obs_dict, dict) else normalized_obs_dict
            nrows = obs.shape[0]
        rnn = dict(states=rnn_states) if (self.rnn_kind is not None and rnn_states is not None) else None
        heads = self.forward_heads(obs, nrows, sample_stride=self.obs_elems, rnn=rnn)[-1]
        res = dict(values=heads[:, 0], new_rnn_states=self.new_rnn_states if rnn is not None else rnn_states)
        if not values_only:
            res["action_logits"] = heads[:, 1:1 + self.num_action_params]
        return res

    def backward(self, acts, g_heads, obs, n, *, sample_stride, index=None, offset=0, traj_T=0) -> None:
        self.flat_grads.zero_()
        self._train_heads.backward(g_heads)   # accumulates into the p.grad views of flat_grads
        self._train_heads = None

    # ---- checkpoints in the module's own names (+ the reference's normaliser keys)
    def state_dict(self):
        sd = {k: v.detach().cpu().clone() for k, v in self.module.state_dict().items()}
        for nm in self._norms.values():
            sd.update(nm.state_dict())
        if self.returns_normalizer is not None:
            sd.update(self.returns_normalizer.state_dict("returns_normalizer."))
        return sd

    def load_state_dict(self, sd, strict=True):
        own = self.module.state_dict()
        with torch.no_grad():
            for k, v in own.items():
                if k in sd:
                    v.copy_(torch.as_tensor(sd[k]).to(v.dtype))
                elif strict:
                    raise KeyError(k)
        for nm in self._norms.values():
            nm.load_state_dict(sd)
        if self.returns_normalizer is not None and "returns_normalizer.running_mean" in sd:
            self.returns_normalizer.load_state_dict(sd, "returns_normalizer.")

    def flat_to_ref(self, flat: torch.Tensor):
        return {n: flat[o:o + p.numel()].view(p.shape).detach().cpu().clone()
                for n, p, o in zip(self._names, self._params, self._offs)}

"""Observation dicts with SEVERAL keys on the NATIVE kernels (reference: MultiInputEncoder inside ActorCriticSharedWeights,
sample_factory/model/encoder.py:33-69, model/actor_critic.py:136-196): one encoder per key in sorted key order — an MLP for a
1-D key, the conv stack + its fully connected layers for an image key — their outputs concatenated column-wise, then the
recurrent core, the decoder and the two heads.

Built from `ActorCritic` towers (model/actor_critic.py) seated on ONE flat parameter / gradient buffer
([encoder of key 0 | encoder of key 1 | ... | trunk]): one tower per key in its `part="encoder"` form and one `part="trunk"` tower
(core + decoder + fused heads on the concatenated feature batch), so that clip + Adam / Lamb, the gradient exchange, checkpoints
and the weight snapshots of async mode see a single parameter vector, as for the single-key model.  Forward: every encoder
tower runs its own kernels on its key's slab leaf (in place: u8 frames, index / offset / trajectory addressing are the
tower's), its [n, f_k] output is copied into columns [o_k, o_k + f_k) of the [n, F] feature batch (sf_copy_rows), the trunk
runs on that.  Backward: the trunk's first data gradient IS d(loss) / d(pre-activation of the encoders' outputs) (the
activation derivative is fused into its epilogue as between any two layers), its column blocks are the encoders' incoming
gradients.  Parameter names and order are the reference's (encoder.encoders.<key>.*, core.*, decoder.*, critic_linear.*,
action_parameterization.*); per-key normalisation rules are normalize.py:24-70's (mean shift / scale on the key "obs" only,
running statistics for the keys in cfg.normalize_input_keys)."""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch

from sample_factory_amd import lib
from sample_factory_amd.envs import spaces
from sample_factory_amd.model.actor_critic import ActorCritic


class _KeyedNormalizers:
    """what the Learner updates once per dataset (learner.py:957-961) when the observation is a dict of several keys"""

    def __init__(self, towers: Dict[str, ActorCritic]):
        self.towers = towers

    def update(self, obs, stride: int, n: int, **_):
        for k, t in self.towers.items():
            if t.obs_normalizer is not None:
                t.obs_normalizer.update(obs[k], t.obs_elems, n)


class MultiKeyActorCritic:
    multi_key = True  # the Learner / rollout runner pass {key: slab view} dicts

    def __init__(self, cfg, obs_space, action_space, device="cuda", all_reduce=None):
        self.cfg = cfg
        self.obs_keys = sorted(k for k in obs_space.spaces.keys() if k != "action_mask")
        if len(self.obs_keys) < 2:
            raise NotImplementedError("MultiKeyActorCritic: an observation dict with at least two keys")
        if not cfg.actor_critic_share_weights:
            raise NotImplementedError("separate actor / critic weights with several observation keys")
        self.encoders: Dict[str, ActorCritic] = {
            k: ActorCritic(cfg, obs_space, action_space, device, all_reduce=all_reduce, obs_key=k, part="encoder")
            for k in self.obs_keys}
        self.feat_of = {k: e.feat for k, e in self.encoders.items()}
        self.col0, F = {}, 0
        for k in self.obs_keys:
            self.col0[k] = F
            F += self.feat_of[k]
        self.F = F
        tspace = spaces.Dict({"obs": spaces.Box(-np.inf, np.inf, (F,), np.float32)})
        self.trunk = ActorCritic(cfg, tspace, action_space, device, all_reduce=all_reduce, part="trunk")
        t = self.trunk
        self.towers: List[ActorCritic] = [self.encoders[k] for k in self.obs_keys] + [t]
        self.device, self.obs_space, self.action_space = t.device, obs_space, action_space
        self.obs_shapes = {k: e.obs_shape for k, e in self.encoders.items()}
        main = "obs" if "obs" in self.encoders else self.obs_keys[0]
        self.obs_shape, self.obs_elems, self.obs_u8 = (self.encoders[main].obs_shape, self.encoders[main].obs_elems,
                                                       self.encoders[main].obs_u8)
        self.num_action_params, self.heads_ld = t.num_action_params, t.heads_ld
        self.rnn_kind, self.rnn_H, self.rnn_S, self.rnn_L, self.rnn_SL = t.rnn_kind, t.rnn_H, t.rnn_S, t.rnn_L, t.rnn_SL
        self.nonadaptive_std, self.tanh_scale = t.nonadaptive_std, t.tanh_scale
        self.training = True
        # ---- ONE flat buffer
        self._base, off = [], 0
        for tw in self.towers:
            self._base.append(off)
            off += tw.num_flat
        self.num_flat = off
        self.flat_params = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.flat_grads = torch.zeros_like(self.flat_params)
        self.flat_params_t = torch.zeros_like(self.flat_params)
        for tw, o in zip(self.towers, self._base):
            tw.seat_flat(self.flat_params[o:o + tw.num_flat], self.flat_grads[o:o + tw.num_flat],
                         self.flat_params_t[o:o + tw.num_flat])
        self.returns_normalizer = t.returns_normalizer
        self.obs_normalizer = _KeyedNormalizers(self.encoders) if any(
            e.obs_normalizer is not None for e in self.encoders.values()) else None
        self._snap = None

    # ------------------------------------------------------------------------------------------ reference surface
    def num_params(self) -> int:
        return sum(tw.num_params() for tw in self.towers)

    def ref_param_shapes(self):
        return [x for tw in self.towers for x in tw.ref_param_shapes()]

    def train(self, mode=True):
        self.training = mode
        for tw in self.towers:
            tw.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def model_to_device(self, device):
        assert torch.device(device).type == "cuda", "the native model only lives on the GPU"

    def normalize_obs(self, obs):
        return obs

    def state_dict(self) -> Dict[str, torch.Tensor]:
        parts = [tw.state_dict() for tw in self.towers]
        sd = {k: v for p in parts for k, v in p.items() if k.startswith(("obs_normalizer.", "returns_normalizer."))}
        for p in parts:  # the parameters in the reference's registration order
            sd.update({k: v for k, v in p.items() if not k.startswith(("obs_normalizer.", "returns_normalizer."))})
        return sd

    def load_state_dict(self, sd, strict=True):
        for tw in self.towers:
            if strict:
                for n, _ in tw.ref_param_shapes():
                    if n not in sd:
                        raise KeyError(n)
            tw.load_state_dict(sd, strict=strict)

    def flat_to_ref(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        out = {}
        for tw, o in zip(self.towers, self._base):
            out.update(tw.flat_to_ref(flat[o:o + tw.num_flat]))
        return out

    def tensor_segment_ids(self):
        """Lamb's per-tensor statistics (optimizers.py:108-135): the towers' maps side by side"""
        segs, base = [], 0
        for tw in self.towers:
            s, n = tw.tensor_segment_ids()
            segs.append(torch.where(s == 255, s, s + base))
            base += n
        if base > 64:
            raise NotImplementedError("Lamb with more than 64 parameter tensors")
        return torch.cat(segs), base

    # ---- the tower protocol of model/actor_critic_separate.py (this model as the actor's / the critic's tower)
    def seat_flat(self, flat_params: torch.Tensor, flat_grads: torch.Tensor, flat_params_t: torch.Tensor) -> None:
        for tw, o in zip(self.towers, self._base):  # (every tower carries its current values over)
            tw.seat_flat(flat_params[o:o + tw.num_flat], flat_grads[o:o + tw.num_flat], flat_params_t[o:o + tw.num_flat])
        self.flat_params, self.flat_grads, self.flat_params_t = flat_params, flat_grads, flat_params_t

    @property
    def heads_layer(self):
        return self.trunk.layers[-1]

    def share_normalizers_from(self, other) -> None:
        for k, e in self.encoders.items():
            e.obs_normalizer = other.encoders[k].obs_normalizer
        self.obs_normalizer = other.obs_normalizer
        self.trunk.returns_normalizer = self.returns_normalizer = None

    def share_seq_sync_from(self, other) -> None:
        self.trunk.share_seq_sync_from(other.trunk)

    def _seq_sync_buf(self):
        return self.trunk._seq_sync_buf()

    def share_snapshot_tables_from(self, other) -> None:
        for k, e in self.encoders.items():
            e._snap_tabs = other.encoders[k]._snap_tabs

    def normalizer_state(self) -> Dict[str, torch.Tensor]:
        return {k: v for e in self.encoders.values() for k, v in e.normalizer_state().items()}

    def load_normalizer_state(self, sd) -> None:
        for e in self.encoders.values():
            e.load_normalizer_state(sd)

    # ------------------------------------------------------------------------------------------ compute plumbing
    def params_changed(self) -> None:
        for tw in self.towers:
            tw.params_changed()

    def _buf(self, key, shape, dtype=torch.float32):
        return self.trunk._buf(("mk",) + tuple(key), shape, dtype)

    def _zbuf(self, key, shape):
        return self.trunk._zbuf(("mk",) + tuple(key), shape)

    def launch_key(self, tag: str = "inf"):
        keys = [tw.launch_key(tag) for tw in self.towers]
        return tuple(k[0] for k in keys), keys[0][1]

    @property
    def snap_read(self):
        return self.trunk.snap_read

    @snap_read.setter
    def snap_read(self, v):
        for tw in self.towers:
            tw.snap_read = v

    def enable_weight_snapshots(self) -> None:
        for tw in self.towers:
            tw.enable_weight_snapshots()
        self._snap = True

    def publish_weights(self, slot: int) -> None:
        for tw in self.towers:
            tw.publish_weights(slot)

    def rnn_abort_word(self):
        return self.trunk.rnn_abort_word()

    def rnn_abort_clear(self) -> None:
        self.trunk.rnn_abort_clear()

    def rnn_pass_aborted(self) -> bool:
        return self.trunk.rnn_pass_aborted()

    # ------------------------------------------------------------------------------------------ forward / backward
    def _stride_of(self, k, view, traj_T):
        """elements between two samples of key k: the learner addresses whole slab leaves [E, T + 1, ...] by dataset row
        (sf_common.h sample_base: dense frames), a rollout / bootstrap step hands one column view [B, ...] of the slab"""
        return self.encoders[k].obs_elems if (traj_T or view.is_contiguous()) else view.stride(0)

    def forward_heads(self, obs, n: int, *, sample_stride: int = 0, index=None, offset: int = 0, traj_T: int = 0, tag="inf",
                      rnn=None) -> List[torch.Tensor]:
        """obs: {key: view}; returns the trunk's layer outputs (last = heads [n, heads_ld])"""
        cat = self._buf((tag, "features"), (n, self.F))
        for k in self.obs_keys:
            e, v = self.encoders[k], obs[k]
            out = e.forward_heads(v, n, sample_stride=self._stride_of(k, v, traj_T), index=index, offset=offset,
                                  traj_T=traj_T, tag=tag)[-1]
            c = self.col0[k]
            if e.out_chw is None:
                lib.copy_rows(cat[:, c:c + e.feat], out.view(n, e.feat))
            else:  # conv output [n][pixel][channel] -> the reference's [channel][pixel] feature order (a torch copy kernel)
                C, HW = e.out_chw[0], e.out_chw[1] * e.out_chw[2]
                lib.recording_unsafe("feature re-ordering of a conv-last encoder is a torch op")
                cat[:, c:c + e.feat].unflatten(1, (C, HW)).copy_(out.view(n, HW, C).transpose(1, 2))
        return self.trunk.forward_heads(cat, n, sample_stride=self.F, tag=tag, rnn=rnn)

    def backward(self, acts, g_heads: torch.Tensor, obs, n: int, *, sample_stride: int = 0, index=None, offset: int = 0,
                 traj_T: int = 0, on_layer_done=None) -> None:
        t = self.trunk
        t.backward(acts, g_heads, None, n, sample_stride=self.F)
        gin = t.g_input  # [n, F]
        for k in self.obs_keys:
            e, v, c = self.encoders[k], obs[k], self.col0[k]
            g = e._buf(("g", "out"), (n, e.feat))
            if e.out_chw is None:
                lib.copy_rows(g, gin[:, c:c + e.feat])
            else:
                C, HW = e.out_chw[0], e.out_chw[1] * e.out_chw[2]
                g.view(n, HW, C).copy_(gin[:, c:c + e.feat].unflatten(1, (C, HW)).transpose(1, 2))
            e.backward(None, g, v, n, sample_stride=self._stride_of(k, v, traj_T), index=index, offset=offset, traj_T=traj_T)

    def new_rnn_parts_of(self, tag: str = "inf"):
        return self.trunk.new_rnn_parts_of(tag)

    def new_rnn_states_of(self, tag: str = "inf") -> torch.Tensor:
        return self.trunk.new_rnn_states_of(tag)

    @property
    def new_rnn_states(self) -> torch.Tensor:
        return self.new_rnn_states_of("inf")

    def forward(self, normalized_obs_dict, rnn_states=None, values_only: bool = False, action_mask=None):
        """Inference-style forward on dense per-key batches {key: [B, ...]} (the reference's ActorCritic.forward surface)"""
        obs = {k: normalized_obs_dict[k].contiguous() for k in self.obs_keys}
        B = obs[self.obs_keys[0]].shape[0]
        rnn = dict(states=rnn_states) if self.rnn_kind is not None else None
        heads = self.forward_heads(obs, B, rnn=rnn)[-1]
        res = dict(values=heads[:, 0])
        if not values_only:
            res["action_logits"] = heads[:, 1:1 + self.num_action_params]
        res["new_rnn_states"] = self.new_rnn_states if self.rnn_kind is not None else rnn_states
        return res

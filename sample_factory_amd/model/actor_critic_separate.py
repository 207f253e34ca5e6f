"""cfg.actor_critic_share_weights=False on the NATIVE kernels (reference: ActorCriticSeparateWeights,
sample_factory/model/actor_critic.py:198-334): the actor and the critic own an encoder, a recurrent core and a decoder each;
`critic_linear` reads the critic's features, `action_parameterization.distribution_linear` the actor's; the recurrent state of a
sample is [actor state | critic state] (model_utils.py:20-22 doubles its width).

Built from two `ActorCritic` towers (model/actor_critic.py) seated on ONE flat parameter / gradient buffer ([actor | critic]), so
that clip + Adam / Lamb, the gradient exchange, checkpoints and the weight snapshots of async mode see a single parameter vector,
exactly as for the shared-weights model.  Every tower keeps the fused heads GEMM [features, 1 + A]; the columns a tower does not
own (the value column in the actor, the action columns in the critic) have zero weights, zero bias and — because each tower's
backward pass receives the loss gradient with the other tower's columns zeroed — zero gradients: they never move.  The network
kernels, the fused sequence passes and the rollout's one-step path are the towers' own; this class only splits states / gradients
and merges heads.  Parameter names and order are the reference's (actor_encoder.*, actor_core.*, critic_encoder.*, critic_core.*,
actor_decoder.*, critic_decoder.*, critic_linear.*, action_parameterization.*)."""
from __future__ import annotations

import copy
from typing import Dict, List, Optional

import torch

from sample_factory_amd import lib
from sample_factory_amd.model.actor_critic import ActorCritic, get_rnn_size

_PARTS = ("encoder.", "core.", "decoder.")


class SeparateActorCritic:
    def __init__(self, cfg, obs_space, action_space, device="cuda", all_reduce=None):
        self.cfg = cfg
        tcfg = copy.copy(cfg)
        tcfg.actor_critic_share_weights = True  # a tower is the shared-weights architecture
        keys = [k for k in obs_space.spaces.keys() if k != "action_mask"]
        if len(keys) > 1:  # observation dicts with several keys: each tower is the multi-key composite (encoder towers + trunk)
            from sample_factory_amd.model.actor_critic_multikey import MultiKeyActorCritic as Tower
        else:
            Tower = ActorCritic
        self.actor = Tower(tcfg, obs_space, action_space, device, all_reduce=all_reduce)
        self.critic = Tower(tcfg, obs_space, action_space, device, all_reduce=all_reduce)
        self.towers = (self.actor, self.critic)
        a, c = self.towers
        self.multi_key, self.obs_keys = bool(getattr(a, "multi_key", False)), list(getattr(a, "obs_keys", ["obs"]))
        self.device, self.obs_space, self.action_space = a.device, obs_space, action_space
        self.obs_shape, self.obs_elems, self.obs_u8 = a.obs_shape, a.obs_elems, a.obs_u8
        self.num_action_params, self.heads_ld = a.num_action_params, a.heads_ld
        self.rnn_kind, self.rnn_H = a.rnn_kind, a.rnn_H
        self.rnn_S = get_rnn_size(cfg)
        assert self.rnn_S == a.rnn_S + c.rnn_S
        self.training = True
        # ---- ONE flat buffer [actor | critic]
        na = a.num_flat
        self.num_flat = a.num_flat + c.num_flat
        self.flat_params = torch.zeros(self.num_flat, dtype=torch.float32, device=self.device)
        self.flat_grads = torch.zeros_like(self.flat_params)
        self.flat_params_t = torch.zeros_like(self.flat_params)
        a.seat_flat(self.flat_params[:na], self.flat_grads[:na], self.flat_params_t[:na])
        c.seat_flat(self.flat_params[na:], self.flat_grads[na:], self.flat_params_t[na:])
        # ---- shared: input / return normalisers (actor_critic.py:44-61: they belong to the model, not to a tower), the
        # fused sequence passes' sync words (one sticky abort word for the optimiser's skip flag)
        c.share_normalizers_from(a)
        self.obs_normalizer, self.returns_normalizer = a.obs_normalizer, a.returns_normalizer
        if self.rnn_kind is not None:
            c.share_seq_sync_from(a)
        self._snap = None
        self._zero_foreign_columns()

    # ------------------------------------------------------------------------------------------ heads bookkeeping
    def _zero_foreign_columns(self) -> None:
        """the value column of the actor's heads and the action columns of the critic's: not parameters of the model"""
        with torch.no_grad():
            Ha, Hc = self.actor.heads_layer, self.critic.heads_layer
            Ha.w[:, 0].zero_()
            Ha.b[0].zero_()
            Hc.w[:, 1:].zero_()
            Hc.b[1:].zero_()
        self.params_changed()

    # ------------------------------------------------------------------------------------------ reference surface
    def num_params(self) -> int:
        return sum(int(torch.Size(s).numel()) for _, s in self.ref_param_shapes())

    @staticmethod
    def _rename(name: str, who: str) -> Optional[str]:
        for p in _PARTS:
            if name.startswith(p):
                return f"{who}_{name}"
        if name.startswith("critic_linear."):
            return name if who == "critic" else None
        if name.startswith("action_parameterization."):
            return name if who == "actor" else None
        return None  # normaliser statistics: handled once, by the model

    def ref_param_shapes(self):
        """(name, shape) in the reference's registration order (actor_critic.py:206-226): actor encoder, actor core, critic
        encoder, critic core, actor decoder, critic decoder, critic_linear, action_parameterization"""
        by = {}
        for who, t in (("actor", self.actor), ("critic", self.critic)):
            for n, shp in t.ref_param_shapes():
                r = self._rename(n, who)
                if r is not None:
                    by.setdefault((who, n.split(".")[0]), []).append((r, shp))
        order = [("actor", "encoder"), ("actor", "core"), ("critic", "encoder"), ("critic", "core"), ("actor", "decoder"),
                 ("critic", "decoder"), ("critic", "critic_linear"), ("actor", "action_parameterization")]
        return [x for k in order for x in by.get(k, [])]

    def train(self, mode=True):
        self.training = mode
        for t in self.towers:
            t.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def model_to_device(self, device):
        assert torch.device(device).type == "cuda", "the native model only lives on the GPU"

    def normalize_obs(self, obs):
        return obs

    def _split_sd(self, sd: Dict, who: str) -> Dict:
        out = {}
        for k, v in sd.items():
            r = self._rename(k, who)
            if r is not None:
                out[r] = v
        return out

    def state_dict(self) -> Dict[str, torch.Tensor]:
        sd = dict(self.actor.normalizer_state())
        if self.returns_normalizer is not None:
            sd.update(self.returns_normalizer.state_dict("returns_normalizer."))
        parts = {who: self._split_sd(t.state_dict(), who) for who, t in (("actor", self.actor), ("critic", self.critic))}
        for name, _ in self.ref_param_shapes():
            sd[name] = parts["actor" if name in parts["actor"] else "critic"][name]
        return sd

    def load_state_dict(self, sd, strict=True):
        for who, t in (("actor", self.actor), ("critic", self.critic)):
            own = t.state_dict()  # the tower's own names; foreign heads entries keep their (zero) values
            for k in list(own):
                r = self._rename(k, who)
                if r is not None:
                    if r in sd:
                        own[k] = sd[r]
                    elif strict:
                        raise KeyError(r)
            for k in [k_ for k_ in own if k_.startswith(("obs_normalizer.", "returns_normalizer."))]:
                del own[k]
            t.load_state_dict(own, strict=False)
        self._zero_foreign_columns()
        self.actor.load_normalizer_state(sd)
        if self.returns_normalizer is not None and "returns_normalizer.running_mean" in sd:
            self.returns_normalizer.load_state_dict(sd, "returns_normalizer.")
        elif strict and self.returns_normalizer is not None:
            raise KeyError("returns_normalizer.* missing from state dict")

    def flat_to_ref(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        na = self.actor.num_flat
        out = self._split_sd(self.actor.flat_to_ref(flat[:na]), "actor")
        out.update(self._split_sd(self.critic.flat_to_ref(flat[na:]), "critic"))
        return {n: out[n] for n, _ in self.ref_param_shapes()}

    def tensor_segment_ids(self):
        """Lamb's per-tensor statistics (optimizers.py:108-135): the towers' maps side by side; the heads tensors a tower does
        not own (critic_linear in the actor, the action parameterisation in the critic) count as padding"""
        sa, na = self.actor.tensor_segment_ids()
        sc, nc = self.critic.tensor_segment_ids()
        sa, sc = sa.clone(), sc.clone()
        extra = 1 if self.actor.nonadaptive_std else 0
        ha, hc = na - 4 - extra, nc - 4 - extra  # first heads id in a tower's map: critic w, critic b, dist w, dist b [, stddev]
        sa[(sa == ha) | (sa == ha + 1)] = 255
        sc[(sc >= hc + 2) & (sc != 255)] = 255
        sc = torch.where(sc == 255, sc, sc + na)
        if na + nc > 64:
            raise NotImplementedError("Lamb with more than 64 parameter tensors")
        return torch.cat([sa, sc]), na + nc

    # ------------------------------------------------------------------------------------------ compute plumbing
    def params_changed(self) -> None:
        for t in self.towers:
            t.params_changed()

    def _buf(self, key, shape, dtype=torch.float32):
        return self.actor._buf(("sep",) + tuple(key), shape, dtype)

    def _zbuf(self, key, shape):
        return self.actor._zbuf(("sep",) + tuple(key), shape)

    def launch_key(self, tag: str = "inf"):
        ka, kc = self.actor.launch_key(tag), self.critic.launch_key(tag)
        return (ka[0], kc[0]), ka[1]

    @property
    def snap_read(self):
        return self.actor.snap_read

    @snap_read.setter
    def snap_read(self, v):
        for t in self.towers:
            t.snap_read = v

    def enable_weight_snapshots(self) -> None:
        for t in self.towers:
            t.enable_weight_snapshots()
        self.critic.share_snapshot_tables_from(self.actor)  # one normaliser (per key), one pair of published tables
        self._snap = True

    def publish_weights(self, slot: int) -> None:
        # (the shared normalisation tables are copied by both calls: the same few KB twice, stream-ordered)
        self.actor.publish_weights(slot)
        self.critic.publish_weights(slot)

    def rnn_abort_word(self):
        return self.actor.rnn_abort_word()

    def rnn_abort_clear(self) -> None:
        self.actor.rnn_abort_clear()

    def rnn_pass_aborted(self) -> bool:
        return self.actor.rnn_pass_aborted()

    # ------------------------------------------------------------------------------------------ forward / backward
    def _rnn_of(self, rnn, which: int):
        """a tower's half of the recurrent inputs: state rows / chunk-start states are [actor | critic] column-wise"""
        if rnn is None:
            return None
        half = self.actor.rnn_S
        sl = slice(which * half, (which + 1) * half)
        out = dict(rnn)
        if "states" in rnn:
            out["states"] = rnn["states"][:, sl]
        if "h0" in rnn:
            out["h0"] = rnn["h0"][:, sl]
        return out

    def forward_heads(self, obs, n: int, *, sample_stride: int, index=None, offset: int = 0, traj_T: int = 0, tag="inf",
                      rnn=None) -> List[torch.Tensor]:
        """both towers on the same observations; returns [heads [n, heads_ld]]: column 0 from the critic, the action
        parameters from the actor"""
        ha = self.actor.forward_heads(obs, n, sample_stride=sample_stride, index=index, offset=offset, traj_T=traj_T, tag=tag,
                                      rnn=self._rnn_of(rnn if self.rnn_kind is not None else None, 0))[-1]
        hc = self.critic.forward_heads(obs, n, sample_stride=sample_stride, index=index, offset=offset, traj_T=traj_T, tag=tag,
                                       rnn=self._rnn_of(rnn if self.rnn_kind is not None else None, 1))[-1]
        out = self._buf((tag, "heads"), (n, self.heads_ld))
        lib.copy_rows(out, ha)                    # (library launches, not torch ops: a rollout step of this model is
        lib.copy_rows(out[:, 0:1], hc[:, 0:1])    #  recordable as a launch program, lib.LaunchProgram)
        return [out]

    def backward(self, acts, g_heads: torch.Tensor, obs, n: int, *, sample_stride: int, index=None, offset: int = 0,
                 traj_T: int = 0, on_layer_done=None) -> None:
        """d(loss)/d(heads) split by owner: the actor tower sees the action columns, the critic tower the value column"""
        ga = self._buf(("g", "heads_actor"), (n, self.heads_ld))
        gc = self._zbuf(("g", "heads_critic"), (n, self.heads_ld))
        ga.copy_(g_heads)
        ga[:, 0].zero_()
        gc[:, 0].copy_(g_heads[:, 0])
        self.actor.backward(None, ga, obs, n, sample_stride=sample_stride, index=index, offset=offset, traj_T=traj_T)
        self.critic.backward(None, gc, obs, n, sample_stride=sample_stride, index=index, offset=offset, traj_T=traj_T)

    def new_rnn_parts_of(self, tag: str = "inf"):
        pa, pc = self.actor.new_rnn_parts_of(tag), self.critic.new_rnn_parts_of(tag)
        return None if pa is None or pc is None else list(pa) + list(pc)

    def new_rnn_states_of(self, tag: str = "inf") -> torch.Tensor:
        return torch.cat([self.actor.new_rnn_states_of(tag), self.critic.new_rnn_states_of(tag)], dim=1)

    @property
    def new_rnn_states(self) -> torch.Tensor:
        return self.new_rnn_states_of("inf")

    def forward(self, normalized_obs_dict, rnn_states=None, values_only: bool = False, action_mask=None):
        if self.multi_key:
            obs = {k: normalized_obs_dict[k].contiguous() for k in self.obs_keys}
            B, stride = obs[self.obs_keys[0]].shape[0], 0
        else:
            obs = normalized_obs_dict["obs"] if isinstance(normalized_obs_dict, dict) else normalized_obs_dict
            B, stride = obs.shape[0], (self.obs_elems if obs.is_contiguous() else obs.stride(0))
        rnn = dict(states=rnn_states) if self.rnn_kind is not None else None
        heads = self.forward_heads(obs, B, sample_stride=stride, rnn=rnn)[-1]
        res = dict(values=heads[:, 0])
        if not values_only:
            res["action_logits"] = heads[:, 1:1 + self.num_action_params]
        res["new_rnn_states"] = self.new_rnn_states if self.rnn_kind is not None else rnn_states
        return res

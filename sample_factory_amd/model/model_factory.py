"""Model plugin surface — sample_factory/model/model_factory.py:16-60 + algo/utils/context.py (global_model_factory).

    global_model_factory().register_actor_critic_factory(make_actor_critic_func)   # (cfg, obs_space, action_space) -> nn.Module
    global_model_factory().register_encoder_factory(make_encoder_func)             # (cfg, obs_space) -> Encoder module
    global_model_factory().register_model_core_factory / register_decoder_factory  # (cfg, in_size) -> module

The DEFAULT models (Nature-CNN / MLP encoders, GRU/LSTM core, MLP decoder) run on the native HIP path
(`model/actor_critic.py`).  A user-registered torch module keeps working: `create_actor_critic` wraps it in
`TorchPolicyAdapter` (model/torch_policy.py), which drives the user's module through torch autograd on the GPU while
everything around the network — rollout sampling, slab protocol, GAE, returns normaliser, PPO loss forward/backward,
gradient clipping, Adam/Lamb on one flat buffer, data-parallel all-reduce — stays on the native kernels (SURVEY.md §8b:
"native fast-path only when the factory is the default, otherwise fall back to autograd through the user module").
"""
from __future__ import annotations

from typing import Callable, Optional


class ModelFactory:
    def __init__(self):
        self.make_actor_critic_func: Optional[Callable] = None   # None = the native default
        self.make_model_encoder_func: Optional[Callable] = None
        self.make_model_core_func: Optional[Callable] = None
        self.make_model_decoder_func: Optional[Callable] = None

    def register_actor_critic_factory(self, make_actor_critic_func: Callable):
        """Override the default actor-critic with a custom model: f(cfg, obs_space, action_space) -> nn.Module"""
        self.make_actor_critic_func = make_actor_critic_func

    def register_encoder_factory(self, make_model_encoder_func: Callable):
        """observations -> ENCODER -> core -> decoder -> heads: f(cfg, obs_space) -> module with get_out_size()"""
        self.make_model_encoder_func = make_model_encoder_func

    def register_model_core_factory(self, make_model_core_func: Callable):
        self.make_model_core_func = make_model_core_func

    def register_decoder_factory(self, make_model_decoder_func: Callable):
        self.make_model_decoder_func = make_model_decoder_func

    def is_default(self) -> bool:
        return (self.make_actor_critic_func is None and self.make_model_encoder_func is None and
                self.make_model_core_func is None and self.make_model_decoder_func is None)

    def reset(self):
        self.__init__()


_FACTORY = ModelFactory()


def global_model_factory() -> ModelFactory:
    return _FACTORY


def create_actor_critic(cfg, obs_space, action_space, device, all_reduce=None):
    """model/actor_critic.py:337-342 create_actor_critic: the native model unless the user registered something"""
    f = global_model_factory()
    from sample_factory_amd.model.torch_policy import TorchPolicyAdapter, build_torch_actor_critic, obs_keys_of
    separate = not bool(cfg.actor_critic_share_weights)
    # (stacked recurrent layers, cfg.rnn_num_layers > 1, run on the native model since round 6: SF_NATIVE_STACKED_RNN=0 sends
    # them back to the torch path)
    import os
    stacked_rnn = bool(cfg.use_rnn) and int(cfg.rnn_num_layers) > 1 and os.environ.get("SF_NATIVE_STACKED_RNN", "1") == "0"
    if f.is_default() and len(obs_keys_of(obs_space)) <= 1 and not stacked_rnn and not separate:
        from sample_factory_amd.model.actor_critic import ActorCritic
        return ActorCritic(cfg, obs_space, action_space, device, all_reduce=all_reduce)
    import torch
    multi = len(obs_keys_of(obs_space)) > 1
    if (f.is_default() and not stacked_rnn and separate and torch.device(device).type == "cuda"
            and os.environ.get("SF_NATIVE_SEPARATE_WEIGHTS", "1") != "0"
            and (not multi or os.environ.get("SF_NATIVE_MULTIKEY", "1") != "0")):
        # cfg.actor_critic_share_weights=False (ActorCriticSeparateWeights, model/actor_critic.py:198-334) on the native
        # kernels since round 6: two towers on one flat parameter buffer (model/actor_critic_separate.py)
        from sample_factory_amd.model.actor_critic_separate import SeparateActorCritic
        return SeparateActorCritic(cfg, obs_space, action_space, device, all_reduce=all_reduce)
    if (f.is_default() and len(obs_keys_of(obs_space)) > 1 and not stacked_rnn and not separate
            and torch.device(device).type == "cuda" and os.environ.get("SF_NATIVE_MULTIKEY", "1") != "0"):
        # observation dicts of several keys (model/encoder.py:33-69, MultiInputEncoder: one encoder per key, concatenated) on
        # the native kernels since round 6: one encoder tower per key + a trunk on one flat parameter buffer
        # (model/actor_critic_multikey.py).  A shape the towers refuse keeps the torch path below.
        from sample_factory_amd.model.actor_critic_multikey import MultiKeyActorCritic
        try:
            return MultiKeyActorCritic(cfg, obs_space, action_space, device, all_reduce=all_reduce)
        except NotImplementedError as e:
            from sample_factory_amd.utils.utils import log
            log.warning("multi-key observations: torch path (%s)", e)
    # cfg.actor_critic_share_weights=False with several keys, CPU devices, user-registered parts:
    # the default architecture in torch: same fallback as a user-registered model, everything around the network
    # stays native
    if f.make_actor_critic_func is not None:
        module = f.make_actor_critic_func(cfg, obs_space, action_space)
    else:
        module = build_torch_actor_critic(cfg, obs_space, action_space, f)
    return TorchPolicyAdapter(cfg, obs_space, action_space, device, module, all_reduce=all_reduce)

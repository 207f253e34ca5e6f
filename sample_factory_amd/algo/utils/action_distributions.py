"""Action distributions as OBJECTS — the API surface of sample_factory/algo/utils/action_distributions.py:14-323
(`get_action_distribution`, `CategoricalActionDistribution`, `TupleActionDistribution`, `ContinuousActionDistribution`,
`sample_actions_log_probs`, `argmax_actions`) for code written against the reference: custom models, evaluation scripts,
the distribution object `Learner._calculate_losses` returns.

The training loop never goes through these objects: sampling is fused into sf_sample_write_step*, log-prob / entropy / KL
/ symmetric-KL and their gradients into sf_ppo_loss (csrc/sf_rl.hip), both pinned against the reference's known answers
(tests/golden/action_dist.npz).  The classes below state the same mathematics as plain tensor expressions on whatever
device the logits live on, so that values read off a returned distribution agree with what the kernels computed.

Formulas (reference lines):
  categorical      p = softmax(z), log p = log_softmax(z); masked: z + (mask == 0) * -1e9, p renormalised over the mask
                   (:84-96); H = -sum p log p (:150-152); KL(self || other) = sum p (log p - log q) (:154-158, :179-180);
                   symmetric KL with the uniform prior u = 1/A: (KL(p || u) + KL(u || p)) / 2 (:168-177)
  tuple of heads   independent categoricals over consecutive logit slices: log-probs, entropies and KLs add (:232-282)
  diagonal normal  params = [mean | log_std], std = clamp(exp(log_std), 1e-4, 1e4) (:294-306); event log-prob / entropy
                   summed over the action dimensions; KL(self || other) closed form, summed (:308-310)
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence

import torch
import torch.nn.functional as F

from sample_factory_amd.envs.spaces import (action_head_sizes, calc_num_action_parameters, calc_num_actions, is_box,
                                             is_discrete)

__all__ = ["calc_num_actions", "calc_num_action_parameters", "is_continuous_action_space", "get_action_distribution",
           "sample_actions_log_probs", "argmax_actions", "masked_softmax", "masked_log_softmax",
           "CategoricalActionDistribution", "TupleActionDistribution", "ContinuousActionDistribution"]

_MASKED_OUT = -1e9
_STD_MIN, _STD_MAX = 1e-4, 1e4


def is_continuous_action_space(action_space) -> bool:
    return is_box(action_space)


def _push_down(logits: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    return logits + (mask == 0).to(logits.dtype) * _MASKED_OUT


def masked_softmax(logits: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    p = F.softmax(_push_down(logits, mask), dim=-1) * mask
    return p / (p.sum(dim=-1, keepdim=True) + 1e-13)


def masked_log_softmax(logits: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    return F.log_softmax(_push_down(logits, mask), dim=-1)


class CategoricalActionDistribution:
    """one Discrete(A) head over `raw_logits` [..., A]; `action_mask` [..., A] (1 = allowed) is optional"""

    def __init__(self, raw_logits: torch.Tensor, action_mask: Optional[torch.Tensor] = None):
        self.raw_logits = raw_logits
        self.action_mask = action_mask
        self._p = self._logp = None

    @property
    def probs(self) -> torch.Tensor:
        if self._p is None:
            self._p = F.softmax(self.raw_logits, dim=-1) if self.action_mask is None else \
                masked_softmax(self.raw_logits, self.action_mask)
        return self._p

    @property
    def log_probs(self) -> torch.Tensor:
        if self._logp is None:
            self._logp = F.log_softmax(self.raw_logits, dim=-1) if self.action_mask is None else \
                masked_log_softmax(self.raw_logits, self.action_mask)
        return self._logp

    def sample(self) -> torch.Tensor:
        """[..., 1] int64 indices, one draw per row (rows whose every action is masked out draw uniformly)"""
        p = self.probs
        if self.action_mask is not None:
            dead = p.sum(dim=-1, keepdim=True) == 0
            p = torch.where(dead, torch.full_like(p, 1e-6), p)
        flat = p.reshape(-1, p.shape[-1])
        return torch.multinomial(flat, 1, replacement=True).reshape(p.shape[:-1] + (1,))

    def sample_gumbel(self) -> torch.Tensor:
        """arg-max of logits + Gumbel noise: a draw from the same distribution without normalising it"""
        noise = -torch.empty_like(self.raw_logits).exponential_().log_()
        noisy = self.raw_logits + noise
        if self.action_mask is not None:
            noisy = noisy * self.action_mask
        return noisy.argmax(dim=-1)

    def log_prob(self, value: torch.Tensor) -> torch.Tensor:
        idx = value.long().reshape(self.log_probs.shape[:-1] + (1,))
        return self.log_probs.gather(-1, idx).reshape(-1)

    def entropy(self) -> torch.Tensor:
        return -(self.probs * self.log_probs).sum(dim=-1)

    def _kl(self, other_log_probs: torch.Tensor) -> torch.Tensor:          # KL(self || other)
        return (self.probs * (self.log_probs - other_log_probs)).sum(dim=-1)

    def _kl_inverse(self, other_log_probs: torch.Tensor) -> torch.Tensor:  # KL(other || self)
        return (other_log_probs.exp() * (other_log_probs - self.log_probs)).sum(dim=-1)

    def _kl_symmetric(self, other_log_probs: torch.Tensor) -> torch.Tensor:
        return 0.5 * (self._kl(other_log_probs) + self._kl_inverse(other_log_probs))

    def symmetric_kl_with_uniform_prior(self) -> torch.Tensor:
        A = self.log_probs.shape[-1]
        log_u = -math.log(A)
        to_uniform = (self.probs * (self.log_probs - log_u)).sum(dim=-1)
        from_uniform = ((log_u - self.log_probs) / A).sum(dim=-1)
        return 0.5 * (to_uniform + from_uniform)

    def kl_divergence(self, other: "CategoricalActionDistribution") -> torch.Tensor:
        return self._kl(other.log_probs)

    def dbg_print(self) -> str:
        s = (f"entropy={float(self.entropy().mean()):.3f} min_logit={float(self.raw_logits.min()):.3f} "
             f"max_logit={float(self.raw_logits.max()):.3f} min_prob={float(self.probs.min()):.3f} "
             f"max_prob={float(self.probs.max()):.3f}")
        return s


class TupleActionDistribution:
    """independent heads over consecutive slices of `logits_flat` [..., sum(params per head)]; actions [..., sum(actions
    per head)] hold one column per Discrete head (or D columns per Box(D) head)"""

    def __init__(self, action_space, logits_flat: torch.Tensor, action_mask: Optional[torch.Tensor] = None):
        self.action_space = action_space
        spaces = list(action_space.spaces if hasattr(action_space, "spaces") else action_space)
        self._param_sizes = [calc_num_action_parameters(s) for s in spaces]
        self._action_sizes = [calc_num_actions(s) for s in spaces]
        assert sum(self._param_sizes) == logits_flat.shape[-1], (self._param_sizes, tuple(logits_flat.shape))
        masks: Sequence[Optional[torch.Tensor]] = [None] * len(spaces)
        if action_mask is not None:
            masks = torch.split(action_mask, self._param_sizes, dim=-1)
        self.distributions: List = [get_action_distribution(s, z, m) for s, z, m in
                                    zip(spaces, torch.split(logits_flat, self._param_sizes, dim=-1), masks)]

    def _per_head(self, actions: torch.Tensor):
        return torch.split(actions, self._action_sizes, dim=-1)

    def sample(self) -> torch.Tensor:
        return torch.cat([_as_columns(d.sample()) for d in self.distributions], dim=-1)

    def argmax(self) -> torch.Tensor:
        return torch.cat([_as_columns(argmax_actions(d)) for d in self.distributions], dim=-1)

    def sample_actions_log_probs(self):
        per_head = [_as_columns(d.sample()) for d in self.distributions]
        logp = sum(d.log_prob(a) for d, a in zip(self.distributions, per_head))
        return torch.cat(per_head, dim=-1), logp

    def log_prob(self, actions: torch.Tensor) -> torch.Tensor:
        return sum(d.log_prob(a) for d, a in zip(self.distributions, self._per_head(actions)))

    def entropy(self) -> torch.Tensor:
        return sum(d.entropy() for d in self.distributions)

    def kl_divergence(self, other: "TupleActionDistribution") -> torch.Tensor:
        return sum(d.kl_divergence(o) for d, o in zip(self.distributions, other.distributions))

    def symmetric_kl_with_uniform_prior(self) -> torch.Tensor:
        return sum(d.symmetric_kl_with_uniform_prior() for d in self.distributions)

    def dbg_print(self) -> str:
        return " | ".join(d.dbg_print() for d in self.distributions if hasattr(d, "dbg_print"))


def _as_columns(a: torch.Tensor) -> torch.Tensor:
    """[N] or [N, k] sampled actions -> [N, k] (heads are concatenated column-wise)"""
    return a.reshape(a.shape[0], -1) if a.dim() > 1 else a.reshape(-1, 1)


class ContinuousActionDistribution(torch.distributions.Independent):
    """diagonal normal over Box(D) actions from `params` = [mean (D) | log_std (D)] — an `Independent(Normal, 1)` exactly
    as the reference builds it (:290-323, validate_args=False: NaN parameters do not raise here either), so `.stddev`,
    `.base_dist`, `.rsample`, `.entropy` and `torch.distributions.kl_divergence` behave as they do there"""
    stddev_min: float = _STD_MIN
    stddev_max: float = _STD_MAX

    def __init__(self, params: torch.Tensor):
        self.means, self.log_std = torch.chunk(params, 2, dim=-1)
        self.stddevs = self.log_std.exp().clamp(self.stddev_min, self.stddev_max)
        super().__init__(torch.distributions.Normal(self.means, self.stddevs, validate_args=False), 1, validate_args=False)

    def log_prob(self, value: torch.Tensor) -> torch.Tensor:
        return super().log_prob(value.reshape(self.means.shape))

    def kl_divergence(self, other: "ContinuousActionDistribution") -> torch.Tensor:
        """KL(self || other) = sum_d log(s_o / s) + (s^2 + (m - m_o)^2) / (2 s_o^2) - 1/2"""
        return torch.distributions.kl.kl_divergence(self, other)

    def summaries(self) -> dict:
        return dict(action_mean=self.means.mean(), action_mean_min=self.means.min(), action_mean_max=self.means.max(),
                    action_log_std_mean=self.log_std.mean(), action_log_std_min=self.log_std.min(),
                    action_log_std_max=self.log_std.max(), action_stddev_mean=self.stddev.mean(),
                    action_stddev_min=self.stddev.min(), action_stddev_max=self.stddev.max())


def get_action_distribution(action_space, raw_logits: torch.Tensor, action_mask: Optional[torch.Tensor] = None):
    """the distribution class that belongs to `action_space` (:45-61); raw_logits [..., calc_num_action_parameters]"""
    assert calc_num_action_parameters(action_space) == raw_logits.shape[-1], \
        f"{calc_num_action_parameters(action_space)} action parameters expected, got {tuple(raw_logits.shape)}"
    if is_discrete(action_space):
        return CategoricalActionDistribution(raw_logits, action_mask)
    if is_box(action_space):
        return ContinuousActionDistribution(raw_logits)
    if len(action_head_sizes(action_space)) > 1 or hasattr(action_space, "spaces"):
        return TupleActionDistribution(action_space, raw_logits, action_mask)
    raise NotImplementedError(f"Action space type {type(action_space)} not supported!")


def sample_actions_log_probs(distribution):
    """(actions, log-probabilities of exactly those actions) (:64-70)"""
    if isinstance(distribution, TupleActionDistribution):
        return distribution.sample_actions_log_probs()
    actions = distribution.sample()
    return actions, distribution.log_prob(actions)


def argmax_actions(distribution) -> torch.Tensor:
    """deterministic evaluation: most likely index per Discrete head, the mean of a normal (:73-81)"""
    if isinstance(distribution, TupleActionDistribution):
        return distribution.argmax()
    if isinstance(distribution, ContinuousActionDistribution):
        return distribution.means
    if hasattr(distribution, "probs"):
        return torch.argmax(distribution.probs, dim=-1)  # [N] (no trailing axis), as the reference returns it
    raise NotImplementedError(f"Action distribution type {type(distribution)} does not support argmax!")

"""`Discretized`: a continuous control axis offered to the policy as n evenly spaced choices
(sample_factory/algo/utils/spaces/discretized.py:4-14; sf_examples/vizdoom/doom/action_space.py builds its turn / look axes
with it).  It IS a Discrete space (`.n`), so the categorical heads, the sampler and the loss kernels take it unchanged;
`to_continuous` is what the env wrapper calls on the sampled index."""
from __future__ import annotations

try:  # a real gymnasium space when gymnasium is installed (env code may isinstance-check it), the bundled descriptor otherwise
    from gymnasium.spaces import Discrete as _Discrete
except Exception:  # noqa: BLE001
    from sample_factory_amd.envs.spaces import Discrete as _Discrete


class Discretized(_Discrete):
    def __init__(self, n, min_action, max_action):
        super().__init__(n)
        self.min_action = min_action
        self.max_action = max_action

    def to_continuous(self, discrete_action):
        """index 0 -> min_action, index n - 1 -> max_action (n = 11 over [-1, 1]: steps of 0.2)"""
        step = (self.max_action - self.min_action) / (self.n - 1)
        return self.min_action + discrete_action * step

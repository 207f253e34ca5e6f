"""`LinearDecay` (sample_factory/utils/decay.py:4-47): a value interpolated piecewise-linearly between (step, value)
milestones and held constant outside them; the learner spaces its summaries with it (every 2 s early in training, every 2 min
after a million SGD steps: learner.py:164, 312-317)."""
from __future__ import annotations

import bisect
import math
from typing import Optional, Sequence, Tuple


class LinearDecay:
    def __init__(self, milestones: Sequence[Tuple[float, float]], staircase: Optional[float] = None):
        """milestones: [(step, value), ...] in any order, e.g. [(0, 100), (1000, 50)] = 100 up to step 0, a straight line to
        50 at step 1000, 50 from there on.  staircase: None = exact values; s = values rounded DOWN to multiples of s (never
        below the first milestone's value, as in the reference)."""
        if len(milestones) == 0:
            raise Exception("Milestones list should not be empty!")
        self._schedule = sorted(milestones)
        self._steps = [m[0] for m in self._schedule]
        self._staircase = staircase

    def at(self, step):
        sch = self._schedule
        if step <= sch[0][0]:
            return sch[0][1]
        if step >= sch[-1][0]:
            return sch[-1][1]
        i = bisect.bisect_left(self._steps, step)  # first milestone at or after `step` (>= 1 here)
        (x0, y0), (x1, y1) = sch[i - 1], sch[i]
        span = x1 - x0
        value = y0 * (1 - (step - x0) / span) + y1 * (1 - (x1 - step) / span)  # the reference's arithmetic, bit for bit
        if self._staircase is None:
            return value
        return max(math.floor(value / self._staircase) * self._staircase, sch[0][1])

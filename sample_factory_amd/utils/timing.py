"""`Timing` — the nested wall-clock profiler every Sample Factory component owns (sample_factory/utils/timing.py:16-161):

    timing = Timing("Runner profile")
    with timing.add_time("rollout"):          # accumulates over calls
        with timing.timeit("inference"):      # keeps the last measurement
            ...
        with timing.time_avg("env_step", 10): # average of the last 10
            ...
    print(timing)                             # tree view, children under the context they were opened in

`timing.<key>` / `timing["key"]` read the flat values (a float, or an AvgTime printing its mean).  Wall-clock around
asynchronous GPU launches measures ENQUEUE time — the reference's own docs say so (docs/07-advanced-topics/profiling.md:147-155);
device-side numbers come from HIP events (`bench.py`, `algo/learning/dp.py`) and rocprofv3."""
from __future__ import annotations

import time
from collections import OrderedDict, deque
from typing import Optional

from sample_factory_amd.algo.utils.misc import EPS
from sample_factory_amd.utils.attr_dict import AttrDict
from sample_factory_amd.utils.utils import log


class AvgTime:
    """the last `num_values_to_avg` measurements; prints their mean"""

    def __init__(self, num_values_to_avg):
        self.values = deque([], maxlen=num_values_to_avg)

    def __str__(self):
        return f"{sum(self.values) / max(1, len(self.values)):.4f}"


class TimingTreeNode:
    def __init__(self, self_time=0.0):
        self.self_time = self_time
        self.timing: "OrderedDict[str, TimingTreeNode]" = OrderedDict()


class TimingContext:
    """one `with` block: measures itself, stores into the flat dict AND into its node of the tree"""

    def __init__(self, timing: "Timing", key: str, additive: bool = False, average: Optional[int] = None):
        self._timing, self._key, self._additive, self._average = timing, key, additive, average
        self.timing_tree_node: Optional[TimingTreeNode] = None
        self._t0 = None

    def set_tree_node(self, node: TimingTreeNode) -> None:
        self.timing_tree_node = node

    def initial_value(self):
        return AvgTime(self._average) if self._average is not None else 0.0

    def __enter__(self):
        self._t0 = time.time()
        self._timing._open_contexts_stack.append(self)
        return self

    def __exit__(self, exc_type, exc, tb):
        dt = max(time.time() - self._t0, EPS)  # never 0: the values are divided by
        node = self.timing_tree_node
        if self._additive:
            self._timing[self._key] += dt
            node.self_time += dt
        elif self._average is not None:
            self._timing[self._key].values.append(dt)
            node.self_time.values.append(dt)
        else:
            self._timing[self._key] = dt
            node.self_time = dt
        self._timing._open_contexts_stack.pop()
        return False


_PRIVATE = ("_root_context", "_open_contexts_stack")  # (the profile's `_name` IS part of flat_str(), as in the reference)


class Timing(AttrDict):
    def __init__(self, name="Profile", *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._name = name
        self._root_context = TimingContext(self, "~")
        self._root_context.set_tree_node(TimingTreeNode())
        self._open_contexts_stack = [self._root_context]

    def _context(self, key: str, **kw) -> TimingContext:
        ctx = TimingContext(self, key, **kw)
        if key not in self:
            self[key] = ctx.initial_value()
        parent = self._open_contexts_stack[-1].timing_tree_node  # the block this one is opened inside
        if key not in parent.timing:
            parent.timing[key] = TimingTreeNode(ctx.initial_value())
        ctx.set_tree_node(parent.timing[key])
        return ctx

    def timeit(self, key: str) -> TimingContext:
        """the LAST measurement of the block"""
        return self._context(key)

    def add_time(self, key: str) -> TimingContext:
        """the SUM over all executions of the block"""
        return self._context(key, additive=True)

    def time_avg(self, key: str, average: int = 10) -> TimingContext:
        """the mean over the last `average` executions"""
        return self._context(key, average=average)

    @staticmethod
    def _fmt(value) -> str:
        return f"{value:.4f}" if isinstance(value, float) else str(value)

    def flat_str(self) -> str:
        return ", ".join(f"{k}: {self._fmt(v)}" for k, v in self.items() if k not in _PRIVATE)

    @classmethod
    def _tree_lines(cls, node: TimingTreeNode, depth: int):
        pad = " " * (2 * depth)
        leaves = [f"{k}: {cls._fmt(v.self_time)}" for k, v in node.timing.items() if not v.timing]
        lines = [pad + ", ".join(leaves)] if leaves else []
        for k, v in node.timing.items():
            if v.timing:
                lines.append(f"{pad}{k}: {cls._fmt(v.self_time)}")
                lines.extend(cls._tree_lines(v, depth + 1))
        return lines

    def tree_str(self) -> str:
        return "\n".join([f"{self._name} tree view:"] + self._tree_lines(self._root_context.timing_tree_node, 0))

    def __str__(self):
        return self.tree_str()


TIMING: Optional[Timing] = None


def init_global_profiler(t: Timing) -> None:
    """debugging aid of the reference (timing.py:156-161): one process-wide Timing; normally it is passed around"""
    global TIMING
    log.info("Setting global profiler in process %d", __import__("os").getpid())
    TIMING = t

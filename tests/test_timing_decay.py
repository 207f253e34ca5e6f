"""`Timing` (utils/timing.py) and `LinearDecay` (utils/decay.py): the reference's profiler / schedule API with this engine's own
implementation.  Held to known answers here and — where /root/reference is present — to the reference's classes on the same
scripted clock (identical tree / flat strings, bit-equal schedule values)."""
import itertools
import os

import pytest

import sample_factory_amd.utils.timing as timing_mod
from sample_factory_amd.utils.decay import LinearDecay
from sample_factory_amd.utils.timing import AvgTime, Timing


def _script(T, mod):
    clock = itertools.count(0.0, 0.25)  # every time.time() call advances the clock by 0.25 s
    orig = mod.time.time
    mod.time.time = lambda: next(clock)
    try:
        t = T("Learner profile")
        for _ in range(3):
            with t.add_time("train"):
                with t.timeit("prepare"):
                    pass
                with t.add_time("epoch"):
                    with t.time_avg("minibatch", 2):
                        pass
                    with t.timeit("losses"):
                        pass
            with t.timeit("publish"):
                pass
        return str(t), t.flat_str(), float(t.train), float(t["epoch"]), str(t.minibatch), t.publish
    finally:
        mod.time.time = orig


def test_timing_tree_known_answer():
    tree, flat, train, epoch, mb, publish = _script(Timing, timing_mod)
    assert tree == ("Learner profile tree view:\npublish: 0.2500\ntrain: 6.7500\n  prepare: 0.2500\n  epoch: 3.7500\n"
                    "    minibatch: 0.2500, losses: 0.2500")
    assert train == 6.75 and epoch == 3.75 and mb == "0.2500" and publish == 0.25
    assert flat.startswith("_name: Learner profile, train: 6.7500, prepare: 0.2500, epoch: 3.7500")
    t = Timing()
    with t.time_avg("x", average=3):
        pass
    assert isinstance(t.x, AvgTime) and len(t.x.values) == 1 and t.x.values[0] > 0  # never 0 (EPS floor)
    with pytest.raises(ZeroDivisionError):
        with t.add_time("failing"):
            1 / 0
    assert t.failing > 0 and len(t._open_contexts_stack) == 1  # the block is closed and counted even when it raises


def test_linear_decay_known_answers():
    d = LinearDecay([(1000000, 120.0), (0, 2.0), (100000, 60.0)])  # the learner's summary spacing (learner.py:164), any order
    assert [d.at(s) for s in (-3, 0, 50000, 100000, 550000, 1000000, 5e6)] == [2.0, 2.0, 31.0, 60.0, 90.0, 120.0, 120.0]
    s = LinearDecay([(0, 100), (1000, 50)], staircase=10)
    assert s.at(250) == 100 and s.at(1000) == 50  # stairs never go below the FIRST milestone's value (the reference's rule)
    with pytest.raises(Exception, match="Milestones"):
        LinearDecay([])


_COMPARE = r"""
import importlib, itertools, sys
from oracle import ref_import  # noqa: F401  (stand-ins for the reference's third-party imports + /root/reference on sys.path)
ref_t = importlib.import_module("sample_factory.utils.timing")
ref_d = importlib.import_module("sample_factory.utils.decay")
assert "/root/reference" in ref_t.__file__, ref_t.__file__
import sample_factory_amd.utils.timing as ours_t
from sample_factory_amd.utils.decay import LinearDecay
sys.path.insert(0, "tests")
from test_timing_decay import _script
assert _script(ref_t.Timing, ref_t) == _script(ours_t.Timing, ours_t)
ms = [(0, 2.0), (100000, 60.0), (1000000, 120.0)]
pts = [-3, 0, 1, 999, 50000, 100000, 100001, 555555, 777777.7, 1000000, 2000000]
for st in (None, 10, 0.5):
    assert [ref_d.LinearDecay(ms, st).at(p) for p in pts] == [LinearDecay(ms, st).at(p) for p in pts], st
print("EQUAL")
"""


@pytest.mark.skipif(not os.path.isdir("/root/reference/sample_factory"), reason="needs /root/reference")
def test_equal_to_the_reference_classes():
    """in a process of its own: importing the reference installs stand-ins for gymnasium & co. into sys.modules"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", _COMPARE], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "EQUAL" in r.stdout, r.stdout[-500:] + r.stderr[-1500:]


def test_runner_profiles_its_loop_and_learner_spaces_summaries_with_linear_decay():
    """the two places the engine itself uses them, driven without a GPU: Runner.iteration() (synchronous mode) on stubbed
    halves, Learner._should_save_summaries() on a stub with the class's schedule"""
    import time

    from sample_factory_amd.algo.learning.learner import Learner
    from sample_factory_amd.train import Runner
    from sample_factory_amd.utils.attr_dict import AttrDict

    r = Runner(AttrDict(async_rl=False))
    calls = []
    r.learner = AttrDict(train_step=7)
    r._ready = []

    def rollout(version):
        calls.append(("rollout", version))
        if len(calls) == 2:
            r._ready.append(slice(0, 4))

    r._rollout_all = rollout
    r._train_dataset = lambda ds: calls.append(("train", ds)) or dict(ok=1)
    assert r.iteration() == dict(ok=1)
    assert calls == [("rollout", 7.0), ("rollout", 7.0), ("train", slice(0, 4))]
    assert r.timing.rollout > 0 and r.timing.train > 0 and str(r.timing).startswith("Runner profile tree view:\nrollout: ")

    assert Learner._summary_rate_decay.at(0) == 2.0 and Learner._summary_rate_decay.at(10 ** 7) == 120.0
    stub = AttrDict(train_step=50000, cfg=AttrDict(), _summary_rate_decay=Learner._summary_rate_decay,
                    _last_summary_time=time.time() - 30.0)  # 31 s between summaries at 50 k SGD steps
    assert Learner._should_save_summaries(stub) is False
    stub._last_summary_time = time.time() - 32.0
    assert Learner._should_save_summaries(stub) is True
    stub.cfg.summaries_every_train = True
    stub._last_summary_time = time.time()
    assert Learner._should_save_summaries(stub) is True

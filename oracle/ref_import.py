"""Import shim for the *reference* (Sample Factory): /root/reference inside the build container, or — where that does
not exist (the GPU box) — the archive `make -C oracle ref` staged from it under oracle/_ref/.

TEST / BENCH INFRASTRUCTURE ONLY.  This module is used by ``oracle/gen_golden.py`` (golden vectors generated from the
reference itself) and by ``oracle/ref_cpu_tier_b.py`` (bench.py's cpu_baseline leg, a subprocess of its own: the
reference timed on the host cores) to import the reference's own Python modules.  The
reference cannot be imported as-is here because six third-party packages it imports at module top are not
installed (gymnasium, signal_slot/faster_fifo, colorlog, tensorboardX, cv2, wandb) and there is no network.
We register minimal stand-ins for those names in ``sys.modules`` *before* importing ``sample_factory``.
None of the stubs implements any arithmetic that is on the hot path: spaces are shape/dtype holders, the
event loop / queue / logger / summary-writer classes are inert.

Nothing under ``sample_factory_amd/`` may import this file; ``/root/reference`` does not exist on the GPU box.
"""
from __future__ import annotations

import logging
import queue
import sys
import types

import numpy as np

from oracle.ref_import_path import REFERENCE_ROOT, reference_available  # noqa: E402,F401


def _mod(name: str) -> types.ModuleType:
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


# ----------------------------------------------------------------------------- gymnasium
class _Space:
    def __init__(self, shape=None, dtype=None):
        self.shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)

    def sample(self):
        raise NotImplementedError

    def seed(self, seed=None):
        return [seed]


class Discrete(_Space):
    def __init__(self, n, start=0):
        super().__init__((), np.int64)
        self.n = int(n)
        self.start = int(start)

    def sample(self):
        return np.random.randint(self.n)

    def __repr__(self):
        return f"Discrete({self.n})"


class Box(_Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.asarray(low).shape
        super().__init__(shape, dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape).copy()

    def sample(self):
        return np.random.uniform(-1, 1, self.shape).astype(self.dtype)

    def __repr__(self):
        return f"Box{self.shape}"


class Tuple(_Space):
    def __init__(self, spaces):
        super().__init__(None, None)
        self.spaces = tuple(spaces)

    def __iter__(self):
        return iter(self.spaces)

    def __len__(self):
        return len(self.spaces)

    def __getitem__(self, i):
        return self.spaces[i]


class Dict(_Space):
    def __init__(self, spaces=None, **kw):
        super().__init__(None, None)
        self.spaces = dict(spaces or {})
        self.spaces.update(kw)

    def keys(self):
        return self.spaces.keys()

    def items(self):
        return self.spaces.items()

    def values(self):
        return self.spaces.values()

    def __getitem__(self, k):
        return self.spaces[k]

    def __iter__(self):
        return iter(self.spaces)

    def __len__(self):
        return len(self.spaces)


class Env:
    metadata: dict = {}
    observation_space = None
    action_space = None
    render_mode = None

    def reset(self, **kwargs):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError

    def close(self):
        pass

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    def __init__(self, env):
        self.env = env
        self.observation_space = env.observation_space
        self.action_space = env.action_space

    def __getattr__(self, name):
        return getattr(self.env, name)

    def reset(self, **kwargs):
        return self.env.reset(**kwargs)

    def step(self, action):
        return self.env.step(action)

    @property
    def unwrapped(self):
        return self.env.unwrapped


def install_stubs() -> None:
    if "gymnasium" in sys.modules and getattr(sys.modules["gymnasium"], "_sf_amd_stub", False):
        return
    gym = _mod("gymnasium")
    gym._sf_amd_stub = True
    spaces = _mod("gymnasium.spaces")
    for cls in (Discrete, Box, Tuple, Dict):
        setattr(spaces, cls.__name__, cls)
    spaces.Space = _Space
    gym.spaces = spaces
    gym.Space = _Space
    gym.Env = Env
    gym.Wrapper = Wrapper
    gym.ObservationWrapper = Wrapper
    gym.RewardWrapper = Wrapper
    gym.ActionWrapper = Wrapper
    gym.make = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("gymnasium stub: no envs"))
    core = _mod("gymnasium.core")
    core.ActType = object
    core.ObsType = object
    core.Env = Env
    core.Wrapper = Wrapper
    gym.core = core
    wrappers = _mod("gymnasium.wrappers")
    gym.wrappers = wrappers
    utils = _mod("gymnasium.utils")
    gym.utils = utils

    colorlog = _mod("colorlog")

    class ColoredFormatter(logging.Formatter):
        def __init__(self, fmt=None, datefmt=None, *a, **k):
            super().__init__("%(message)s", datefmt)

    colorlog.ColoredFormatter = ColoredFormatter

    ss_pkg = _mod("signal_slot")
    ss = _mod("signal_slot.signal_slot")
    ss_pkg.signal_slot = ss

    class _Inert:
        def __init__(self, *a, **k):
            pass

        def __getattr__(self, name):
            return _Inert()

        def __call__(self, *a, **k):
            return _Inert()

    class EventLoopObject:
        def __init__(self, event_loop=None, object_id=None):
            self.event_loop = event_loop
            self.object_id = object_id

        def emit(self, *a, **k):
            pass

        def connect(self, *a, **k):
            pass

        def detach(self):
            pass

    class EventLoop(EventLoopObject):
        def __init__(self, unique_loop_name="loop", serial_mode=True):
            super().__init__(self, unique_loop_name)
            self.owner = None

        def exec(self):
            return 0

    class EventLoopStatus:
        NORMAL_TERMINATION, INTERRUPTED = 0, 1

    ss.EventLoop = EventLoop
    ss.EventLoopObject = EventLoopObject
    ss.EventLoopProcess = _Inert
    ss.EventLoopStatus = EventLoopStatus
    ss.Timer = _Inert
    ss.TightLoop = _Inert
    ss.BoundMethod = _Inert
    ss.StatusCode = int
    ss.signal = lambda f: f
    ss.process_name = lambda *a, **k: "main"
    ss.configure_logger = lambda *a, **k: None
    ss.log = logging.getLogger("signal_slot_stub")
    qu = _mod("signal_slot.queue_utils")
    qu.get_queue = lambda serial=True, buffer_size_bytes=0: queue.Queue()
    ss_pkg.queue_utils = qu

    tbx = _mod("tensorboardX")
    tbx.SummaryWriter = _Inert
    _mod("cv2")
    _mod("wandb")
    ff = _mod("faster_fifo")
    ff.Queue = queue.Queue

    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


install_stubs()

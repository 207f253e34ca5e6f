"""Process-context helpers an env integration imports (sample_factory/algo/utils/multiprocessing_utils.py:8-42:
`get_mp_ctx` in sf_examples/dmlab/train_dmlab.py, `get_mp_lock` in dmlab_level_cache.py).  The engine itself starts its
env worker processes from `algo/sampling/parallel_env.py` with the same "spawn" context."""
from __future__ import annotations

import multiprocessing
from multiprocessing.context import BaseContext
from typing import Optional

_SPAWN: Optional[BaseContext] = None


def get_mp_ctx(serial: bool) -> Optional[BaseContext]:
    """None in serial mode, otherwise the one shared "spawn" context (fork would duplicate a live HIP runtime)"""
    global _SPAWN
    if serial:
        return None
    if _SPAWN is None:
        _SPAWN = multiprocessing.get_context("spawn")
    return _SPAWN


class FakeLock:
    """what serial mode hands out instead of an OS lock: every operation is a no-op"""

    def acquire(self, *args, **kwargs):
        return True

    def release(self, *args, **kwargs):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def get_mp_lock(mp_ctx: Optional[BaseContext] = None):
    return (multiprocessing if mp_ctx is None else mp_ctx).Lock()


def get_lock(serial: bool = False, mp_ctx: Optional[BaseContext] = None):
    return FakeLock() if serial else get_mp_lock(mp_ctx)

"""The helper names a Sample Factory user script imports from `sample_factory.utils.utils`
(reference: sample_factory/utils/utils.py — `log` :26-53, `init_file_logger` :65-80, `is_module_available` :83-88,
`static_vars` :126-132, `str2bool` :191-199, directory helpers :361-425, `log_every_n` :480-497, CPU affinity :314-358).

Own implementation on the standard library only (the reference needs colorlog / signal_slot / psutil): one process-wide
logger named "rl" writing `[time][pid] message` to stderr, ANSI colours when stderr is a terminal.
"""
from __future__ import annotations

import argparse
import getpass
import importlib.util
import logging
import os
import sys
import tempfile
from os.path import join

# ---------------------------------------------------------------------------------------------------------- logging
_COLOURS = {logging.DEBUG: "\033[36m", logging.INFO: "\033[1;37m", logging.WARNING: "\033[33m",
            logging.ERROR: "\033[1;31m", logging.CRITICAL: "\033[1;31;47m"}


class _Formatter(logging.Formatter):
    def __init__(self, colour: bool):
        super().__init__("[%(asctime)s][%(process)05d] %(message)s")
        self.colour = colour

    def format(self, record):
        s = super().format(record)
        if self.colour:
            s = _COLOURS.get(record.levelno, "") + s + "\033[0m"
        return s


log = logging.getLogger("rl")
log.setLevel(logging.DEBUG)
log.propagate = False
if not log.handlers:
    _h = logging.StreamHandler()
    _h.setLevel(logging.DEBUG if os.environ.get("SF_LOG_LEVEL", "").lower() == "debug" else logging.INFO)
    _h.setFormatter(_Formatter(colour=hasattr(sys.stderr, "isatty") and sys.stderr.isatty()))
    log.addHandler(_h)


def has_file_handler() -> bool:
    return any(isinstance(h, logging.FileHandler) for h in log.handlers)


def init_file_logger(cfg) -> None:
    """sf_log.txt inside the experiment directory (cfg.log_to_file), once per process"""
    if not getattr(cfg, "log_to_file", True) or has_file_handler():
        return
    fh = logging.FileHandler(join(experiment_dir(cfg), "sf_log.txt"))
    fh.setLevel(logging.DEBUG)
    fh.setFormatter(_Formatter(colour=False))
    log.addHandler(fh)


def static_vars(**kwargs):
    """decorator: attach attributes to a function (a poor man's function-static variable)"""
    def decorate(func):
        for k, v in kwargs.items():
            setattr(func, k, v)
        return func
    return decorate


@static_vars(history=dict())
def log_every_n(n, _level, msg, *args, **kwargs):
    """log `msg` on every n-th call with that message (call count keyed by the format string only)"""
    seen = log_every_n.history.get(msg, 0)
    if seen % n == 0:
        log.log(_level, f"{msg} ({seen} times)" if seen > 1 else msg, *args, **kwargs)
    log_every_n.history[msg] = seen + 1


def debug_log_every_n(n, msg, *args, **kwargs):
    log_every_n(n, logging.DEBUG, msg, *args, **kwargs)


# ---------------------------------------------------------------------------------------------------------- CLI / misc
def str2bool(v):
    """argparse type for the boolean flags: only 'true' / 'false' (any case) and real bools are accepted"""
    if isinstance(v, bool):
        return v
    if isinstance(v, str) and v.lower() == "true":
        return True
    if isinstance(v, str) and v.lower() == "false":
        return False
    raise argparse.ArgumentTypeError("Boolean value expected")


def is_module_available(module_name: str) -> bool:
    try:
        return importlib.util.find_spec(module_name) is not None
    except (ImportError, ValueError):
        return False


def set_attr_if_exists(obj, attr_name, attr_value) -> None:
    if hasattr(obj, attr_name):
        setattr(obj, attr_name, attr_value)


def scale_to_range(np_array, min_, max_):
    lo, hi = np_array.min(), np_array.max()
    if hi - lo < 1e-8:
        return np_array * 0 + (min_ + max_) / 2
    return (np_array - lo) / (hi - lo) * (max_ - min_) + min_


def memory_consumption_mb() -> float:
    """resident set of this process in MB (/proc; psutil is not a dependency)"""
    try:
        with open("/proc/self/statm") as f:
            return int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE") / (1024 * 1024)
    except Exception:  # noqa: BLE001
        return 0.0


# ---------------------------------------------------------------------------------------------------------- CPU affinity
def cores_for_worker_process(worker_idx: int, num_workers: int, cpu_count: int):
    """the cores worker `worker_idx` of `num_workers` is pinned to: an even share when there are more cores than workers
    (cores divisible) or more workers than cores (workers divisible), otherwise None = leave it to the scheduler"""
    worker_idx = worker_idx % num_workers
    if cpu_count > num_workers:
        if cpu_count % num_workers != 0:
            return None
        per = cpu_count // num_workers
        return list(range(worker_idx * per, (worker_idx + 1) * per))
    if num_workers % cpu_count != 0:
        return None
    return [worker_idx % cpu_count]


def set_process_cpu_affinity(worker_idx: int, num_workers: int) -> None:
    if not hasattr(os, "sched_getaffinity"):
        return
    available = sorted(os.sched_getaffinity(0))
    cores = cores_for_worker_process(worker_idx, num_workers, len(available))
    if cores is not None:
        os.sched_setaffinity(0, [available[c] for c in cores])
    log.debug("Worker %d uses CPU cores %r", worker_idx, sorted(os.sched_getaffinity(0)))


# ---------------------------------------------------------------------------------------------------------- directories
def ensure_dir_exists(path) -> str:
    os.makedirs(path, exist_ok=True)
    return path


def maybe_ensure_dir_exists(path, mkdir: bool) -> str:
    return ensure_dir_exists(path) if mkdir else path


safe_ensure_dir_exists = ensure_dir_exists


def remove_if_exists(file) -> None:
    if os.path.isfile(file):
        os.remove(file)


def get_username() -> str:
    try:
        return getpass.getuser()
    except Exception:  # noqa: BLE001 - no passwd entry for the uid (containers)
        return str(os.getuid())


def project_tmp_dir(mkdir: bool = True) -> str:
    """per-user scratch directory under the shared temporary directory; created (and kept) private: mode 0700"""
    path = join(tempfile.gettempdir(), f"sf2_{get_username()}")
    if mkdir:
        os.makedirs(path, mode=0o700, exist_ok=True)
        try:
            if os.stat(path).st_uid == os.getuid():
                os.chmod(path, 0o700)
        except OSError:
            pass
    return path


def experiments_dir(cfg, mkdir=True) -> str:
    return maybe_ensure_dir_exists(cfg.train_dir, mkdir)


def experiment_dir(cfg, mkdir=True) -> str:
    return maybe_ensure_dir_exists(join(experiments_dir(cfg, mkdir), cfg.experiment), mkdir)


def summaries_dir(experiment_dir_, mkdir=True) -> str:
    return maybe_ensure_dir_exists(join(experiment_dir_, ".summary"), mkdir)


def cfg_file(cfg) -> str:
    return join(experiment_dir(cfg=cfg), "config.json")

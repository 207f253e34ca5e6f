"""Configuration surface: the reference's flag names and defaults for every flag the hot path reads
(sample_factory/cfg/cfg.py:9-819, cfg/arguments.py:24-62).  Flag names/defaults are the API; nothing else of the
reference's CLI machinery (two-pass env-specific overrides, config.json merging, PBT/wandb groups) is rebuilt.
"""
from __future__ import annotations

import argparse
import os
from typing import List, Optional, Tuple


def _bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("true", "1", "yes", "y", "t"):
        return True
    if v.lower() in ("false", "0", "no", "n", "f"):
        return False
    raise argparse.ArgumentTypeError(f"boolean expected, got {v}")


# (name, type, default [, nargs])
FLAGS = [
    ("algo", str, "APPO"), ("env", str, None), ("experiment", str, "default_experiment"),
    ("train_dir", str, os.path.join(os.getcwd(), "train_dir")), ("restart_behavior", str, "resume"),
    ("device", str, "gpu"), ("seed", int, None),
    ("num_policies", int, 1), ("async_rl", _bool, True), ("serial_mode", _bool, False),
    ("batched_sampling", _bool, False), ("num_batches_to_accumulate", int, 2), ("worker_num_splits", int, 2),
    ("policy_workers_per_policy", int, 1), ("max_policy_lag", int, 1000),
    ("num_workers", int, 8), ("num_envs_per_worker", int, 2),
    ("batch_size", int, 1024), ("num_batches_per_epoch", int, 1), ("num_epochs", int, 1), ("rollout", int, 32),
    ("recurrence", int, -1), ("shuffle_minibatches", _bool, False),
    ("gamma", float, 0.99), ("reward_scale", float, 1.0), ("reward_clip", float, 1000.0),
    ("value_bootstrap", _bool, False), ("normalize_returns", _bool, True),
    ("exploration_loss_coeff", float, 0.003), ("value_loss_coeff", float, 0.5), ("kl_loss_coeff", float, 0.0),
    ("exploration_loss", str, "entropy"), ("gae_lambda", float, 0.95), ("ppo_clip_ratio", float, 0.1),
    ("ppo_clip_value", float, 1.0), ("with_vtrace", _bool, False), ("vtrace_rho", float, 1.0),
    ("vtrace_c", float, 1.0),
    ("optimizer", str, "adam"), ("adam_eps", float, 1e-6), ("adam_beta1", float, 0.9), ("adam_beta2", float, 0.999),
    ("max_grad_norm", float, 4.0), ("learning_rate", float, 1e-4), ("lr_schedule", str, "constant"),
    ("lr_schedule_kl_threshold", float, 0.008), ("lr_adaptive_min", float, 1e-6), ("lr_adaptive_max", float, 1e-2),
    ("obs_subtract_mean", float, 0.0), ("obs_scale", float, 1.0), ("normalize_input", _bool, True),
    ("normalize_input_keys", str, None, "*"),
    ("decorrelate_experience_max_seconds", int, 0), ("decorrelate_envs_on_one_worker", _bool, True),
    ("actor_worker_gpus", int, [], "*"), ("set_workers_cpu_affinity", _bool, True),
    ("force_envs_single_thread", _bool, False), ("default_niceness", int, 0),
    ("log_to_file", _bool, True), ("experiment_summaries_interval", int, 10), ("flush_summaries_interval", int, 30),
    ("stats_avg", int, 100), ("summaries_use_frameskip", _bool, True), ("heartbeat_interval", int, 20),
    ("heartbeat_reporting_interval", int, 180),
    ("train_for_env_steps", int, int(1e10)), ("train_for_seconds", int, int(1e10)),
    ("save_every_sec", int, 120), ("keep_checkpoints", int, 2), ("load_checkpoint_kind", str, "latest"),
    ("save_milestones_sec", int, -1), ("save_best_every_sec", int, 5), ("save_best_metric", str, "reward"),
    ("save_best_after", int, 100000), ("benchmark", _bool, False),
    ("encoder_mlp_layers", int, [512, 512], "*"), ("encoder_conv_architecture", str, "convnet_simple"),
    ("encoder_conv_mlp_layers", int, [512], "*"), ("use_rnn", _bool, True), ("rnn_size", int, 512),
    ("rnn_type", str, "gru"), ("rnn_num_layers", int, 1), ("decoder_mlp_layers", int, [], "*"),
    ("nonlinearity", str, "elu"), ("policy_initialization", str, "orthogonal"), ("policy_init_gain", float, 1.0),
    ("actor_critic_share_weights", _bool, True), ("adaptive_stddev", _bool, True),
    ("continuous_tanh_scale", float, 0.0), ("initial_stddev", float, 1.0),
    ("use_env_info_cache", _bool, False), ("env_gpu_actions", _bool, False), ("env_gpu_observations", _bool, True),
    ("env_frameskip", int, 1), ("env_framestack", int, 1), ("pixel_format", str, "CHW"),
    ("use_record_episode_statistics", _bool, False), ("with_wandb", _bool, False), ("with_pbt", _bool, False),
    ("help", _bool, False),
]


# flags of THIS engine (no counterpart in the reference's cfg; everything above mirrors cfg/cfg.py)
ENGINE_FLAGS = [
    ("data_parallel", _bool, False),         # learner replicas over torch.distributed (set automatically when WORLD_SIZE > 1)
    ("dp_overlap", _bool, True),             # all-reduce the fc/heads gradient bucket while the conv layers back-propagate
    ("dp_epoch_moments", _bool, True),       # GAE: every minibatch's advantage moments exchanged once per epoch (False: one 24-byte all-reduce per SGD step)
    ("dp_oneshot_bytes", int, 0),            # > 0: f32 / f64 SUM buckets up to this size (the 0.31 MB conv bucket, the moment / invalid-count scalars) go through the one-shot mailbox exchange (sf_dp_oneshot_*) instead of a ring all-reduce
    ("dp_native_rccl", _bool, False),        # gradient buckets through the C-ABI (sf_allreduce_grads) instead of torch.distributed
    ("dp_force_collectives", _bool, False),  # issue the collectives in a group of one rank (tests)
    ("device_shuffle", _bool, False),        # shuffle_minibatches with the stateless on-device permutation
    ("sampler_thread", _bool, None),         # None: a sampler thread iff async_rl with a host env
    ("record_grad_norm", _bool, False),
    ("env_workers_mode", str, "auto"),       # host envs: "process" = cfg.num_workers env worker processes (the reference's
    #                                          rollout workers), "inline" = in this process, "auto" = processes iff
    #                                          serial_mode=False and the env is a host env (algo/sampling/parallel_env.py)
    ("env_worker_start_method", str, "spawn"),  # multiprocessing start method of the env workers
]


def parse_sf_args(argv: Optional[List[str]] = None, evaluation: bool = False) -> Tuple[argparse.ArgumentParser, argparse.Namespace]:
    """cfg/arguments.py:24-52 — returns (parser, partially parsed args); scripts may add args to the parser."""
    import sys

    if argv is None:
        argv = sys.argv[1:]
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter, add_help=False)
    for spec in FLAGS + ENGINE_FLAGS:
        name, typ, default = spec[:3]
        kw = dict(type=typ, default=default)
        if len(spec) > 3:
            kw["nargs"] = spec[3]
        p.add_argument(f"--{name}", **kw)
    if evaluation:
        p.add_argument("--max_num_episodes", type=int, default=int(1e9))
        p.add_argument("--max_num_frames", type=int, default=int(1e9))
        p.add_argument("--eval_deterministic", type=_bool, default=False)
        p.add_argument("--no_render", type=_bool, default=True)
        p.add_argument("--policy_index", type=int, default=0)
    args, _ = p.parse_known_args(argv)
    return p, args


def parse_full_cfg(parser: argparse.ArgumentParser, argv: Optional[List[str]] = None) -> argparse.Namespace:
    """cfg/arguments.py:55-62"""
    import sys

    if argv is None:
        argv = sys.argv[1:]
    args = parser.parse_args(argv)
    args.command_line = " ".join(argv)
    # the flags that were actually given (cfg/arguments.py:64-72): what load_from_checkpoint lets override a saved config
    given = argparse.ArgumentParser(add_help=False, argument_default=argparse.SUPPRESS)
    for a in parser._actions:
        if a.option_strings:
            given.add_argument(*a.option_strings, type=a.type, nargs=a.nargs, default=argparse.SUPPRESS)
    args.cli_args = vars(given.parse_known_args(argv)[0])
    return args


def default_cfg(**overrides) -> argparse.Namespace:
    """Convenience: the reference defaults with keyword overrides (used by tests, bench.py and examples)."""
    parser, _ = parse_sf_args([])
    cfg = parser.parse_args([])
    for k, v in overrides.items():
        setattr(cfg, k, v)
    cfg.command_line = ""
    return cfg


def preprocess_cfg(cfg, env_info) -> bool:
    """cfg/arguments.py:97-103: recurrence=-1 -> rollout if use_rnn else 1; then verify."""
    if cfg.recurrence == -1:
        cfg.recurrence = cfg.rollout if cfg.use_rnn else 1
    return verify_cfg(cfg, env_info)


def verify_cfg(cfg, env_info) -> bool:
    """The constraints of cfg/arguments.py:105-201 that concern the hot path."""
    ok = True

    def err(msg):
        nonlocal ok
        ok = False
        print(f"[sample_factory_amd] config error: {msg}")

    if cfg.num_policies != 1:
        err(f"{cfg.num_policies=}: this engine trains ONE policy per process group (multi-policy / PBT populations are "
            "outside the hot-path scope)")
    if cfg.num_envs_per_worker % cfg.worker_num_splits != 0:
        err(f"{cfg.num_envs_per_worker=} must be a multiple of {cfg.worker_num_splits=}"
            " (for double-buffered sampling you need to use even number of envs per worker)")
    if cfg.normalize_returns and cfg.with_vtrace:
        err("Normalized returns are not supported with vtrace!")
    if cfg.with_vtrace and not (cfg.recurrence == cfg.rollout and cfg.recurrence > 1):
        err("V-trace requires recurrence == rollout > 1")
    if cfg.rollout % cfg.recurrence != 0:
        err(f"{cfg.rollout=} must be a multiple of {cfg.recurrence=}")
    if cfg.batch_size % cfg.rollout != 0:
        err(f"{cfg.batch_size=} must be a multiple of {cfg.rollout=}")
    if cfg.use_rnn and cfg.recurrence <= 1:
        err("RNN policies need recurrence > 1")
    total_agents = cfg.num_workers * cfg.num_envs_per_worker * env_info.num_agents
    per_iter = cfg.num_batches_per_epoch * cfg.batch_size
    per_rollout = total_agents * cfg.rollout // cfg.num_policies
    if not cfg.async_rl and not (per_iter % per_rollout == 0 and per_iter >= per_rollout):
        err(f"sync mode needs batch_size*num_batches_per_epoch ({per_iter}) to be a multiple of "
            f"agents*rollout ({per_rollout})")
    return ok


def cfg_dict(cfg) -> dict:
    return dict(cfg) if isinstance(cfg, dict) else dict(vars(cfg))


def cfg_str(cfg) -> str:
    return "\n".join(f"{k}={v}" for k, v in cfg_dict(cfg).items())


def load_from_checkpoint(cfg):
    """cfg/arguments.py:227-260: the experiment's saved config.json is the base configuration; flags given on the command
    line (cfg.cli_args) override it; flags the file does not know are added"""
    import json

    from sample_factory_amd.utils.attr_dict import AttrDict
    name = os.path.join(cfg.train_dir, cfg.experiment, "config.json")
    if not os.path.isfile(name):
        raise FileNotFoundError(f"Could not load saved parameters for experiment {cfg.experiment} (file {name} not found). "
                                "Check that you have the correct experiment name and --train_dir is set correctly.")
    with open(name) as f:
        loaded = AttrDict(json.load(f))
    for key, value in getattr(cfg, "cli_args", {}).items():
        if key in loaded and loaded[key] != value:
            loaded[key] = value
    for key, value in cfg_dict(cfg).items():
        if key not in loaded:
            loaded[key] = value
    return loaded


def maybe_load_from_checkpoint(cfg):
    """cfg/arguments.py:263-275: resume = saved configuration + command-line overrides; no saved configuration = a fresh
    experiment with the given one (returned as a new AttrDict: the caller's object is never mutated)"""
    from sample_factory_amd.utils.attr_dict import AttrDict
    from sample_factory_amd.utils.utils import log
    if not os.path.isfile(os.path.join(cfg.train_dir, cfg.experiment, "config.json")):
        log.warning("Saved parameter configuration for experiment %s not found! Starting experiment from scratch!",
                    cfg.experiment)
        return AttrDict(cfg_dict(cfg))
    return load_from_checkpoint(cfg)


def checkpoint_override_defaults(cfg, parser) -> None:
    """cfg/arguments.py:75-94 (used by enjoy-style scripts with their own parsers): the saved configuration becomes the
    parser's defaults so that a second parse only overrides what the command line names"""
    import json
    name = os.path.join(cfg.train_dir, cfg.experiment, "config.json")
    if os.path.isfile(name):
        with open(name) as f:
            parser.set_defaults(**json.load(f))

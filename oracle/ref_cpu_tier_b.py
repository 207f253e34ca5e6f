"""SURVEY.md §8(d) "Tier B" CPU baseline: the REFERENCE's own code (real ActorCritic.forward through
prepare_and_normalize_obs, real Learner.train) timed on host cores for the BASELINE configs[1] workload shape.

TEST/BENCH INFRASTRUCTURE; runs where the reference is importable — /root/reference in the build container, or the
archive `make -C oracle ref` staged from it (oracle/_ref/, travels to the GPU box) — under the import stubs of
oracle/ref_import.py, always as a process of its own (bench.py's cpu_baseline leg starts it AFTER the timed region).
A bounded sample is timed and extrapolated linearly to one 4096-env x 32-step iteration:
env-steps/s = dataset size / (T x t_inference(4096 obs) + t_train(dataset)).

  python -m oracle.ref_cpu_tier_b [envs_sample] [inference_seconds]
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

from oracle import ref_import  # noqa: F401
import gymnasium as gym  # the stub
from oracle.gen_golden import C2_MODEL_ARGS, fill_batch, make_cfg, make_learner
from sample_factory.algo.utils.rl_utils import prepare_and_normalize_obs
from sample_factory.algo.utils.shared_buffers import alloc_trajectory_tensors
from sample_factory.model.model_utils import get_rnn_size


def main():
    cores = len(os.sched_getaffinity(0))
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 128   # trajectories in the timed sample
    t_budget = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
    threads_arg = sys.argv[3] if len(sys.argv) > 3 else "auto"
    torch.set_num_threads(cores if threads_arg == "auto" else int(threads_arg))
    T, nb, full_envs = 32, 4, 4096
    obs_space = gym.spaces.Dict({"obs": gym.spaces.Box(0, 255, (4, 84, 84), np.uint8)})
    cfg = make_cfg(C2_MODEL_ARGS + [f"--rollout={T}", f"--batch_size={E * T // nb}", f"--num_batches_per_epoch={nb}",
                                    "--num_epochs=1", "--exploration_loss_coeff=0.01"])
    learner, env_info = make_learner(cfg, obs_space, gym.spaces.Discrete(6), E)
    ac = learner.actor_critic
    g = torch.Generator().manual_seed(0)
    b = alloc_trajectory_tensors(env_info, E, T, get_rnn_size(cfg), "cpu", False)
    fill_batch(b, g, 6, p_done=0.01)
    # ---- inference: one policy step on `E` observations (inference_worker.py:313-341), eval mode, no grad
    ac.eval()
    obs0 = {"obs": b["obs"]["obs"][:, 0].clone()}
    rnn0 = b["rnn_states"][:, 0].clone()
    scan = {}
    with torch.no_grad():
        ac(prepare_and_normalize_obs(ac, obs0), rnn0)  # warm-up
        if threads_arg == "auto":
            # torch's intra-op pool on every core of a many-core host is far from its best operating point for these batch
            # sizes (256 threads: 2.8 s per 512-obs forward on the MI355X host): time the forward at a few pool sizes and
            # keep the fastest for BOTH legs — the baseline is the reference at ITS best thread count on this box
            for th in sorted({t for t in (8, 16, 32, 64, 128, cores) if t <= cores}):
                torch.set_num_threads(th)
                ac(prepare_and_normalize_obs(ac, obs0), rnn0)
                t0, reps = time.perf_counter(), 0
                while time.perf_counter() - t0 < 0.6 and reps < 50:
                    ac(prepare_and_normalize_obs(ac, obs0), rnn0)
                    reps += 1
                scan[th] = (time.perf_counter() - t0) / reps
            torch.set_num_threads(min(scan, key=scan.get))
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < t_budget:
            ac(prepare_and_normalize_obs(ac, obs0), rnn0)
            reps += 1
        t_inf = (time.perf_counter() - t0) / reps
    # ---- Learner.train on the E x T dataset
    ac.train()
    t1 = time.perf_counter()
    learner.train(b)
    t_train = time.perf_counter() - t1
    scale = full_envs / E
    t_iter = T * t_inf * scale + t_train * scale
    print(json.dumps(dict(
        value=round(full_envs * T / t_iter, 1), unit="env-steps/s", cores=torch.get_num_threads(), host_cores=cores,
        kind="reference", thread_scan_ms_per_forward={str(k): round(v * 1e3, 1) for k, v in scan.items()},
        sample=f"reference ActorCritic.forward on {E} obs ({t_inf * 1e3:.1f} ms, {reps} reps) and Learner.train on a "
               f"{E}x{T} dataset in {nb} minibatches ({t_train:.2f} s), torch {torch.__version__} CPU fp32, "
               f"{torch.get_num_threads()} threads (best of the pool sizes tried) of {cores} host cores; extrapolated x{scale:.0f} to one {full_envs}x{T} iteration (env excluded)",
        t_inference_ms_per_step_sample=round(t_inf * 1e3, 2), t_train_s_sample=round(t_train, 3),
        reference_from=ref_import.REFERENCE_ROOT,
        where="build container" if os.path.isdir("/root/reference") else "GPU box host cores")))


if __name__ == "__main__":
    main()

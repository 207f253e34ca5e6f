"""SURVEY.md §8(d) "Tier B" CPU baseline: the REFERENCE's own code (real ActorCritic.forward through
prepare_and_normalize_obs, real Learner.train) timed on host cores for the BASELINE configs[1] workload shape.

TEST/BENCH INFRASTRUCTURE; runs where the reference is importable — /root/reference in the build container, or the
archive `make -C oracle ref` staged from it (oracle/_ref/, travels to the GPU box) — under the import stubs of
oracle/ref_import.py, always as a process of its own (bench.py's cpu_baseline leg starts it AFTER the timed region).
A bounded sample (bench.py: 1024 of the 4096 trajectories; 4096 = the full Tier-B workload, ~45 GB of the reference's f32
frame copies and ~30 s per Learner.train call) is timed and extrapolated linearly to one 4096-env x 32-step iteration:
env-steps/s = dataset size / (T x t_inference(4096 obs) + t_train(dataset)).  Learner.train runs once untimed, then
`train_repeats` times on fresh copies of the dataset; the value uses the median and the line carries the spread.

  python -m oracle.ref_cpu_tier_b [envs_sample] [inference_seconds] [threads|auto] [train_repeats]
  python -m oracle.ref_cpu_tier_b --device cuda [envs] [inference_seconds]

`--device cuda` (context only, reported by bench.py as secondary workload "reference_torch_rocm"): the SAME reference code
with `--device=gpu`, i.e. its stock PyTorch-ROCm / MIOpen path on the MI355X of the box, at the FULL workload size
(4096 trajectories x 32 steps, nothing extrapolated), after one untimed iteration that absorbs MIOpen's kernel search.
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


def main_cuda(argv):
    """the reference on the GPU of this box through stock PyTorch-ROCm (learner.py / actor_critic.py unchanged)"""
    os.environ.setdefault("CUDA_VISIBLE_DEVICES", "0")  # gpu_utils.get_available_gpus reads it
    E = int(argv[0]) if len(argv) > 0 else 4096
    t_budget = float(argv[1]) if len(argv) > 1 else 3.0
    T, nb = 32, 4
    obs_space = gym.spaces.Dict({"obs": gym.spaces.Box(0, 255, (4, 84, 84), np.uint8)})
    cfg = make_cfg(C2_MODEL_ARGS + [f"--rollout={T}", f"--batch_size={E * T // nb}", f"--num_batches_per_epoch={nb}",
                                    "--num_epochs=1", "--exploration_loss_coeff=0.01", "--device=gpu"])
    assert cfg.device == "gpu"
    t_setup = time.perf_counter()
    learner, env_info = make_learner(cfg, obs_space, gym.spaces.Discrete(6), E)
    ac = learner.actor_critic
    dev = learner.device
    assert dev.type == "cuda", dev
    g = torch.Generator().manual_seed(0)
    b = alloc_trajectory_tensors(env_info, E, T, get_rnn_size(cfg), "cpu", False)
    fill_batch(b, g, 6, p_done=0.01)
    from sample_factory.algo.utils.tensor_dict import TensorDict

    def to_dev(td):
        return TensorDict({k: to_dev(v) if isinstance(v, dict) else v.to(dev) for k, v in td.items()})

    b = to_dev(b)
    obs0 = {"obs": b["obs"]["obs"][:, 0].clone()}
    rnn0 = b["rnn_states"][:, 0].clone()
    from sample_factory.algo.utils.tensor_dict import clone_tensordict

    def infer():
        with torch.no_grad():
            ac(prepare_and_normalize_obs(ac, obs0), rnn0)

    # ---- untimed: one inference step + one Learner.train at the full shapes (MIOpen find / hipBLASLt heuristics)
    ac.eval()
    infer()
    ac.train()
    learner.train(clone_tensordict(b))
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup
    # ---- timed
    ac.eval()
    infer()
    torch.cuda.synchronize()
    t0, reps = time.perf_counter(), 0
    while time.perf_counter() - t0 < t_budget:
        infer()
        torch.cuda.synchronize()  # the reference's inference worker synchronises every step (inference_worker.py:337)
        reps += 1
    t_inf = (time.perf_counter() - t0) / reps
    ac.train()
    trains = []
    for _ in range(2):
        bb = clone_tensordict(b)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        learner.train(bb)
        torch.cuda.synchronize()
        trains.append(time.perf_counter() - t1)
    t_train = min(trains)
    t_iter = T * t_inf + t_train
    print(json.dumps(dict(
        workload="reference_torch_rocm", value=round(E * T / t_iter, 1), unit="env-steps/s", kind="reference",
        device=torch.cuda.get_device_name(0), ms_per_step=round(t_iter * 1e3, 2),
        sample=f"the reference's ActorCritic.forward on {E} obs ({t_inf * 1e3:.2f} ms per step, {reps} reps) x {T} steps + "
               f"Learner.train on the {E}x{T} dataset in {nb} minibatches ({t_train * 1e3:.1f} ms, best of 2), torch "
               f"{torch.__version__} on {dev} (stock PyTorch-ROCm / MIOpen, fp32); env excluded; one untimed warm-up "
               f"iteration ({t_setup:.1f} s incl. MIOpen search)",
        t_inference_ms_per_step=round(t_inf * 1e3, 3), t_train_ms=round(t_train * 1e3, 2), warmup_s=round(t_setup, 1),
        miopen_find_mode=os.environ.get("MIOPEN_FIND_MODE", "default"), reference_from=ref_import.REFERENCE_ROOT)))


def main():
    if "--device" in sys.argv:
        i = sys.argv.index("--device")
        dev = sys.argv[i + 1]
        rest = sys.argv[1:i] + sys.argv[i + 2:]
        if dev == "cuda":
            return main_cuda(rest)
        sys.argv = sys.argv[:1] + rest
    cores = len(os.sched_getaffinity(0))
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 128   # trajectories in the timed sample (4096 = the full Tier-B workload)
    t_budget = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
    threads_arg = sys.argv[3] if len(sys.argv) > 3 else "auto"
    repeats = int(sys.argv[4]) if len(sys.argv) > 4 else 1   # timed Learner.train calls (after one untimed call)
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
            # keep the fastest for BOTH legs — the baseline is the reference at ITS best thread count on this box.  The scan
            # runs on the forward that is timed below (the best pool size depends on the batch: 16 threads were best for 256
            # observations and 20 x slower per observation on 1024).
            for th in sorted({t for t in (8, 16, 32, 64, 128, cores) if t <= cores}):
                torch.set_num_threads(th)
                ac(prepare_and_normalize_obs(ac, obs0), rnn0)
                t0, reps = time.perf_counter(), 0
                while (time.perf_counter() - t0 < 0.6 or reps < 2) and reps < 50:
                    ac(prepare_and_normalize_obs(ac, obs0), rnn0)
                    reps += 1
                    if time.perf_counter() - t0 > 6.0:
                        break
                scan[th] = (time.perf_counter() - t0) / reps
            torch.set_num_threads(min(scan, key=scan.get))
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < t_budget or reps < 3:
            ac(prepare_and_normalize_obs(ac, obs0), rnn0)
            reps += 1
        t_inf = (time.perf_counter() - t0) / reps
    # ---- Learner.train on the E x T dataset: one UNTIMED call, then `repeats` timed ones, every call on a fresh copy of the
    # dataset (_prepare_batch mutates its input: learner.py:953,990).  The first call on a learner carries one-off costs
    # (autograd / oneDNN primitive set-up for these shapes, thread-pool start, first touch of the allocator's pages: 2.4 s
    # against 0.6 s per call on a 64-trajectory sample) which a single timed call used to carry into the extrapolation.
    from sample_factory.algo.utils.tensor_dict import TensorDict

    def fresh(td):
        return TensorDict({k: fresh(v) if isinstance(v, dict) else v.clone() for k, v in td.items()})

    ac.train()
    trains = []
    for rep in range(repeats + 1):
        bb = fresh(b) if rep < repeats else b
        t1 = time.perf_counter()
        learner.train(bb)
        if rep > 0:
            trains.append(time.perf_counter() - t1)
        del bb
    scale = full_envs / E
    srt = sorted(trains)
    t_train = srt[len(srt) // 2] if len(srt) % 2 else (srt[len(srt) // 2 - 1] + srt[len(srt) // 2]) / 2
    rates = [full_envs * T / (T * t_inf * scale + t * scale) for t in trains]
    value = full_envs * T / (T * t_inf * scale + t_train * scale)
    extrap = "nothing extrapolated" if scale == 1 else f"extrapolated x{scale:.0f} to one {full_envs}x{T} iteration"
    print(json.dumps(dict(
        value=round(value, 1), unit="env-steps/s", cores=torch.get_num_threads(), host_cores=cores,
        kind="reference", thread_scan_ms_per_forward={str(k): round(v * 1e3, 1) for k, v in scan.items()},
        sample=f"reference ActorCritic.forward on {E} obs ({t_inf * 1e3:.1f} ms, {reps} reps) and Learner.train on a "
               f"{E}x{T} dataset in {nb} minibatches ({', '.join(f'{t:.2f}' for t in trains)} s after one untimed call; value "
               f"uses the median), torch {torch.__version__} CPU fp32, {torch.get_num_threads()} threads (best of the pool "
               f"sizes tried) of {cores} host cores; {extrap} (env excluded)",
        sample_envs=E, extrapolation_factor=scale, repeats=[round(r, 1) for r in rates],
        spread_pct=round(100.0 * (max(rates) - min(rates)) / value, 1) if len(rates) > 1 else None,
        t_inference_ms_per_step_sample=round(t_inf * 1e3, 2), t_train_s_sample=round(t_train, 3),
        t_train_s_repeats=[round(t, 3) for t in trains],
        reference_from=ref_import.REFERENCE_ROOT,
        where="build container" if os.path.isdir("/root/reference") else "GPU box host cores")))


if __name__ == "__main__":
    main()

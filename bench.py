"""bench.py — env-steps/sec of the APPO hot path on the BASELINE.json workload.

A "step" is one full pass of the hot path over one batch of synthetic input: a rollout of `rollout` env steps for all
`envs` device-resident synthetic envs (policy inference + sampling + env + trajectory writes) followed by Learner.train
on that dataset (bootstrap values, GAE, returns normaliser, `num_batches` minibatches of forward / PPO loss / backward
/ clip / Adam).  Workload = BASELINE.json configs[1] ("synthetic vector env 4096 envs, 84x84x4 uint8 obs, discrete(6),
Nature-CNN actor-critic, 1xMI355X") with the NS-2 learner preset of SURVEY.md §8d.  With --gpus N (launched by
torch.distributed.run, one rank per GPU) every rank runs the same per-GPU workload on its own env shard and gradients /
advantage moments are all-reduced over RCCL: weak scaling, value = whole-job env-steps/s.

  python bench.py [--gpus N] [--steps K] [--warmup W]
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
PEAK_HBM_GBS = 8000.0


def kernel_flops(key):
    """algorithmic FLOPs of one launch of a network kernel (2*M*N*K of the implicit GEMM)"""
    op, n, Cin, H, W, Cout, K, S, OH, OW = key[:10]
    return 2.0 * n * OH * OW * Cout * (K * K * Cin)  # fwd, wgrad and (gather-form, no zero taps) dgrad are equal


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--rollout", type=int, default=32)
    ap.add_argument("--num_batches", type=int, default=4)
    ap.add_argument("--num_epochs", type=int, default=1)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_kernel_events", action="store_true", help="skip per-launch HIP events (no roofline object)")
    ap.add_argument("--cpu_baseline_envs", type=int, default=256)
    ap.add_argument("--env_instances", type=int, default=1,
                    help="split the envs of a GPU into this many vector-env instances (num_envs_per_worker = "
                         "worker_num_splits = this): their rollouts run on separate HIP streams")
    ap.add_argument("--async_rl", action="store_true", help="overlap rollout k+1 with train(k) (policy lag of one dataset)")
    args = ap.parse_args()

    import torch

    from sample_factory_amd import lib
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_synthetic_env
    from sample_factory_amd.train import make_runner

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} must be launched as `python -m torch.distributed.run --nproc-per-node "
                         f"{args.gpus} bench.py --gpus {args.gpus} ...` (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
        torch.distributed.init_process_group(os.environ.get("SF_DP_BACKEND", "nccl"))  # nccl = RCCL on ROCm (gloo: tests)

    register_env("synthetic_atari", make_synthetic_env)
    B, T = args.envs, args.rollout
    cfg = default_cfg(
        env="synthetic_atari", use_rnn=False, recurrence=1, encoder_conv_architecture="convnet_atari",
        nonlinearity="relu", encoder_conv_mlp_layers=[512], obs_scale=255.0, normalize_input=False,
        normalize_returns=True, rollout=T, batch_size=B * T // args.num_batches, num_batches_per_epoch=args.num_batches,
        num_epochs=args.num_epochs, gamma=0.99, gae_lambda=0.95, ppo_clip_ratio=0.1, ppo_clip_value=1.0,
        value_loss_coeff=0.5, exploration_loss_coeff=0.01, max_grad_norm=4.0, learning_rate=1e-4, adam_eps=1e-6,
        async_rl=args.async_rl, serial_mode=not args.async_rl, batched_sampling=True, num_workers=1,
        num_envs_per_worker=args.env_instances, worker_num_splits=args.env_instances, env_gpu_observations=True, env_gpu_actions=True, actor_worker_gpus=[0], seed=0,
        synthetic_num_agents=B // args.env_instances, synthetic_env0=rank * B, data_parallel=world > 1)
    cfg, runner = make_runner(cfg)
    runner.init()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    # warm-up.  The last warm-up step is instrumented per launch (HIP events on the launch stream) to rank the network
    # kernels; inside the TIMED region only the dominant kernel keeps its events (timing every launch costs ~7 %).
    for _ in range(max(0, args.warmup - 1)):
        runner.iteration()
    warm_prof = None
    if args.warmup > 0:
        if not args.no_kernel_events:
            lib.PROFILE, lib.PROFILE_SYNC = {}, True  # ranking pass: every launch timed in isolation
        runner.iteration()
        torch.cuda.synchronize()
        warm_prof, lib.PROFILE, lib.PROFILE_SYNC = lib.PROFILE, None, False
    if not args.no_kernel_events:
        if not warm_prof:
            raise SystemExit("need --warmup >= 1 to rank kernels (or pass --no_kernel_events)")
        # "kernel" = one instantiation (key[-1], the name rocprofv3 prints), summed over the shapes it is launched at
        # robust total per shape = median launch time x launches: a single launch that happens to bracket a one-off
        # host-side stall (allocator growth, lazy code-object load) must not decide which kernel is "dominant"
        def robust_total(evs):
            ms = sorted(s_.elapsed_time(e_) for s_, e_ in evs)
            return ms[len(ms) // 2] * len(ms)

        by_name = {}
        for key, evs in warm_prof.items():
            by_name[key[-1]] = by_name.get(key[-1], 0.0) + robust_total(evs)
        dominant = max(by_name, key=by_name.get)
        lib.PROFILE, lib.PROFILE_ONLY = {}, {key for key in warm_prof if key[-1] == dominant}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.iteration()
    barrier()
    dt = time.perf_counter() - t0
    prof, lib.PROFILE, lib.PROFILE_ONLY = lib.PROFILE, None, None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    env_steps = args.steps * B * T * world
    value = env_steps / dt

    if not prof:
        if rank == 0:
            print(json.dumps({"value": round(value, 1), "ms_per_step": round(dt / args.steps * 1e3, 2), "n_gpus": world}))
        return
    # ---- roofline of the dominant kernel instantiation (largest total time; HIP events over the timed region).
    # achieved = mean algorithmic FLOPs per launch / mean launch duration = sum(flops) / sum(duration) over its launches;
    # avg_launch_ms is directly comparable with AverageNs of the same name in profiles/r01_*_kernel_stats.csv
    total_ms, launches, flops, shapes = 0.0, 0, 0.0, []
    for key, evs in prof.items():
        ms = [s_.elapsed_time(e_) for s_, e_ in evs]
        total_ms += sum(ms)
        launches += len(ms)
        flops += kernel_flops(key) * len(ms)
        shapes.append({"shape": f"{key[0]}:{key[2]}x{key[3]}->{key[5]} n={key[1]}", "launches": len(ms),
                       "avg_ms": round(sum(ms) / len(ms), 4),
                       "tflops": round(kernel_flops(key) / (sum(ms) / len(ms) * 1e-3) / 1e12, 1)})
    shapes.sort(key=lambda d: -d["avg_ms"] * d["launches"])
    avg_ms = total_ms / launches
    achieved = flops / (total_ms * 1e-3) / 1e12
    traffic, traffic_src = None, None
    tj = os.path.join(ROOT, "profiles", "r01_traffic.json")  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench
    if os.path.exists(tj):
        ent = json.load(open(tj)).get(dominant)
        if ent:
            traffic, traffic_src = ent["hbm_bytes"], "profiles/r01_traffic.json (2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes)"
    kern = []  # ranking of all network kernels from the instrumented warm-up step (NOT the timed region)
    for k2, evs2 in warm_prof.items():
        m2 = sorted(s_.elapsed_time(e_) for s_, e_ in evs2)
        med = m2[len(m2) // 2]
        kern.append((med * len(m2), k2, len(m2), med))
    kern.sort(reverse=True, key=lambda k: k[0])
    roofline = {"bound": "mfma", "kernel": dominant, "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                "traffic_source": traffic_src, "avg_launch_ms": round(avg_ms, 4), "launches": launches,
                "gflop_per_launch": round(flops / launches / 1e9, 3),
                "share_of_step_time": round(total_ms / (dt * 1e3), 4), "shapes": shapes}
    net_ms = sum(k[0] for k in kern)
    net_flops = sum(kernel_flops(k[1]) * k[2] for k in kern)
    breakdown = [{"kernel": f"{k[1][0]}:{k[1][2]}x{k[1][3]}->{k[1][5]} n={k[1][1]}", "name": k[1][-1],
                  "ms_total": round(k[0], 2), "launches": k[2],
                  "tflops": round(kernel_flops(k[1]) / (k[3] * 1e-3) / 1e12, 1)} for k in kern[:12]]

    out = {
        "metric": "env-steps/sec (whole node), 4096 envs, 84x84x4 obs", "value": round(value, 1), "unit": "env-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"BASELINE.json configs[1]: synthetic vector env {B} envs/GPU, 84x84x4 u8 obs, "
                               f"Discrete(6), Nature-CNN actor-critic (1,687,719 params), APPO "
                               f"{'async (rollout k+1 || train k)' if args.async_rl else 'sync'}, rollout={T}, "
                               f"batch_size={cfg.batch_size} x {args.num_batches} minibatches x {args.num_epochs} epoch(s)",
                   "envs_per_gpu": B, "rollout": T, "batch_size": cfg.batch_size, "num_batches_per_epoch": args.num_batches,
                   "num_epochs": args.num_epochs, "parallelism": f"dp{world}"},
        "roofline": roofline,
        "network_kernels": {"source": "instrumented warm-up step (every launch timed in isolation; not the timed region)",
                            "ms_per_step": round(net_ms, 2), "tflops_avg": round(net_flops / (net_ms * 1e-3) / 1e12, 2),
                            "top": breakdown},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cpu_baseline  # checker/baseline leg only; never on the measured path
        out["cpu_baseline"] = cpu_baseline.run(num_envs=args.cpu_baseline_envs, rollout=T, num_minibatches=args.num_batches)
        out["cpu_baseline"]["value"] = round(out["cpu_baseline"]["value"], 1)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

"""bench.py — env-steps/sec of the APPO hot path on the BASELINE.json workload.

A "step" is one full pass of the hot path over one batch of synthetic input: a rollout of `rollout` env steps for all
`envs` device-resident synthetic envs (policy inference + sampling + env + trajectory writes) followed by Learner.train
on that dataset (bootstrap values, GAE, returns normaliser, `num_batches` minibatches of forward / PPO loss / backward
/ clip / Adam).  Workload = BASELINE.json configs[1] ("synthetic vector env 4096 envs, 84x84x4 uint8 obs, discrete(6),
Nature-CNN actor-critic, 1xMI355X") with the NS-2 learner preset of SURVEY.md §8d.  With --gpus N (launched by
torch.distributed.run, one rank per GPU) every rank runs the same per-GPU workload on its own env shard and gradients /
advantage moments are all-reduced over RCCL: weak scaling, value = whole-job env-steps/s.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c5]

`--workload c5` = BASELINE.json configs[4] on one GPU per rank: Ant-shaped synthetic continuous env (obs f32[27],
Box(8)), 2048 envs, MLP[64,64] tanh encoder + LSTM-512 core, V-trace, KL loss, rollout = recurrence = 32.

`--gpus N` without a torchrun environment re-launches itself as `python -m torch.distributed.run --nproc-per-node N`
(one rank per GPU, RCCL); every rank checks it has its own device and the JSON line carries `rccl_ranks` (the result of
a rank-stamped all-reduce) and the device of every rank.
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


class ClockSampler:
    """Shader clock of this rank's GPU DURING the timed region: the roofline peak is priced at the 2.4 GHz maximum, the chip
    runs at its power budget (MI355X_MICROARCH.md, DVFS) — `roofline.clock_ghz` / `frac_at_measured_clock` say how much of the
    gap is clock.  Source: `sf_clock_probe`, a one-wave kernel that spins for 200 k shader cycles (~85 us) and counts the
    ticks of the constant 100 MHz wall clock meanwhile.  The probes are enqueued on a side stream during the plain warm-up
    steps right before the timed region (same steady-state load; see the call site for why not inside it) and run beside
    whatever the measured stream is executing; read back after the region.
    (amdgpu's pp_dpm_sclk reported 95 MHz and 2.39 GHz for the same workload on two boxes — not used.)"""

    MAX = 32

    def __init__(self):
        import torch
        self.side = torch.cuda.Stream()
        self.out = torch.zeros((self.MAX, 2), dtype=torch.int64, device="cuda")
        self.n = 0

    def probe(self):
        if self.n < self.MAX:
            from sample_factory_amd import lib
            lib.clock_probe(self.out[self.n], 200000, on_stream=self.side)
            self.n += 1

    def stop(self):
        self.side.synchronize()
        rows = self.out[:self.n].tolist()
        good = sorted(0.1 * c / w for c, w in rows if w > 0 and 0.3 < 0.1 * c / w < 3.5)
        if not good:
            return None
        return {"ghz": round(good[len(good) // 2], 3), "min_ghz": round(good[0], 3), "max_ghz": round(good[-1], 3),
                "samples": len(good), "source": "sf_clock_probe (shader cycles per 100 MHz wall-clock tick, side stream)"}


def kernel_flops(key):
    """algorithmic FLOPs of one launch of a network kernel (2*M*N*K of the implicit GEMM)"""
    op, n, Cin, H, W, Cout, K, S, OH, OW = key[:10]
    return 2.0 * n * OH * OW * Cout * (K * K * Cin)  # fwd, wgrad and (gather-form, no zero taps) dgrad are equal


PEAK_HBM_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak


def exact_bf16_kernel(name: str) -> bool:
    """conv1 on raw u8 frames runs on the bf16 matrix pipe with EXACT products (csrc/sf_nn_u8.h): those launches are
    bounded by HBM traffic, not by the f32-MFMA rate the other network kernels are priced against"""
    return name.startswith("k_conv1_u8_bf16") or name.startswith("k_conv1_wgrad_bf16")


def kernel_bytes(key):
    """algorithmic HBM bytes of one launch of a conv1 kernel: the u8 frames once + the f32 output (forward) or output
    gradient (weight gradient) once"""
    op, n, Cin, H, W, Cout, K, S, OH, OW = key[:10]
    return float(n) * (Cin * H * W + OH * OW * Cout * 4)


def cpu_baseline_leg(args, T):
    """AFTER the timed region, rank 0, N = 1: the reference's CPU path timed on this box's host cores.
    kind "reference" = the reference's own ActorCritic.forward + Learner.train (SURVEY.md 8(d) Tier B) executed from the
    archive `make -C oracle ref` staged (oracle/_ref/, or /root/reference where that exists), in a process of its own
    (its `sample_factory` package must not meet this repo's alias package); kind "port" (torch-CPU network + C oracle,
    oracle/cpu_baseline.py) only when no reference is available or the reference run fails."""
    import subprocess
    from oracle import ref_import_path  # checker/baseline leg only; never on the measured path
    err = None
    if ref_import_path.reference_available():
        try:
            # sample: 1024 of the 4096 trajectories (x4 to the full Tier-B iteration; the full 4096 x 32 dataset is ~30 s per
            # Learner.train call and ~45 GB of the reference's f32 frame copies: --cpu_reference_envs 4096 runs it where the
            # host allows), one untimed + three timed Learner.train calls, value from the median, spread in the line
            r = subprocess.run([sys.executable, "-m", "oracle.ref_cpu_tier_b", str(args.cpu_reference_envs), "4", "auto",
                                str(args.cpu_reference_repeats)],
                               cwd=ROOT, capture_output=True, text=True, timeout=420)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode == 0 and line:
                return json.loads(line[-1])
            err = f"rc {r.returncode}: {r.stderr[-300:]}"
        except Exception as e:  # noqa: BLE001 - the baseline leg must never take the bench line down
            err = repr(e)
    from oracle import cpu_baseline
    res = cpu_baseline.run(num_envs=args.cpu_baseline_envs, rollout=T, num_minibatches=args.num_batches)
    res["value"] = round(res["value"], 1)
    if err:
        res["reference_run_failed"] = err
    return res


def secondary_lines(args):
    """AFTER the C2 timed region (rank 0, N = 1): the other single-GPU BASELINE configurations this engine runs —
    configs[4] (c5: LSTM-512 + V-trace + Box(8), 2048 envs) and the configs[2] stand-in (c3: 1024 HOST envs ingested over
    PCIe, env side unpinned: envpool is not installable here) — each as a process of its own running this very script,
    summarised into the headline JSON line so that the driver observes them too.  Same contract per entry: warm-up, K
    timed steps bracketed by synchronize, whole-job env-steps/s, roofline of that run's dominant kernel."""
    import subprocess
    out = []
    # c2_atari_preset = SURVEY.md 8(d)'s second measurement point: the C2 workload at the learner settings of the reference's
    # envpool-Atari preset (rollout=128, num_epochs=4, num_batches_per_epoch=4 per dataset of the preset -> here 16 minibatches
    # of 32768 over the 4096 x 128 dataset; /root/reference/sf_examples/envpool/atari/envpool_atari_params.py:31-34)
    extra = {"c2_normalize_input": ["--normalize_input"],
             "c2_atari_preset": ["--rollout", "128", "--num_batches", "16", "--num_epochs", "4"]}
    for wl, steps, warm in (("c5", 16, 3), ("c3", 12, 3), ("c2_normalize_input", 12, 2), ("c2_atari_preset", 3, 2)):
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", wl.split("_")[0], "--steps", str(steps), "--warmup",
               str(warm), "--no_cpu_baseline", "--no_secondary"] + extra.get(wl, [])
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=200)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not line:
                out.append({"workload": wl, "error": f"rc {r.returncode}: {r.stderr[-200:]}"})
                continue
            d = json.loads(line[-1])
            rf = d.get("roofline", {})
            ent = {"workload": wl, "metric": d["metric"], "value": d["value"], "unit": d["unit"],
                   "ms_per_step": d["ms_per_step"], "steps": d["steps"], "warmup": d["warmup"], "dtype": d["dtype"],
                   "data": "synthetic", "config": d["config"]["workload"],
                   "rollout_launch_programs": d.get("rollout_launch_programs"),
                   "roofline": {k: rf.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic",
                                                       "traffic_source", "clock_ghz", "frac_at_measured_clock",
                                                       "avg_launch_ms", "launches", "share_of_step_time", "scope")},
                   "wall_s": round(time.perf_counter() - t0, 1)}
            if "ingest" in d:
                ent["ingest"] = {k: d["ingest"].get(k) for k in ("h2d_bytes_per_env_step", "obs_dma_gbs", "sampler_thread",
                                                                  "dma_share_of_wall_clock", "env_worker_processes",
                                                                  "obs_dma_from_worker_pages_in_place", "host_s")}
                ent["env_side"] = "unpinned (synthetic host frames; envpool/ALE not installable)"
            if wl == "c5":
                ent["env_side"] = "unpinned (Ant-shaped synthetic env; envpool/mujoco not installable)"
            out.append(ent)
        except Exception as e:  # noqa: BLE001 - a secondary line must never take the headline line down
            out.append({"workload": wl, "error": repr(e)})
    out.append(reference_on_this_gpu())
    return out


def reference_on_this_gpu():
    """CONTEXT, not credit: the reference's own ActorCritic.forward + Learner.train through stock PyTorch-ROCm / MIOpen on
    the same MI355X (oracle/ref_cpu_tier_b.py --device cuda, a process of its own, after the timed region), full workload
    size — says whether the hand-written path beats the stock stack on this box, next to the CPU baseline."""
    import subprocess
    from oracle import ref_import_path  # baseline leg only; never on the measured path
    if not ref_import_path.reference_available():
        return {"workload": "reference_torch_rocm", "error": "no reference archive (make -C oracle ref)"}
    t0 = time.perf_counter()
    try:
        # MIOpen's kernel search for the reference's 18 fp32 convolution problems takes ~3 minutes on a fresh box; its
        # result (the user find-db written by such a run on an MI355X) is committed under oracle/miopen_db and handed to
        # the subprocess through a scratch copy, so the reference runs on the kernels MIOpen itself picked, in seconds
        import shutil
        import tempfile
        env = dict(os.environ)
        src_db = os.path.join(ROOT, "oracle", "miopen_db")
        if os.path.isdir(src_db) and "MIOPEN_USER_DB_PATH" not in env:
            tmp_db = tempfile.mkdtemp(prefix="sf_miopen_db_")
            for f in os.listdir(src_db):
                shutil.copy(os.path.join(src_db, f), tmp_db)
            env["MIOPEN_USER_DB_PATH"] = tmp_db
        r = subprocess.run([sys.executable, "-m", "oracle.ref_cpu_tier_b", "--device", "cuda", "4096", "2"], cwd=ROOT,
                           capture_output=True, text=True, timeout=300, env=env)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if r.returncode != 0 or not line:
            return {"workload": "reference_torch_rocm", "error": f"rc {r.returncode}: {r.stderr[-300:]}"}
        d = json.loads(line[-1])
        d["wall_s"] = round(time.perf_counter() - t0, 1)
        return d
    except Exception as e:  # noqa: BLE001
        return {"workload": "reference_torch_rocm", "error": repr(e)}


def _free_port() -> int:
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` outside torchrun: become the launcher (one rank per GPU on this node)."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL across processes)
    return subprocess.call(cmd, env=env)


def _cpulist(text: str):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def numa_plan(local_rank: int, device_nodes, allowed, node_cpus, core_of=None):
    """cores of `local_rank`: the cores of its GPU's NUMA node that this process may use, split evenly (in local-rank order)
    among the ranks whose GPUs sit on the same node.  device_nodes[r] = NUMA node of rank r's GPU (-1 / None: unknown ->
    no pinning for that rank), node_cpus[node] = cpu ids, core_of[cpu] = physical core of a logical cpu (SMT siblings stay
    with one rank: a node lists them as "0-63,128-191", and a split of the plain order would hand one rank the first threads
    and its neighbour the second threads of the SAME cores).  Pure function (tests/test_abi_and_host.py)."""
    node = device_nodes[local_rank]
    if node is None or node < 0 or node not in node_cpus:
        return None
    cpus = sorted(set(node_cpus[node]) & set(allowed), key=(lambda c: (core_of.get(c, c), c)) if core_of else None)
    peers = [r for r, nd in enumerate(device_nodes) if nd == node]
    k, m = peers.index(local_rank), len(peers)
    share = cpus[k * len(cpus) // m:(k + 1) * len(cpus) // m]
    return sorted(share) or None


def pin_rank_to_gpu_numa(local_rank: int, local_world: int, sysfs: str = "/sys"):
    """One process per GPU: keep the rank — and the env worker processes it starts later, which inherit the mask — on the
    cores of the NUMA node its GPU hangs off (H2D of host-env frames and the launch path stay off the inter-socket link).
    Returns {"numa_node", "cpus"} or a reason why nothing was pinned; never fails the run."""
    try:
        import torch
        ndev = torch.cuda.device_count()
        nodes = []
        for r in range(local_world):
            pr = torch.cuda.get_device_properties(r % ndev)
            bus = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id"), getattr(pr, "pci_bus_id"), getattr(pr, "pci_device_id"))
            try:
                nodes.append(int(open(f"{sysfs}/bus/pci/devices/{bus}/numa_node").read()))
            except OSError:
                nodes.append(None)
        node_cpus = {}
        for nd in {n for n in nodes if n is not None and n >= 0}:
            node_cpus[nd] = _cpulist(open(f"{sysfs}/devices/system/node/node{nd}/cpulist").read())
        core_of = {}
        for cpus_ in node_cpus.values():
            for c_ in cpus_:
                try:
                    t_ = f"{sysfs}/devices/system/cpu/cpu{c_}/topology/"
                    core_of[c_] = (int(open(t_ + "physical_package_id").read()), int(open(t_ + "core_id").read()))
                except (OSError, ValueError):
                    pass
        share = numa_plan(local_rank, nodes, os.sched_getaffinity(0), node_cpus, core_of or None)
        if not share:
            return {"numa_node": nodes[local_rank], "cpus": None, "note": "GPU's NUMA node unknown: affinity left as it was"}
        os.sched_setaffinity(0, share)
        return {"numa_node": nodes[local_rank], "cpus": len(share), "first_cpu": share[0]}
    except Exception as e:  # noqa: BLE001 - placement is an optimisation
        return {"numa_node": None, "cpus": None, "note": repr(e)}


def init_replicas(world: int, rank: int, need_gpu: bool = True):
    """process group + a self-check of the collective path: all-reduce of a rank-stamped tensor must give
    sum(2^r) = 2^world - 1 (every rank contributed exactly once) and the ranks must sit on distinct devices.
    Returns (rccl_ranks, [device index of every rank], backend)."""
    import torch
    import torch.distributed as dist
    backend = os.environ.get("SF_DP_BACKEND", "nccl")  # nccl = RCCL on ROCm (gloo: CPU tests)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if backend == "nccl" and need_gpu and ndev < int(os.environ.get("LOCAL_WORLD_SIZE", world)):
        raise SystemExit(f"[rank {rank}] bench.py --gpus {world}: need {world} devices on this node, found {ndev} "
                         f"(one rank per GPU; SF_DP_BACKEND=gloo shares a device for protocol tests)")
    if ndev:
        torch.cuda.set_device(local % ndev)
    dist.init_process_group(backend)
    dev = torch.device("cuda", torch.cuda.current_device()) if (ndev and backend == "nccl") else torch.device("cpu")
    stamp = torch.tensor([float(2 ** rank), float(local % ndev if ndev else -1)], dtype=torch.float64, device=dev)
    devs = [torch.zeros_like(stamp) for _ in range(world)]
    dist.all_gather(devs, stamp)
    tot = stamp[:1].clone()
    dist.all_reduce(tot)
    if int(tot.item()) != 2 ** world - 1:
        raise SystemExit(f"[rank {rank}] collective self-check failed: {tot.item()} != {2 ** world - 1}")
    dev_ids = [int(d[1].item()) for d in devs]
    if backend == "nccl" and len(set(dev_ids)) != world:
        raise SystemExit(f"[rank {rank}] ranks share devices {dev_ids}: one rank per GPU expected")
    return world, dev_ids, backend


def workload_cfg(args, rank, world):
    """(cfg, env name, env factory, description, metric) of the selected BASELINE.json configuration"""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.synthetic import make_synthetic_continuous_env, make_synthetic_env
    B, T = args.envs, args.rollout
    common = dict(rollout=T, batch_size=B * T // args.num_batches, num_batches_per_epoch=args.num_batches,
                  num_epochs=args.num_epochs, async_rl=args.async_rl, serial_mode=not args.async_rl, batched_sampling=True,
                  num_workers=1, num_envs_per_worker=args.env_instances, worker_num_splits=args.env_instances,
                  env_gpu_observations=True, env_gpu_actions=True, actor_worker_gpus=[0], seed=0,
                  synthetic_num_agents=B // args.env_instances, data_parallel=world > 1,
                  # SF_DP_NATIVE=1: gradient buckets through the C-ABI (sf_allreduce_grads, one RCCL communicator per rank)
                  # instead of torch.distributed — both exchange paths can be measured by the same --gpus N run
                  dp_native_rccl=os.environ.get("SF_DP_NATIVE", "0") not in ("", "0"),
                  # SF_DP_ONESHOT=<bytes> (1 = 1 MiB): small SUM buckets through the one-shot mailbox exchange (sf_dp_oneshot_*)
                  dp_oneshot_bytes=(lambda v: (1 << 20) if v == 1 else v)(int(os.environ.get("SF_DP_ONESHOT", "0") or 0)))
    mode = "async (rollout k+1 || train k)" if args.async_rl else "sync"
    if args.workload == "c2":
        cfg = default_cfg(
            env="synthetic_atari", use_rnn=False, recurrence=1, encoder_conv_architecture="convnet_atari",
            nonlinearity="relu", encoder_conv_mlp_layers=[512], obs_scale=255.0,
            normalize_input=bool(args.normalize_input), normalize_returns=True, gamma=0.99, gae_lambda=0.95, ppo_clip_ratio=0.1, ppo_clip_value=1.0,
            value_loss_coeff=0.5, exploration_loss_coeff=0.01, max_grad_norm=4.0, learning_rate=1e-4, adam_eps=1e-6,
            synthetic_env0=rank * B, **common)
        desc = (f"BASELINE.json configs[1]: synthetic vector env {B} envs/GPU, 84x84x4 u8 obs, Discrete(6), Nature-CNN "
                f"actor-critic (1,687,719 params), APPO {mode}, rollout={T}, batch_size={cfg.batch_size} x "
                f"{args.num_batches} minibatches x {args.num_epochs} epoch(s)")
        if args.normalize_input:
            desc += (", normalize_input=True (the reference's default: per-pixel running mean / std of the frames, moments "
                     "in one pass over the u8 slab, normalisation inside conv1's loader — no f32 copy of the frames)")
        return cfg, "synthetic_atari", make_synthetic_env, desc, "env-steps/sec (whole node), 4096 envs, 84x84x4 obs"
    if args.workload == "c5":  # sf_examples/mujoco/mujoco_params.py:1-38 + LSTM core + V-trace
        cfg = default_cfg(
            env="synthetic_ant", use_rnn=True, rnn_type=args.rnn_type, rnn_size=512, recurrence=T, encoder_mlp_layers=[64, 64],
            nonlinearity="tanh", normalize_input=True, normalize_returns=False, with_vtrace=True, kl_loss_coeff=0.1,
            adaptive_stddev=False, policy_initialization="torch_default", value_bootstrap=True, max_grad_norm=3.5,
            ppo_clip_ratio=0.2, value_loss_coeff=1.3, exploration_loss_coeff=0.0, learning_rate=0.00295, gamma=0.99,
            gae_lambda=0.95, **common)
        cfg.seed = rank  # torch-generator env: a different stream per replica
        desc = (f"BASELINE.json configs[4]: Ant-shaped synthetic continuous env {B} envs/GPU (obs f32[27], Box(8)), "
                f"MLP[64,64] tanh + {args.rnn_type.upper()}-512 core + learned stddev, V-trace + KL loss + value bootstrap, APPO {mode}, "
                f"rollout=recurrence={T}, batch_size={cfg.batch_size} x {args.num_batches} minibatches x "
                f"{args.num_epochs} epoch(s)")
        return cfg, "synthetic_ant", make_synthetic_continuous_env, desc, \
            f"env-steps/sec (whole node), 2048 envs, obs f32[27], Box(8), {args.rnn_type.upper()}-512 + V-trace"
    if args.workload == "c3":  # sf_examples/envpool/atari/envpool_atari_params.py:26-45 on a HOST vector env
        from sample_factory_amd.envs.synthetic import make_host_frame_env
        cfg = default_cfg(
            env="host_atari", use_rnn=False, recurrence=1, encoder_conv_architecture="convnet_atari", nonlinearity="relu",
            encoder_conv_mlp_layers=[512], obs_scale=255.0, normalize_input=False, normalize_returns=False, gamma=0.99,
            gae_lambda=0.95, ppo_clip_ratio=0.1, ppo_clip_value=1.0, value_loss_coeff=0.5, exploration_loss_coeff=0.01,
            max_grad_norm=0.5, learning_rate=0.00025, adam_eps=1e-5, host_env_sim_ms=args.host_env_sim_ms, **common)
        cfg.seed = rank
        cfg.env_gpu_observations = cfg.env_gpu_actions = False
        if args.env_workers > 0:  # the reference's deployment for CPU envs: env worker processes, double-buffered sampling
            cfg.env_workers_mode, cfg.num_workers, cfg.num_envs_per_worker, cfg.worker_num_splits = "process", args.env_workers, 2, 2
            cfg.synthetic_num_agents = B // (args.env_workers * 2)
            assert cfg.synthetic_num_agents * args.env_workers * 2 == B, "--envs must be a multiple of 2 * --env_workers"
            how = (f"{args.env_workers} env worker processes x 2 instances x {cfg.synthetic_num_agents} agents, 2 sampling splits "
                   f"pipelined against inference; frames DMA'd into the slab from the workers' shared pages in place")
        else:
            cfg.env_workers_mode = "inline"
            how = f"{args.env_instances} instance(s) in the sampler thread -> pinned staging -> pitched H2D into the slab"
        desc = (f"BASELINE.json configs[2] stand-in: HOST vector env {B} envs/GPU ({how}; numpy u8 [4,84,84] frames as "
                f"envpool returns them, int32 "
                f"actions D2H), Discrete(6), Nature-CNN, APPO {mode}, rollout={T}, batch_size={cfg.batch_size} x "
                f"{args.num_batches} minibatches x {args.num_epochs} epoch(s); from the reference's envpool-Atari preset "
                f"(envpool_atari_params.py:26-45): async_rl, num_epochs=4, num_batches_per_epoch=4, lr 2.5e-4, adam_eps 1e-5, "
                f"max_grad_norm 0.5, no input / return normalisation; NOT from it: rollout {T} (preset 128) and batch_size "
                f"{cfg.batch_size} (preset 256) - one dataset = one rollout of all {B} envs here")
        return cfg, "host_atari", make_host_frame_env, desc, \
            "env-steps/sec (whole node), 1024 host envs, 84x84x4 obs ingested over PCIe"
    raise SystemExit(f"unknown workload {args.workload}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c2", choices=["c2", "c5", "c3"])
    ap.add_argument("--rnn_type", default="lstm", choices=["lstm", "gru"],
                    help="c5: core type (BASELINE configs[4] names LSTM; gru = the reference's default core at the same width)")
    ap.add_argument("--host_env_sim_ms", type=float, default=0.0, help="c3: simulated emulator time per env step")
    ap.add_argument("--check_launch", action="store_true",
                    help="stop after the replica-group self-check (launcher / env plumbing test; needs no GPU with "
                         "SF_DP_BACKEND=gloo)")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (default: 4096 for c2, 2048 for c5)")
    ap.add_argument("--rollout", type=int, default=32)
    ap.add_argument("--num_batches", type=int, default=4)
    ap.add_argument("--num_epochs", type=int, default=None, help="default: 1 for c2, 2 for c5 (mujoco preset)")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_secondary", action="store_true",
                    help="c2, one GPU: skip the secondary workload lines (c5, c3) measured after the timed region")
    ap.add_argument("--no_kernel_events", action="store_true", help="skip per-launch HIP events (no roofline object)")
    ap.add_argument("--cpu_baseline_envs", type=int, default=256)
    ap.add_argument("--cpu_reference_envs", type=int, default=1024,
                    help="trajectories of the sample the REFERENCE's CPU path is timed on (cpu_baseline kind 'reference'; "
                         "4096 = the full Tier-B workload, nothing extrapolated)")
    ap.add_argument("--cpu_reference_repeats", type=int, default=3,
                    help="timed Learner.train calls of the cpu_baseline leg (after one untimed call)")
    ap.add_argument("--event_stride", type=int, default=1,
                    help="timed region: every k-th launch of the dominant kernel carries a HIP-event pair (1 = all of them, the "
                         "default: a stride of 4 saved 0.2 ms per step but aliased with the 4 minibatches per step — every "
                         "sample was the same minibatch, 1.44 vs 1.53 ms for the n = 32768 shape; use a stride coprime to 4 "
                         "and 33 if at all)")
    ap.add_argument("--env_workers", type=int, default=4,
                    help="c3: env worker PROCESSES (algo/sampling/parallel_env.py); 0 = envs inside this process (sampler thread)")
    ap.add_argument("--env_instances", type=int, default=1,
                    help="split the envs of a GPU into this many vector-env instances (num_envs_per_worker = "
                         "worker_num_splits = this): their rollouts run on separate HIP streams")
    ap.add_argument("--sync_rl", action="store_true", help="c3: synchronous instead of the configuration's async mode")
    ap.add_argument("--one_instance", action="store_true", help="c3: a single env instance (no double-buffered sampling)")
    ap.add_argument("--async_rl", action="store_true", help="overlap rollout k+1 with train(k) (policy lag of one dataset)")
    ap.add_argument("--normalize_input", action="store_true",
                    help="c2 with cfg.normalize_input=True (the reference's default; BASELINE configs[1] / NS-2 runs with False)")
    args = ap.parse_args()
    if args.envs is None:
        args.envs = dict(c2=4096, c5=2048, c3=1024)[args.workload]
    if args.num_epochs is None:
        args.num_epochs = dict(c2=1, c5=2, c3=4)[args.workload]
    if args.workload == "c3":  # config 3 is "async APPO" with double-buffered sampling (worker_num_splits = 2)
        args.async_rl = not args.sync_rl
        if args.env_instances == 1 and not args.one_instance:
            args.env_instances = 2

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    import torch

    rccl_ranks, rank_devices, backend = 1, [0], None
    affinity = None
    if world > 1:
        rccl_ranks, rank_devices, backend = init_replicas(world, rank, need_gpu=not args.check_launch)
        if torch.cuda.is_available() and os.environ.get("SF_NUMA_PIN", "1") != "0":
            mine = pin_rank_to_gpu_numa(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", world)))
            allp = [None] * world
            torch.distributed.all_gather_object(allp, mine)
            affinity = allp
    if args.check_launch:
        if rank == 0:
            print(json.dumps({"launch_check": True, "rccl_ranks": rccl_ranks, "rank_devices": rank_devices,
                              "backend": backend, "n_gpus": world, "workload": args.workload}))
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")

    from sample_factory_amd import lib
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.train import make_runner

    cfg, env_name, env_factory, workload_desc, metric = workload_cfg(args, rank, world)
    register_env(env_name, env_factory)
    B, T = args.envs, args.rollout
    cfg, runner = make_runner(cfg)
    runner.init()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    # warm-up.  The last warm-up step is instrumented per launch (HIP events on the launch stream) to rank the network
    # kernels; inside the TIMED region only the dominant kernel keeps its events (timing every launch costs ~7 %).
    # Shader-clock probes ride on the PLAIN warm-up steps (same steady-state load as the timed region, immediately before
    # it): a probe is a kernel on another queue, and its launch / completion cache maintenance cost the L2-resident
    # hand-offs of the recurrent c5 loop 5 % when probes ran inside the timed region (17.3 vs 16.4 ms per step, same box).
    clk = ClockSampler() if (rank == 0 and os.environ.get("SF_CLOCK_PROBE", "1") != "0") else None
    for _ in range(max(0, args.warmup - 1)):
        runner.iteration()
        if clk is not None:
            for _p in range(4):
                clk.probe()  # enqueued back to back on the side stream: they spread over the next step's kernels
    if clk is not None and args.warmup < 2:
        clk = None  # no plain warm-up step to probe in: roofline.clock_ghz stays null rather than perturbing the timed region
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

        # c3 (host envs, async): the rollout side is hundreds of small per-split inference launches that share the chip with
        # the training stream -- the roofline of that run is taken on the TRAINING-side launches (n = the minibatch size)
        def in_scope(key):
            return args.workload != "c3" or key[1] == cfg.batch_size

        by_name = {}
        for key, evs in warm_prof.items():
            if in_scope(key):
                by_name[key[-1]] = by_name.get(key[-1], 0.0) + robust_total(evs)
        dominant = max(by_name, key=by_name.get)
        lib.PROFILE, lib.PROFILE_ONLY = {}, {key for key in warm_prof if key[-1] == dominant and in_scope(key)}
        # optional subsampling of the event pairs (--event_stride k: every k-th launch of each shape; default 1 = all)
        lib.PROFILE_STRIDE, lib.PROFILE_SEEN = max(1, int(args.event_stride)), {}
    env_steps0, rounds0 = runner.learner.env_steps, runner.sampling_rounds
    grp = getattr(runner.learner, "group", None)
    if world > 1 and grp is not None:
        grp.enable_timing()  # HIP-event pairs around every collective of the timed region (algo/learning/dp.py)
    if args.workload == "c3":
        for sm in runner.samplers:
            sm.ingest_prof, sm.h2d_bytes = {}, 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.iteration()
    barrier()
    dt = time.perf_counter() - t0
    clock = clk.stop() if clk is not None else None
    runner.stop_sampler_thread()  # (threaded async mode, host envs) no-op otherwise
    direct_dma = [sm._direct_ok.get("obs") for sm in runner.samplers] if args.workload == "c3" else None
    runner.close_envs()           # env worker processes (c3) — a no-op for in-process envs
    ingest = None
    if args.workload == "c3":
        tot = dict(act_wait_s=0.0, env_step_s=0.0, stage_s=0.0)
        dma_ms, dma_bytes = 0.0, 0
        for sm in runner.samplers:
            for k in tot:
                tot[k] += sm.ingest_prof.get(k, 0.0)
            for e0, e1, nbytes in sm.ingest_prof.get("dma_events", []):
                dma_ms += e0.elapsed_time(e1)
                dma_bytes += nbytes
            sm.ingest_prof = None
        h2d = sum(sm.h2d_bytes for sm in runner.samplers)
        rounds = max(1, runner.sampling_rounds - rounds0)
        ingest = {"h2d_bytes_per_env_step": round(h2d / (rounds * B * T), 1), "h2d_total_gb": round(h2d / 1e9, 3),
                  "obs_dma_gbs": round(dma_bytes / (dma_ms * 1e-3) / 1e9, 2) if dma_ms else None,
                  "obs_dma_ms_total": round(dma_ms, 1), "h2d_gbs_over_wall_clock": round(h2d / dt / 1e9, 2),
                  "host_s": {k: round(v, 3) for k, v in tot.items()}, "wall_s": round(dt, 3), "sampling_rounds": rounds,
                  "sampler_thread": bool(runner.threaded),
                  "env_worker_processes": int(args.env_workers) if cfg.env_workers_mode == "process" else 0,
                  "obs_dma_from_worker_pages_in_place": direct_dma,
                  "dma_share_of_wall_clock": round(dma_ms * 1e-3 / dt, 4)}
    prof, lib.PROFILE, lib.PROFILE_ONLY = lib.PROFILE, None, None
    seen, lib.PROFILE_STRIDE, lib.PROFILE_SEEN = dict(lib.PROFILE_SEEN), 1, {}
    collectives = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        summ = grp.timing_summary() if grp is not None else None
        if summ is not None:
            # per rank: time the compute stream spent at collectives (exposed), duration of the gradient exchange on its
            # own stream (native path only) and the number of collectives, per step
            mine = torch.tensor([summ[0], -1.0 if summ[1] is None else summ[1], float(summ[2])], dtype=torch.float64,
                                device="cuda" if backend == "nccl" else "cpu")  # (gloo gathers host tensors only)
            allr = [torch.zeros_like(mine) for _ in range(world)]
            torch.distributed.all_gather(allr, mine)
            collectives = {
                "gradient_exchange": "sf_allreduce_grads (C-ABI, own RCCL communicator + exchange stream)"
                if grp.native else "torch.distributed all_reduce (RCCL), async tail bucket",
                "small_buckets": (f"one-shot mailbox exchange (sf_dp_oneshot_*, buckets <= {grp._os_cap} bytes)"
                                  if getattr(grp, "_os", None) is not None else "torch.distributed all_reduce (RCCL)"),
                "exposed_ms_per_step": [round(float(a[0]) / args.steps, 3) for a in allr],
                "allreduce_ms_per_step": [None if float(a[1]) < 0 else round(float(a[1]) / args.steps, 3) for a in allr],
                "collectives_per_step": [round(float(a[2]) / args.steps, 1) for a in allr],
                "note": "exposed = HIP-event pairs on the compute stream around every collective / bucket wait; "
                        "allreduce = pairs on the exchange stream (native path; torch's collectives run on a stream of "
                        "its own that cannot be bracketed from outside)"}
    # the metric as SURVEY.md 8(d) / the reference define it (algo/runners/runner.py:761-764): env steps the LEARNER
    # consumed over the wall-clock of the region — never a product of the arguments.  The synchronous workloads train on
    # every rollout they collect, so the count must also equal steps x envs x rollout x ranks: anything skipped inside the
    # timed region (a dataset not trained on, a rollout not collected) fails here instead of inflating the number.
    env_steps = int(runner.learner.env_steps - env_steps0)
    if args.workload != "c3" and not args.async_rl:
        expect = args.steps * B * T * world
        if env_steps != expect:
            raise SystemExit(f"bench: the learner consumed {env_steps} env steps in the timed region, expected "
                             f"{expect} = steps x envs x rollout x ranks: work was skipped or repeated")
    value = env_steps / dt

    if not prof:
        if rank == 0:
            print(json.dumps({"metric": metric, "value": round(value, 1), "unit": "env-steps/s", "steps": args.steps,
                              "warmup": args.warmup, "higher_is_better": True, "scaling": "weak", "data": "synthetic",
                              "ms_per_step": round(dt / args.steps * 1e3, 2), "n_gpus": world, "rccl_ranks": rccl_ranks,
                              "rank_devices": rank_devices, "config": {"workload": workload_desc},
                              **({"collectives": collectives} if collectives else {}),
                              **({"ingest": ingest} if ingest is not None else {})}))
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    # ---- roofline of the dominant kernel instantiation (largest total time; HIP events over the timed region).
    # achieved = mean algorithmic FLOPs per launch / mean launch duration = sum(flops) / sum(duration) over its launches;
    # avg_launch_ms is directly comparable with AverageNs of the same name in profiles/r01_*_kernel_stats.csv
    # per shape: mean duration of the TIMED launches (every event_stride-th) x ALL launches of that shape in the region
    total_ms, launches, flops, shapes, timed = 0.0, 0, 0.0, [], 0
    for key, evs in prof.items():
        ms = [s_.elapsed_time(e_) for s_, e_ in evs]
        n_all = int(seen.get(key, len(ms))) if len(ms) else 0
        if not ms:
            continue
        avg = sum(ms) / len(ms)
        total_ms += avg * n_all
        launches += n_all
        timed += len(ms)
        flops += kernel_flops(key) * n_all
        shapes.append({"shape": f"{key[0]}:{key[2]}x{key[3]}->{key[5]} n={key[1]}", "launches": n_all, "timed_launches": len(ms),
                       "avg_ms": round(avg, 4),
                       "tflops": round(kernel_flops(key) / (avg * 1e-3) / 1e12, 1)})
    shapes.sort(key=lambda d: -d["avg_ms"] * d["launches"])
    avg_ms = total_ms / launches
    achieved = flops / (total_ms * 1e-3) / 1e12
    traffic, traffic_src = None, None
    # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench (tools/profile_round.sh + tools/pmc_traffic.py).  A
    # counter file is only valid for the kernels it was measured on: it carries the hash of the kernel sources, and the
    # field stays null when no committed file matches the code that just ran (instead of quoting a stale number).
    import glob
    from sample_factory_amd.build import source_sha16
    sha = source_sha16()
    for tj in sorted(glob.glob(os.path.join(ROOT, "profiles", "*traffic*.json")), reverse=True):
        doc = json.load(open(tj))
        ent = doc.get(dominant)
        if ent and doc.get("_kernel_source_sha16") == sha:
            traffic = ent["hbm_bytes"]
            traffic_src = f"profiles/{os.path.basename(tj)} (2*FETCH_SIZE + WRITE_SIZE, KiB -> bytes; kernel sources {sha})"
            break
    if traffic is None:
        traffic_src = f"no PMC pass committed for kernel sources {sha} (profiles/*traffic*.json are stamped with the hash)"
    kern = []  # ranking of all network kernels from the instrumented warm-up step (NOT the timed region)
    for k2, evs2 in warm_prof.items():
        m2 = sorted(s_.elapsed_time(e_) for s_, e_ in evs2)
        med = m2[len(m2) // 2]
        kern.append((med * len(m2), k2, len(m2), med))
    kern.sort(reverse=True, key=lambda k: k[0])
    roofline = {"bound": "mfma", "kernel": dominant, "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                "traffic_source": traffic_src, "avg_launch_ms": round(avg_ms, 4), "launches": launches,
                "timed_launches": timed, "event_stride": int(args.event_stride),
                "gflop_per_launch": round(flops / launches / 1e9, 3),
                "share_of_step_time": round(total_ms / (dt * 1e3), 4), "shapes": shapes}
    if args.workload == "c3":
        roofline["scope"] = (f"training-side launches only (n = {cfg.batch_size}); they run while the rollout side's small "
                             f"inference launches share the chip (async_rl)")
    if clock is not None:  # peak is priced at the 2.4 GHz maximum clock; the timed region ran at clock["ghz"] (median sample)
        roofline["clock_ghz"] = clock["ghz"]
        roofline["clock"] = clock
        roofline["frac_at_measured_clock"] = round(achieved / (PEAK_F32_MFMA_TFLOPS * clock["ghz"] / 2.4), 4)
    else:
        roofline["clock_ghz"] = None
        roofline["clock_note"] = "no clock sample (sf_clock_probe): frac is priced at 2.4 GHz"
    if exact_bf16_kernel(dominant):  # HBM-bound kernel: algorithmic bytes per launch / launch duration against 8 TB/s
        nbytes = sum(kernel_bytes(key) * int(seen.get(key, len(evs))) for key, evs in prof.items() if evs)
        gbs = nbytes / (total_ms * 1e-3) / 1e9
        roofline.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(gbs / PEAK_HBM_GBS, 4), "gbytes_per_launch": round(nbytes / launches / 1e9, 4)})
        roofline.pop("gflop_per_launch")
        roofline.pop("frac_at_measured_clock", None)  # (an HBM-bound kernel is not priced against the shader clock)
    net_ms = sum(k[0] for k in kern)
    net_flops = sum(kernel_flops(k[1]) * k[2] for k in kern)
    breakdown = [{"kernel": f"{k[1][0]}:{k[1][2]}x{k[1][3]}->{k[1][5]} n={k[1][1]}", "name": k[1][-1],
                  "ms_total": round(k[0], 2), "launches": k[2],
                  "tflops": round(kernel_flops(k[1]) / (k[3] * 1e-3) / 1e12, 1),
                  **({"arith": "exact products on the bf16 pipe (tflops = f32-equivalent)",
                      "gbs": round(kernel_bytes(k[1]) / (k[3] * 1e-3) / 1e9, 0)} if exact_bf16_kernel(k[1][-1]) else {})}
                 for k in kern[:14]]

    out = {
        "metric": metric, "value": round(value, 1), "unit": "env-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "dtype_note": ("f32 storage, f32 accumulation everywhere; products: f32 MFMA (v_mfma_f32_32x32x2_f32 / 16x16x4_f32), "
                       "except conv1 on raw u8 frames = v_mfma_f32_16x16x32_bf16 on operands that are EXACT in bf16 (u8 pixels; "
                       "f32 weights / output gradients split exactly into 3 bf16 terms) - every product exact, no rounding of any "
                       "operand (csrc/sf_nn_u8.h)") if args.workload != "c5" else "f32 (f32 MFMA, f32 accumulation)",
        "rccl_ranks": rccl_ranks, "rank_devices": rank_devices,
        **({"rank_affinity": affinity} if affinity is not None else {}),
        "config": {"workload": workload_desc,
                   "envs_per_gpu": B, "rollout": T, "batch_size": cfg.batch_size, "num_batches_per_epoch": args.num_batches,
                   "num_epochs": args.num_epochs, "parallelism": f"dp{world}"},
        "roofline": roofline,
        # host side of the rollout step: the step's library calls recorded once and replayed with one foreign call per
        # launch (lib.LaunchProgram) -- same launches, same arguments, same stream; SF_LAUNCH_PROGRAMS=0 = off
        "rollout_launch_programs": {"enabled": bool(lib.LAUNCH_PROGRAMS),
                                    "replayed_steps": int(sum(getattr(sm, "program_replays", 0) for sm in runner.samplers))},
        **({"collectives": collectives} if collectives else {}),
        **({"ingest": ingest} if ingest is not None else {}),
        "network_kernels": {"source": "instrumented warm-up step (every launch timed in isolation; not the timed region)",
                            "ms_per_step": round(net_ms, 2), "tflops_avg": round(net_flops / (net_ms * 1e-3) / 1e12, 2),
                            "top": breakdown},
    }
    if rank == 0 and world == 1 and args.workload == "c2" and not args.no_secondary and not args.async_rl \
            and args.envs == 4096:
        # release this run's slab and activations first: the secondary runs get the whole device
        del runner
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        out["secondary"] = secondary_lines(args)
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == "c2":
        out["cpu_baseline"] = cpu_baseline_leg(args, T)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        if grp is not None:
            grp.close()  # native communicator / one-shot mailboxes (collective: every rank gets here)
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

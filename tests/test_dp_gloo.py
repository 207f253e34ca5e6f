"""Data-parallel learner protocol with world_size 2 over gloo on CPU tensors.

The collectives are the product's (`sample_factory_amd.algo.learning.dp.ReplicaGroup`); the per-rank compute is the
CPU oracle, so the test checks exactly the parity statement of the design: a 2-replica step on env-sharded data ==
the single-replica step on the concatenated minibatch (up to fp32 summation order)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_batch(N=512, A=6, F=16):
    rng = np.random.default_rng(7)
    return dict(
        feats=rng.standard_normal((N, F)).astype(np.float32),           # last hidden layer (input of the heads GEMM)
        params=rng.standard_normal((N, A)).astype(np.float32), values=rng.standard_normal(N).astype(np.float32),
        actions=rng.integers(0, A, N).astype(np.float32), old_logp=(-rng.random(N) * 2 - 0.2).astype(np.float32),
        old_params=rng.standard_normal((N, A)).astype(np.float32), old_values=rng.standard_normal(N).astype(np.float32),
        adv=(rng.standard_normal(N) * 2 + 0.3).astype(np.float32), targets=rng.standard_normal(N).astype(np.float32),
        valids=rng.random(N) > 0.1, returns=(rng.standard_normal(N) * 3 + 1).astype(np.float32))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from sample_factory_amd.algo.learning.dp import ReplicaGroup
    grp = ReplicaGroup()
    assert grp.world == world and grp.rank == rank and grp.env_shard(256) == (rank * 256, (rank + 1) * 256)
    b = _make_batch()
    N = b["adv"].shape[0]
    sl = slice(rank * N // world, (rank + 1) * N // world)      # shard by environment = contiguous rows
    s = {k: v[sl] for k, v in b.items()}
    # (1) advantage moments: {sum, sumsq, n_valid} all-reduced before the loss
    va = s["valids"]
    mom = torch.tensor([s["adv"][va].astype(np.float64).sum(), (s["adv"][va].astype(np.float64) ** 2).sum(), va.sum()],
                       dtype=torch.float64)
    grp.all_reduce_sum(mom)
    out = oracle.ppo_loss(s["params"], s["values"], s["actions"], s["old_logp"], s["old_params"], s["old_values"],
                          s["adv"], s["targets"], s["valids"], exploration_coeff=0.01, kl_coeff=0.1,
                          ext_moments=mom.numpy())
    # (2) the flat gradient bucket: heads wgrad of this shard, SUM-all-reduced (each shard already carries 1/n_global)
    g_heads = np.concatenate([out["grad_values"][:, None], out["grad_params"]], 1)
    bucket = torch.from_numpy(s["feats"].T.astype(np.float64) @ g_heads.astype(np.float64))
    bucket2 = bucket.clone().reshape(-1)
    grp.all_reduce_sum(bucket)
    # (2b) the learner's two-bucket form: the tail goes out asynchronously (while the backward pass would still be
    # producing the head), then the head, then wait — every element summed once, same result
    cut = bucket2.numel() // 3
    bucket3 = bucket2.clone()
    work = grp.all_reduce_sum_async(bucket2[cut:])
    grp.all_reduce_sum(bucket2[:cut])
    work.wait()
    assert torch.equal(bucket2, bucket.reshape(-1))
    # the entry points the learner actually calls (they route to sf_allreduce_grads under cfg.dp_native_rccl, to the
    # torch.distributed collectives otherwise — this group is not native: no GPU here)
    assert not grp.native
    work = grp.all_reduce_grads_async(bucket3[cut:])
    grp.all_reduce_grads(bucket3[:cut])
    work.wait()
    assert torch.equal(bucket3, bucket.reshape(-1))
    # (3) additive loss sums + max KL
    n_loc = float(va.sum())
    sums = torch.tensor([-out["policy_loss"] * mom[2].item(), 0.0, 0.0, 0.0, out["kl_max"], n_loc, 0.0, 0.0],
                        dtype=torch.float64)
    grp.loss_sums(sums)
    # (3b) the learner's per-epoch pack [minibatches, 8]: additive loss columns, column 6 = max KL and column 7 = the
    # abort word of the fused recurrent passes, both MAX — only rank 1 "aborted", every rank must see it (learner.py: one
    # rank raising alone would leave the others waiting in the next collective)
    pack = torch.arange(16, dtype=torch.float32).reshape(2, 8) * (rank + 1)
    pack[:, 6] = torch.tensor([0.25, 0.5]) * (rank + 1)
    pack[:, 7] = float(rank == 1)
    grp.reduce_sum_max(pack, (6, 7))
    want = torch.arange(16, dtype=torch.float32).reshape(2, 8) * 3.0   # ranks 0 and 1: factors 1 + 2
    want[:, 6] = torch.tensor([0.5, 1.0])
    want[:, 7] = 1.0
    assert torch.equal(pack, want), pack
    # (4) returns normaliser: global batch moments -> identical statistics on every rank
    rmom = torch.tensor([s["returns"].astype(np.float64).sum(), (s["returns"].astype(np.float64) ** 2).sum(),
                         float(len(s["returns"]))], dtype=torch.float64)
    grp.all_reduce_sum(rmom)
    # (5) broadcast of initial weights
    w = torch.full((5,), float(rank + 1))
    grp.broadcast(w, src=0)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), bucket=bucket.numpy(), mom=mom.numpy(), sums=sums.numpy(),
             rmom=rmom.numpy(), w=w.numpy(), g_heads=g_heads)
    dist.destroy_process_group()


def test_two_replicas_equal_one_on_the_concatenated_minibatch(tmp_path):
    import oracle
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(world)]
    b = _make_batch()
    ref = oracle.ppo_loss(b["params"], b["values"], b["actions"], b["old_logp"], b["old_params"], b["old_values"],
                          b["adv"], b["targets"], b["valids"], exploration_coeff=0.01, kl_coeff=0.1)
    g_ref = np.concatenate([ref["grad_values"][:, None], ref["grad_params"]], 1)
    # per-sample gradients of the shards == rows of the single-replica gradient (global normalisation + global 1/n)
    np.testing.assert_allclose(np.concatenate([r[0]["g_heads"], r[1]["g_heads"]]), g_ref, rtol=2e-5, atol=1e-9)
    bucket_ref = b["feats"].T.astype(np.float64) @ g_ref.astype(np.float64)
    for i in range(world):
        np.testing.assert_allclose(r[i]["bucket"], bucket_ref, rtol=1e-5, atol=1e-9)      # identical on every rank
        assert r[i]["mom"][2] == b["valids"].sum()
        np.testing.assert_allclose(r[i]["sums"][0] / r[i]["mom"][2], -ref["policy_loss"], rtol=1e-5)
        assert abs(r[i]["sums"][4] - ref["kl_max"]) < 1e-6 and r[i]["sums"][5] == b["valids"].sum()
        np.testing.assert_array_equal(r[i]["w"], np.ones(5, np.float32))
    np.testing.assert_array_equal(r[0]["bucket"], r[1]["bucket"])                          # replicas stay in lock-step
    # returns normaliser: Chan merge from the all-reduced moments == oracle update on the concatenated returns
    n = r[0]["rmom"][2]
    bm = r[0]["rmom"][0] / n
    bv = (r[0]["rmom"][1] - r[0]["rmom"][0] * bm) / (n - 1)
    st = oracle.rms_update(np.array([0.0, 1.0, 1.0]), b["returns"])
    delta, tot = np.float32(bm) - 0.0, 1.0 + n
    np.testing.assert_allclose([delta * n / tot, (1.0 + np.float32(bv) * n + delta * delta * n / tot) / tot, tot], st, rtol=1e-6)


def test_bench_self_launches_replicas_and_checks_the_collective():
    """`python bench.py --gpus 2` outside torchrun re-launches itself under torch.distributed.run (one rank per GPU),
    every rank joins the group, the rank-stamped all-reduce proves both contributed; --check_launch stops before any GPU
    work so the launcher / argument / env plumbing is testable on a CPU box (gloo)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["SF_DP_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--check_launch", "--workload", "c5"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["launch_check"] and out["rccl_ranks"] == 2 and out["n_gpus"] == 2 and out["backend"] == "gloo"
    assert len(out["rank_devices"]) == 2 and out["workload"] == "c5"


def test_bench_rccl_launch_without_enough_devices_fails_in_rank_code():
    """with the production backend (nccl = RCCL) on a box with fewer GPUs than ranks the error comes from the rank code
    ("need N devices"), not from a launcher guard"""
    import subprocess
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("box has >= 2 GPUs")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "SF_DP_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "need 2 devices on this node" in (r.stderr + r.stdout)


def test_collective_timing_hooks_without_a_gpu():
    """ReplicaGroup.enable_timing() on a box without a GPU: the event brackets are inert (no CUDA), the counters run, and
    the packed sum/max exchange of a single-rank forced group is the identity"""
    from sample_factory_amd.algo.learning import dp
    grp = dp.ReplicaGroup()                      # no process group: inactive
    grp.enable_timing()
    t = torch.arange(6, dtype=torch.float64)
    assert grp.reduce_sum_max(t.clone(), (2,)).equal(t) and grp.all_reduce_sum(t.clone()).equal(t)
    with dp._Exposed(grp) as e:                  # bracket object keeps its group (regression: AttributeError on exit)
        assert e.group is grp
    if not torch.cuda.is_available():
        assert grp.timing["exposed"] == [] and grp.timing["count"] == 0   # inactive group: nothing was issued

"""Data-parallel replicas on the real kernels: 2 ranks (sharing the one GPU of the test box, collectives over gloo with
host staging) must reproduce the single-replica run on the concatenated env set — rollouts bit-identical, weights equal
up to fp32 summation order.  (Production uses nccl = RCCL; the replica protocol is backend-independent.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(num_agents, **kw):
    from sample_factory_amd.cfg.arguments import default_cfg
    return default_cfg(env="synthetic_atari", use_rnn=False, nonlinearity="relu", normalize_input=False, obs_scale=255.0,
                       encoder_conv_architecture="convnet_atari", rollout=8, batch_size=num_agents * 8,
                       num_batches_per_epoch=1, num_epochs=1, num_workers=1, num_envs_per_worker=1, async_rl=False,
                       seed=5, serial_mode=True, synthetic_num_agents=num_agents, exploration_loss_coeff=0.01,
                       kl_loss_coeff=0.05, learning_rate=1e-3, **kw)


def _run(num_agents, iters):
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_synthetic_env
    from sample_factory_amd.train import make_runner
    register_env("synthetic_atari", make_synthetic_env)
    cfg, runner = make_runner(_cfg(num_agents))
    runner.init()
    for _ in range(iters):
        stats = runner.iteration()
    torch.cuda.synchronize()
    ac = runner.learner.actor_critic
    return dict(params=ac.flat_params.cpu().numpy(), actions=runner.traj["actions"].cpu().numpy(),
                rewards=runner.traj["rewards"].cpu().numpy(), rms=ac.returns_normalizer.stats.cpu().numpy(),
                env_steps=stats["learner_env_steps"], loss=stats["train"]["loss"])


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", SF_DP_BACKEND="gloo")
    r = _run(32, 3)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **r)
    torch.distributed.destroy_process_group()


def test_two_replicas_equal_one(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for k in ("WORLD_SIZE", "RANK"):
        os.environ.pop(k, None)
    single = _run(64, 3)
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(2)]
    assert int(r[0]["env_steps"]) == single["env_steps"] == 3 * 64 * 8           # whole-job step accounting
    np.testing.assert_array_equal(r[0]["params"], r[1]["params"])                  # replicas stay in lock-step
    np.testing.assert_array_equal(r[0]["rms"], r[1]["rms"])
    # loss scalars are all-reduced once per epoch (each replica divides its local sums by the GLOBAL n)
    assert float(r[0]["loss"]) == float(r[1]["loss"])
    assert abs(float(r[0]["loss"]) - single["loss"]) < 2e-3 * max(1.0, abs(single["loss"]))
    # identical rollouts (integer actions exact) — requires identical weights after every SGD step, up to sampling
    # thresholds; compare the LAST rollout's actions of the shards with the single run
    acts = np.concatenate([r[0]["actions"], r[1]["actions"]])
    assert (acts == single["actions"]).mean() > 0.995
    # return statistics are functions of the value head, i.e. of weights that differ by fp32 summation order (split
    # counts of the reductions depend on the per-replica batch) amplified by three Adam steps: ~1e-4 relative
    np.testing.assert_allclose(r[0]["rms"], single["rms"], rtol=1e-3)
    diff = np.abs(r[0]["params"] - single["params"])                               # 3 Adam steps at lr 1e-3 move weights by
    # up to 3e-3 possible: Adam turns a sum-order sign flip of a near-zero gradient into a full +-lr step.  (The
    # training forward of conv1 runs the strip-image kernel, the rollout the im2col kernel: 1e-6 apart, both exact f32.)
    assert diff.max() < 1e-3 and (diff > 2e-5).mean() < 2e-3, (diff.max(), (diff > 2e-5).mean())


def _free_port():
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


# ------------------------------------------------------------------------------------------ BASELINE configs[4] shape
def _cfg_c5(num_agents, rnn_type, nb=1, **kw):
    """BASELINE.json configs[4] ("Mujoco Ant-v4 continuous-action, LSTM core + V-trace, 2xMI355X") at test size: Box(8)
    actions with learned stddev, MLP[64,64] tanh encoder, 256-wide recurrent core on the fused sequence kernels, V-trace,
    KL loss, value bootstrap, normalize_input=True (the obs normaliser's moments are a per-dataset all-reduce under DP).
    nb = 1: the dataset is ONE minibatch, so the global minibatch of G replicas is the single replica's minibatch (with
    nb > 1 every replica cuts its own shard into nb pieces: an equally valid but different partition)"""
    from sample_factory_amd.cfg.arguments import default_cfg
    T = 8
    return default_cfg(env="synthetic_ant", use_rnn=True, rnn_type=rnn_type, rnn_size=256, recurrence=T, rollout=T,
                       encoder_mlp_layers=[64, 64], nonlinearity="tanh", normalize_input=True, normalize_returns=False,
                       with_vtrace=True, kl_loss_coeff=0.1, adaptive_stddev=False, policy_initialization="torch_default",
                       value_bootstrap=True, max_grad_norm=3.5, ppo_clip_ratio=0.2, value_loss_coeff=1.3,
                       exploration_loss_coeff=0.0, learning_rate=1e-4, gamma=0.99, gae_lambda=0.95,
                       batch_size=num_agents * T // nb, num_batches_per_epoch=nb, num_epochs=2, num_workers=1,
                       num_envs_per_worker=1, async_rl=False, seed=7, serial_mode=True, synthetic_num_agents=num_agents, **kw)


def _run_c5(num_agents, iters, rnn_type, **kw):
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_synthetic_continuous_env
    from sample_factory_amd.train import make_runner
    register_env("synthetic_ant", make_synthetic_continuous_env)
    cfg, runner = make_runner(_cfg_c5(num_agents, rnn_type, **kw))
    runner.init()
    ac = runner.learner.actor_critic
    out = {}
    for it in range(iters):
        stats = runner.iteration()
        torch.cuda.synchronize()
        if it == 0:  # after ONE dataset: same rollout up to fp32 round-off of the first forward, 4 Adam steps
            out.update(params1=ac.flat_params.cpu().numpy(), obs1=runner.traj["obs"]["obs"].cpu().numpy(),
                       acts1=runner.traj["actions"].cpu().numpy(), loss1=stats["train"]["loss"])
            sd1 = ac.state_dict()
    sd = ac.state_dict()
    pfx = "obs_normalizer.running_mean_std.running_mean_std.obs."
    out.update(params=ac.flat_params.cpu().numpy(), loss=stats["train"]["loss"], env_steps=stats["learner_env_steps"],
               fused=bool(ac._rnn_saved["fused"]), train_step=runner.learner.train_step,
               **{f"obsn_{k}": sd[pfx + k].double().numpy() for k in ("running_mean", "running_var", "count")},
               **{f"obsn1_{k}": sd1[pfx + k].double().numpy() for k in ("running_mean", "running_var", "count")})
    return out


def _worker_c5(rank, world, port, out_dir, rnn_type):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", SF_DP_BACKEND="gloo")
    r = _run_c5(32, 3, rnn_type)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **r)
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("rnn_type", ["lstm", "gru"])
def test_two_replicas_equal_one_recurrent_vtrace_normalized_input(tmp_path, rnn_type):
    """2 replicas x 32 envs == 1 replica x 64 envs on the configs[4] shape: the replicas' weights and observation-normaliser
    statistics are IDENTICAL bit for bit (every collective hands every rank the same bytes), and equal to the single
    replica up to fp32 summation order."""
    mp.spawn(_worker_c5, args=(2, _free_port(), str(tmp_path), rnn_type), nprocs=2, join=True)
    for k in ("WORLD_SIZE", "RANK"):
        os.environ.pop(k, None)
    single = _run_c5(64, 3, rnn_type)
    r = [np.load(tmp_path / f"rank{i}.npz") for i in range(2)]
    assert single["fused"] and bool(r[0]["fused"]) and bool(r[1]["fused"]), "BPTT must run as the persistent sequence kernels"
    assert int(r[0]["env_steps"]) == single["env_steps"] == 3 * 64 * 8 and int(r[0]["train_step"]) == single["train_step"] == 6
    # ---- replicas in lock-step, bit for bit
    for k in ("params", "params1", "obsn_running_mean", "obsn_running_var", "obsn_count", "obsn1_running_mean", "obsn1_running_var"):
        np.testing.assert_array_equal(r[0][k], r[1][k], err_msg=k)
    assert float(r[0]["loss"]) == float(r[1]["loss"]) and float(r[0]["loss1"]) == float(r[1]["loss1"])
    # ---- first dataset: the shards' rollout is the single replica's rollout (same env streams, same sampler keys; the
    # forward of 32 vs 64 rows may pick another tile shape: fp32 round-off in the continuous actions)
    obs = np.concatenate([r[0]["obs1"], r[1]["obs1"]])
    acts = np.concatenate([r[0]["acts1"], r[1]["acts1"]])
    np.testing.assert_allclose(acts, single["acts1"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(obs, single["obs1"], rtol=0, atol=2e-5)
    # GLOBAL observation-normaliser moments (running_mean_std.py:51-62 over both shards' rows): f64 statistics
    assert float(r[0]["obsn1_count"].item()) == float(single["obsn1_count"].item()) == 1.0 + 64 * 9
    np.testing.assert_allclose(r[0]["obsn1_running_mean"], single["obsn1_running_mean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(r[0]["obsn1_running_var"], single["obsn1_running_var"], rtol=1e-5, atol=1e-6)
    assert abs(float(r[0]["loss1"]) - single["loss1"]) < 1e-3 * max(1.0, abs(single["loss1"]))
    # 2 Adam steps at lr 1e-4: a summation-order sign flip of a ~0 gradient is a full +-lr step per SGD step
    d1 = np.abs(r[0]["params1"] - single["params1"])
    assert d1.max() <= 2.1e-4 and (d1 > 2e-5).mean() < 2e-3, (d1.max(), (d1 > 2e-5).mean())
    # ---- three datasets (6 Adam steps through a recurrent core): still the same run
    d = np.abs(r[0]["params"] - single["params"])
    assert d.max() <= 6.5e-4 and (d > 5e-5).mean() < 5e-3, (d.max(), (d > 5e-5).mean())
    np.testing.assert_allclose(r[0]["obsn_running_mean"], single["obsn_running_mean"], rtol=1e-3, atol=1e-4)
    assert abs(float(r[0]["loss"]) - single["loss"]) < 5e-3 * max(1.0, abs(single["loss"]))


@pytest.mark.parametrize("kind", ["c2", "c5"])
def test_rccl_branch_executes_with_one_rank(tmp_path, kind):
    """The production backend (nccl = RCCL) on the ONE GPU of the test box: a single-rank process group with
    cfg.dp_force_collectives issues every collective of the data-parallel learner for real — broadcast of the initial
    weights, the per-minibatch 3-double moment all-reduce, the two-bucket gradient all-reduce (async_op on a slice of
    the flat gradient while the conv layers are still back-propagated), SUM / MAX of the loss scalars, int64 invalid
    counts — so dtype / reduce-op / view / stream-ordering mistakes in that branch fail here, not on the 8-GPU node.
    With one rank every reduction is the identity: the run must equal the run without a process group bit for bit."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import json, os, sys, torch
sys.path.insert(0, %r)
from sample_factory_amd.cfg.arguments import default_cfg
from sample_factory_amd.envs.env_utils import register_env
from sample_factory_amd.envs.synthetic import make_synthetic_env
from sample_factory_amd.train import make_runner
force = os.environ.get("FORCE") in ("1", "2")
native = os.environ.get("FORCE") == "2"
if force:
    torch.cuda.set_device(0)
    torch.distributed.init_process_group("nccl")
register_env("synthetic_atari", make_synthetic_env)
dp_kw = dict(train_dir=%r, experiment="rccl" + str(int(force)), data_parallel=force, dp_force_collectives=force,
             dp_native_rccl=native, lr_schedule="kl_adaptive_epoch")
if os.environ["KIND"] == "c5":   # BASELINE configs[4] shape: recurrent core + V-trace + global obs-normaliser moments
    sys.path.insert(0, os.path.join(%r, "tests"))
    from test_gpu_dp import _cfg_c5
    from sample_factory_amd.envs.synthetic import make_synthetic_continuous_env
    register_env("synthetic_ant", make_synthetic_continuous_env)
    cfg = _cfg_c5(64, "lstm", nb=2, **dp_kw)
else:
    cfg = default_cfg(env="synthetic_atari", use_rnn=False, nonlinearity="relu", normalize_input=False, obs_scale=255.0,
                      encoder_conv_architecture="convnet_atari", rollout=8, batch_size=1024, num_batches_per_epoch=2,
                      num_epochs=2, num_workers=1, num_envs_per_worker=1, worker_num_splits=1, async_rl=False, seed=1,
                      serial_mode=True, synthetic_num_agents=256, **dp_kw)
cfg, runner = make_runner(cfg)
runner.init()
assert runner.learner.dp == force and (not force or os.environ["KIND"] == "c5" or runner.learner._dp_split is not None)
assert (runner.learner.group is not None and runner.learner.group.native) == native
for _ in range(3):
    stats = runner.iteration()
torch.cuda.synchronize()
p = runner.learner.actor_critic.flat_params
print("RESULT " + json.dumps(dict(sum=float(p.double().sum()), abs=float(p.double().abs().sum()), loss=stats["train"]["loss"],
                                  steps=runner.learner.train_step, backend=torch.distributed.get_backend() if force else None)))
if native:
    runner.learner.group.close()
if force:
    torch.distributed.destroy_process_group()
''' % (root, str(tmp_path), root)
    out = {}
    for force in ("0", "1", "2"):  # 2: the gradient buckets through the C-ABI (sf_allreduce_grads), the rest as in 1
        env = {k: v for k, v in os.environ.items() if k not in ("SF_DP_BACKEND",)}
        env.update(FORCE=force, KIND=kind, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2500:]
        out[force] = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert out["1"]["backend"] == "nccl" and out["1"]["steps"] == out["0"]["steps"] == 12
    for k in ("1", "2"):
        assert out[k]["sum"] == out["0"]["sum"] and out[k]["abs"] == out["0"]["abs"] and out[k]["loss"] == out["0"]["loss"], k
    assert out["2"]["steps"] == 12


def test_c_abi_rccl_exchange_single_rank():
    """csrc/sf_dp.hip through ctypes on the one GPU of the box: id -> communicator of ONE rank -> in-place fp32 SUM on a
    side stream (identity), f64 sum / max, byte broadcast, info, destroy.  (Two ranks need two devices: RCCL refuses
    duplicate GPUs, so the multi-rank run belongs to the driver's 8-GPU node.)"""
    from sample_factory_amd import lib
    lib.load()
    torch.cuda.set_device(0)
    ident = lib.dp_unique_id()
    assert len(ident) == 128 and ident != bytes(128)
    comm = lib.dp_comm_create(ident, 1, 0)
    assert lib.dp_comm_info(comm) == (1, 0)
    g = torch.randn(1_687_744, device="cuda")
    want = g.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    lib.allreduce_grads(comm, g, side)
    lib.allreduce_grads(comm, g[1_000_000:], side)          # a tail slice of the flat buffer (the overlap bucket)
    torch.cuda.current_stream().wait_stream(side)
    assert torch.equal(g, want)
    d = torch.tensor([1.5, -2.0, 3.0], dtype=torch.float64, device="cuda")
    lib.dp_allreduce_f64(comm, d, "sum")
    lib.dp_allreduce_f64(comm, d, "max")
    assert d.tolist() == [1.5, -2.0, 3.0]
    b = torch.arange(1000, dtype=torch.int32, device="cuda")
    lib.dp_broadcast(comm, b, 0)
    assert torch.equal(b, torch.arange(1000, dtype=torch.int32, device="cuda"))
    with pytest.raises(lib.SfHipError):
        lib.allreduce_grads(comm, torch.zeros(4))            # host tensor
    torch.cuda.synchronize()
    lib.dp_comm_destroy(comm)


def test_bench_contract_with_two_ranks_sharing_the_gpu():
    """`python bench.py --gpus 2` as the driver invokes it (self-launch under torch.distributed.run, one JSON line from
    rank 0, whole-job aggregate, barrier + max-over-ranks timing) — over gloo with host staging, because the test box has
    ONE GPU and RCCL refuses two ranks on it; everything but the collective backend is the production path."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env["SF_DP_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--envs", "256", "--no_cpu_baseline"], capture_output=True, text=True, env=env, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1000:]          # rank 0 only
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["scaling"] == "weak" and j["steps"] == 2
    assert j["config"]["parallelism"] == "dp2" and j["value"] > 0
    assert abs(j["value"] - 2 * 256 * 32 * 2 / (j["ms_per_step"] * 2e-3)) < 0.02 * j["value"]   # aggregate over both ranks
    # every collective of the timed region was bracketed with HIP events, per rank (algo/learning/dp.py)
    c = j["collectives"]
    assert len(c["exposed_ms_per_step"]) == 2 and all(x > 0 for x in c["exposed_ms_per_step"])
    assert all(x >= 10 for x in c["collectives_per_step"])      # 4 minibatches x (moments + gradient buckets) + per-dataset
    assert c["allreduce_ms_per_step"] == [None, None]           # only the native path has an exchange stream to bracket


# ------------------------------------------------------------------------------------------ overlap ORDER of the buckets
_ORDER_CODE = r'''
import json, os, sys, torch
sys.path.insert(0, %r)
from sample_factory_amd import lib
from sample_factory_amd.cfg.arguments import default_cfg
from sample_factory_amd.envs.env_utils import register_env
from sample_factory_amd.envs.synthetic import make_synthetic_env
from sample_factory_amd.train import make_runner
native = os.environ["FORCE"] == "2"
torch.cuda.set_device(0)
torch.distributed.init_process_group("nccl")
register_env("synthetic_atari", make_synthetic_env)
cfg = default_cfg(env="synthetic_atari", use_rnn=False, nonlinearity="relu", normalize_input=False, obs_scale=255.0,
                  encoder_conv_architecture="convnet_atari", rollout=8, batch_size=2048, num_batches_per_epoch=1,
                  num_epochs=1, num_workers=1, num_envs_per_worker=1, worker_num_splits=1, async_rl=False, seed=1,
                  serial_mode=True, synthetic_num_agents=256, train_dir=%r, experiment="order" + os.environ["FORCE"],
                  data_parallel=True, dp_force_collectives=True, dp_native_rccl=native)
cfg, runner = make_runner(cfg)
runner.init()
lr, grp = runner.learner, runner.learner.group
assert lr.dp and lr._dp_split is not None and grp.native == native
runner.iteration()                       # warm-up (allocations, lazy code objects)
torch.cuda.synchronize()
# ONE SGD step with every network launch bracketed by HIP events (lib.PROFILE) and the bucket calls marked (grp.trace):
# all on the learner's compute stream, so elapsed_time(a, b) >= 0 <=> b was enqueued behind a
lib.PROFILE, grp.trace = {}, []
grp.enable_timing()
runner.iteration()
torch.cuda.synchronize()
prof, lib.PROFILE = lib.PROFILE, None
tr = grp.trace
tags = [t for t, _ in tr]
def ev(tag):
    hits = [e for t, e in tr if t == tag]
    assert len(hits) == 1, (tag, tags)
    return hits[0]
def launches(op, cin):
    out = []
    for key, evs in prof.items():
        if key[0] == op and key[2] == cin and key[1] == 2048:
            out += evs
    assert out, (op, cin, list(prof))
    return out
after = lambda a, b: a.elapsed_time(b)   # ms from a to b on the device time line
enq, wait, sync = ev("grad_async_enqueue"), ev("grad_async_wait"), ev("grad_sync")
fc_w = launches("wgrad", 3136)[-1]        # fc weight gradient: the last launch the tail bucket (fc + heads) depends on
c3_d = launches("dgrad", 64)[0]           # conv3 data gradient = first conv-backward launch
c1_w = launches("wgrad", 4)[-1]           # conv1 weight gradient = last launch of the backward pass
res = dict(
    tags=tags,
    fc_wgrad_end_to_enqueue=after(fc_w[1], enq), enqueue_to_conv3_dgrad_start=after(enq, c3_d[0]),
    conv1_wgrad_end_to_head_bucket=after(c1_w[1], sync), head_bucket_to_tail_wait=after(sync, wait),
    enqueue_to_wait=after(enq, wait), conv_backward_ms=after(c3_d[0], c1_w[1]),
    exposed_ms=grp.timing_summary()[0], exchange_ms=grp.timing_summary()[1])
print("RESULT " + json.dumps(res))
if native:
    grp.close()
torch.distributed.destroy_process_group()
'''


@pytest.mark.parametrize("force", ["1", "2"])
def test_tail_bucket_is_enqueued_before_the_conv_backward_and_waited_for_after_it(tmp_path, force):
    """DESIGN.md §6 prices the data-parallel step on the fc + heads bucket (95 % of the gradient bytes) being exchanged WHILE
    conv3 / conv2 / conv1 back-propagate.  That is a statement about enqueue order, checked here with HIP events on the
    learner's stream for both exchange paths (1: torch.distributed / RCCL, 2: sf_allreduce_grads on the exchange stream), one
    rank with forced collectives: the tail bucket's all-reduce is handed over right behind the fc layer's weight gradient
    and BEFORE the first conv-backward launch (conv3's data gradient); the compute stream first blocks on a bucket only
    behind conv1's weight gradient (the head bucket's exchange, then the tail bucket's wait)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("SF_DP_BACKEND",)}
    env.update(FORCE=force, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, "-c", _ORDER_CODE % (ROOT, str(tmp_path))], capture_output=True, text=True, env=env,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2500:]
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    # program order of the bucket calls of one SGD step
    assert res["tags"] == ["grad_async_enqueue", "grad_sync", "grad_async_wait"], res["tags"]
    assert res["fc_wgrad_end_to_enqueue"] >= 0.0           # the bucket is complete when it is handed over ...
    assert res["enqueue_to_conv3_dgrad_start"] >= 0.0      # ... and handed over before any conv-backward launch
    assert res["conv1_wgrad_end_to_head_bucket"] >= 0.0    # nothing blocks the compute stream before conv1's weight gradient
    assert res["head_bucket_to_tail_wait"] >= 0.0
    # the whole conv backward lies between the hand-over and the wait
    assert res["enqueue_to_wait"] >= res["conv_backward_ms"] > 0.0, res
    print("bucket order:", json.dumps(res))


# ------------------------------------------------------------------------------------------ per-epoch moment exchange
def _worker_mom(rank, world, port, out_dir, epoch_moments):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", SF_DP_BACKEND="gloo")
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_synthetic_env
    from sample_factory_amd.train import make_runner
    register_env("synthetic_atari", make_synthetic_env)
    cfg = _cfg(32, dp_epoch_moments=bool(epoch_moments))
    cfg.batch_size, cfg.num_batches_per_epoch, cfg.num_epochs, cfg.shuffle_minibatches = 64, 4, 2, True
    cfg, runner = make_runner(cfg)
    runner.init()
    grp = runner.learner.group
    runner.iteration()
    grp.enable_timing()
    runner.iteration()
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"mom{int(epoch_moments)}_rank{rank}.npz"),
             params=runner.learner.actor_critic.flat_params.cpu().numpy(), collectives=grp.timing["count"])
    torch.distributed.destroy_process_group()


def test_epoch_moment_exchange_equals_the_per_step_one(tmp_path):
    """cfg.dp_epoch_moments (default): the advantage moments of every minibatch of an epoch travel in ONE all-reduce before
    the epoch's first forward pass instead of one 24-byte all-reduce in front of every loss.  Two replicas sharing the GPU,
    4 shuffled minibatches x 2 epochs: the weights are bit-identical to the per-step exchange's, and the iteration issues
    3 collectives fewer per epoch."""
    out = {}
    for em in (1, 0):
        mp.spawn(_worker_mom, args=(2, _free_port(), str(tmp_path), em), nprocs=2, join=True)
        out[em] = [np.load(tmp_path / f"mom{em}_rank{r}.npz") for r in range(2)]
    for k in ("WORLD_SIZE", "RANK"):
        os.environ.pop(k, None)
    np.testing.assert_array_equal(out[1][0]["params"], out[1][1]["params"])   # replicas in lock-step
    np.testing.assert_array_equal(out[1][0]["params"], out[0][0]["params"])   # same numbers either way
    assert int(out[0][0]["collectives"]) - int(out[1][0]["collectives"]) == 2 * (4 - 1), \
        (int(out[0][0]["collectives"]), int(out[1][0]["collectives"]))


# ------------------------------------------------------------------------------------------ one-shot small-bucket exchange
def _worker_oneshot(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from sample_factory_amd.algo.learning.dp import ReplicaGroup
    grp = ReplicaGroup(oneshot_bytes=1 << 20)
    assert grp._os is not None and grp.world == world
    bad = []
    for it in range(30):  # both slots, many reuses; sizes: scalars, one block, the 0.31 MB conv bucket (20 blocks)
        for n in (3, 1000, 77_777):
            base = (torch.arange(n, device="cuda") % 97 + 1).float()
            t = base * float((rank + 1) * (it + 1))
            grp.all_reduce_sum(t)
            want = base * float(sum(r + 1 for r in range(world)) * (it + 1))  # small integers: exact in f32
            if not torch.equal(t, want):
                bad.append(("f32", it, n, float((t - want).abs().max())))
        d = torch.tensor([1.5 * (rank + 1), -2.0 * it, 7.0], dtype=torch.float64, device="cuda")
        grp.all_reduce_sum(d)
        wd = torch.tensor([1.5 * sum(r + 1 for r in range(world)), -2.0 * it * world, 7.0 * world], dtype=torch.float64)
        if not torch.equal(d.cpu(), wd):
            bad.append(("f64", it, d.tolist()))
        g = torch.full((50_000,), float(rank + 1), device="cuda")  # a slice of a larger buffer, as the head bucket is
        grp.all_reduce_grads(g[100:40_100])
        if not (bool((g[100:40_100] == float(sum(r + 1 for r in range(world)))).all()) and float(g[0]) == rank + 1
                and float(g[-1]) == rank + 1):
            bad.append(("slice", it))
    # the peers' copies are added IN RANK ORDER on every rank: values whose f32 sum depends on the order
    vals = [1e8, 1.0, -1e8, 0.5, 3.0, -0.25, 2e7, 1.0][:world]
    o = torch.full((257,), vals[rank], device="cuda")
    grp.all_reduce_sum(o)
    acc = np.float32(0.0)
    for v in vals:
        acc = np.float32(acc + np.float32(v))
    if not bool((o == float(acc)).all()):
        bad.append(("order", float(o[0]), float(acc)))
    big = torch.ones(1 << 19, device="cuda")  # 2 MiB: above the mailbox size -> the ordinary (staged gloo) path
    grp.all_reduce_sum(big)
    torch.cuda.synchronize()
    ok_big = bool((big == float(world)).all())
    from sample_factory_amd import lib
    lib.dp_oneshot_status(grp._os)
    grp.close()
    np.savez(os.path.join(out_dir, f"oneshot_rank{rank}.npz"), bad=len(bad), first=str(bad[:3]), ok_big=ok_big)
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_one_shot_exchange_between_two_ranks_on_one_device(tmp_path, world):
    """csrc/sf_dp.hip one-shot exchange: mailboxes mapped across PROCESSES with hipIpc, one kernel per rank and
    call.  Two (four) ranks sharing the box's GPU: f32 sums of 3 ... 77 777 elements (the conv bucket's size), f64 scalars, a
    slice of a larger buffer, 30 rounds over both slots — exact results on every rank; a sum whose f32 value depends on the
    order comes out as the RANK-ORDER sum everywhere; a bucket above the mailbox size takes the ordinary path; no wait timed
    out"""
    mp.spawn(_worker_oneshot, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for k in ("WORLD_SIZE", "RANK"):
        os.environ.pop(k, None)
    for r in range(world):
        z = np.load(tmp_path / f"oneshot_rank{r}.npz")
        assert int(z["bad"]) == 0 and bool(z["ok_big"]), (r, str(z["first"]))


def _worker_learner_oneshot(rank, world, port, out_dir, oneshot):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", SF_DP_BACKEND="gloo")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_synthetic_env
    from sample_factory_amd.train import make_runner
    register_env("synthetic_atari", make_synthetic_env)
    cfg = _cfg(32, dp_oneshot_bytes=(1 << 20) if oneshot else 0)
    cfg.batch_size, cfg.num_batches_per_epoch, cfg.num_epochs = 128, 2, 2
    cfg, runner = make_runner(cfg)
    runner.init()
    grp = runner.learner.group
    assert (grp._os is not None) == bool(oneshot)
    for _ in range(2):
        runner.iteration()
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"los{int(oneshot)}_rank{rank}.npz"), params=runner.learner.actor_critic.flat_params.cpu().numpy())
    grp.close()
    torch.distributed.destroy_process_group()


def test_learner_with_the_one_shot_exchange_equals_the_ring_path(tmp_path):
    """cfg.dp_oneshot_bytes: the learner's small SUM buckets (advantage / return moments, invalid counts, the conv bucket of
    the gradient) through the mailbox exchange.  With two ranks a sum has one order, so the weights after two datasets are
    bit-identical to the run on torch.distributed's all-reduce, and the replicas stay in lock-step"""
    out = {}
    for os_ in (1, 0):
        mp.spawn(_worker_learner_oneshot, args=(2, _free_port(), str(tmp_path), os_), nprocs=2, join=True)
        out[os_] = [np.load(tmp_path / f"los{os_}_rank{r}.npz")["params"] for r in range(2)]
    for k in ("WORLD_SIZE", "RANK"):
        os.environ.pop(k, None)
    np.testing.assert_array_equal(out[1][0], out[1][1])
    np.testing.assert_array_equal(out[1][0], out[0][0])

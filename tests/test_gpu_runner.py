"""Runner data plane and life-cycle on the GPU: the slab bookkeeping of BufferMgr / Batcher / RowLedger wired into the
rollout -> train loop (shared_buffers.py:152-239, batcher.py:170-234, rollout_worker.py:108-126), and the checkpoint
life-cycle of runner.py:170-176,207-230,685-698 + learner.py:300-386."""
import glob
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _synthetic_cfg(**over):
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_synthetic_env
    register_env("synthetic_atari", make_synthetic_env)
    base = dict(env="synthetic_atari", use_rnn=False, nonlinearity="relu", normalize_input=False, obs_scale=255.0,
                encoder_conv_architecture="convnet_atari", rollout=8, num_epochs=1, num_workers=1, num_envs_per_worker=1,
                worker_num_splits=1, async_rl=False, seed=2, serial_mode=True, synthetic_num_agents=64,
                summaries_every_train=True)
    base.update(over)
    return default_cfg(**base)


def test_sync_dataset_accumulated_from_two_rollouts():
    """batch_size * num_batches_per_epoch = 2 x (agents * rollout): the slab has two sampling slices, the Batcher merges
    them into one dataset, the learner trains once per two rollouts — both collected by the same policy version (sync
    mode: lag 0 at the first minibatch, tests/envs/mujoco/test_envpool_mujoco.py:99-114 in the reference)."""
    from sample_factory_amd.train import make_runner
    cfg, runner = make_runner(_synthetic_cfg(batch_size=512, num_batches_per_epoch=2))
    runner.init()
    assert runner.buffer_mgr.num_buffers == 128 and runner.traj["rewards"].shape == (128, 8)
    assert list(runner.buffer_mgr.traj_buffer_queue) == [slice(0, 64), slice(64, 128)]
    stats = runner.iteration()
    torch.cuda.synchronize()
    assert runner.sampling_rounds == 2 and stats["learner_env_steps"] == 1024 and runner.learner.train_step == 2
    tr = runner.traj
    assert (tr["policy_version"] == 0.0).all() and (tr["policy_id"] == 0).all()
    assert stats["train"]["version_diff_min"] == 2.0 and stats["train"]["version_diff_max"] == 2.0  # after the LAST minibatch
    # the second rollout continues the first one: obs[:, 0] of rows 64.. is obs[:, T] of rows 0..63
    assert torch.equal(tr["obs"]["obs"][64:, 0], tr["obs"]["obs"][:64, 8])
    import oracle
    np.testing.assert_array_equal(tr["obs"]["obs"][64:, 8].cpu().numpy().reshape(64, -1), oracle.synth_obs(64, 0, 28224, 2, 16))
    # every row went back to the free queue, in sampling-slice units
    assert sorted(s.start for s in runner.buffer_mgr.traj_buffer_queue) == [0, 64] and runner.batcher.in_flight == 0
    stats = runner.iteration()
    assert runner.sampling_rounds == 4 and stats["learner_env_steps"] == 2048
    assert (runner.traj["policy_version"] == 2.0).all()


def test_async_round_split_into_two_datasets():
    """agents * rollout = 2 x (batch_size * num_batches_per_epoch) (only legal in async mode, cfg/arguments.py:147-155):
    one sampling round yields two datasets, both handed to the learner (num_batches_to_accumulate = 2) while the next
    round is being collected; the second one sees the policy lag of the first one's SGD steps (parity trap 14)."""
    from sample_factory_amd.train import make_runner
    cfg, runner = make_runner(_synthetic_cfg(synthetic_num_agents=128, batch_size=256, num_batches_per_epoch=2,
                                             async_rl=True, serial_mode=False, num_batches_to_accumulate=2))
    runner.init()
    assert runner.buffer_mgr.num_buffers == 256 and runner.buffer_mgr.trajectories_per_training_iteration == 64
    assert runner.iteration() is None                   # round 0 only
    stats = runner.iteration()                          # round 1 || train(D0a), train(D0b)
    torch.cuda.synchronize()
    assert runner.sampling_rounds == 2 and runner.learner.train_step == 4 and stats["learner_env_steps"] == 1024
    assert runner.training_iteration_since_resume == 2
    assert stats["train"]["version_diff_min"] == 4.0  # second dataset: sampled 4 SGD steps before its last update
    assert runner.batcher.in_flight == 2 and len(runner._ready) == 2  # round 1's two datasets wait for the learner
    # sync mode rejects this shape the way the reference does: Runner.init() reports the invalid configuration and
    # returns ExperimentStatus.FAILURE (algo/runners/runner.py:527-528), it does not raise
    assert make_runner(_synthetic_cfg(synthetic_num_agents=128, batch_size=256, num_batches_per_epoch=2))[1].init() == 1


def test_async_accumulates_and_throttles():
    """async mode with a dataset of two sampling rounds: slab = max(2 x round, num_batches_to_accumulate x dataset) rows;
    the rollout stream keeps sampling into free slices while the learner trains, and pauses when none is free."""
    from sample_factory_amd.train import make_runner
    cfg, runner = make_runner(_synthetic_cfg(async_rl=True, serial_mode=False, batch_size=512, num_batches_per_epoch=2,
                                             num_batches_to_accumulate=2))
    runner.init()
    assert runner.buffer_mgr.num_buffers == 256 and runner.buffer_mgr.max_batches_to_accumulate == 2
    trained = 0
    for _ in range(9):
        stats = runner.iteration()
        trained += stats is not None
    torch.cuda.synchronize()
    assert runner.sampling_rounds == 9 and trained == 4 and runner.learner.env_steps == 4 * 1024
    assert torch.isfinite(runner.learner.actor_critic.flat_params).all()
    lag = runner.learner.last_summary
    assert lag["version_diff_min"] >= 1.0 and lag["version_diff_max"] <= cfg.max_policy_lag


def test_run_checkpoint_life_cycle(tmp_path):
    """Runner.run(): periodic save, milestone, best-policy save (after save_best_after steps, on the running mean of
    cfg.save_best_metric), final save on stop, config.json; a new Runner resumes from the latest checkpoint;
    restart_behavior=restart moves the old experiment aside; load_checkpoint_kind=best loads best_*."""
    from sample_factory_amd.algo.learning.learner import Learner
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.cartpole import make_cartpole_env
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.train import ExperimentStatus, make_runner
    register_env("CartPole-v1", make_cartpole_env)

    def cfg_(**over):
        base = dict(env="CartPole-v1", use_rnn=False, nonlinearity="tanh", normalize_input=True, encoder_mlp_layers=[32, 32],
                    rollout=16, batch_size=512, num_batches_per_epoch=2, num_epochs=1, num_workers=1, num_envs_per_worker=1,
                    worker_num_splits=1, async_rl=False, seed=3, serial_mode=True, cartpole_num_agents=64,
                    train_dir=str(tmp_path), experiment="life", train_for_env_steps=12 * 1024, save_every_sec=1e-4,
                    save_best_every_sec=1e-4, save_best_after=2048, save_milestones_sec=1e-4, keep_checkpoints=2)
        base.update(over)
        return default_cfg(**base)

    cfg, runner = make_runner(cfg_())
    assert runner.init() == ExperimentStatus.SUCCESS and runner.run() == ExperimentStatus.SUCCESS
    d = Learner.checkpoint_dir(cfg, 0)
    cps = Learner.get_checkpoints(d)
    assert len(cps) == 2 and cps[-1].endswith(f"checkpoint_{runner.learner.train_step:09d}_{runner.learner.env_steps}.pth")
    assert runner.learner.env_steps == 12 * 1024
    assert len(glob.glob(os.path.join(d, "milestones", "checkpoint_*.pth"))) >= 2
    best = Learner.get_checkpoints(d, "best_*")
    assert len(best) == 1 and "_reward_" in best[0] and runner.learner.best_performance > 0
    assert json.load(open(os.path.join(str(tmp_path), "life", "config.json")))["rollout"] == 16
    recs = [json.loads(l) for l in open(os.path.join(str(tmp_path), "life", ".summary", "0", "summaries.jsonl"))]
    assert recs[-1]["env_steps"] == 12 * 1024 and recs[-1]["perf/_fps"] > 0 and "train/loss" in recs[-1]
    assert "policy_stats/avg_reward" in recs[-1] and "train/kl_divergence" in recs[-1]
    steps, params = runner.learner.train_step, runner.learner.actor_critic.flat_params.clone()
    # resume (default)
    cfg2, r2 = make_runner(cfg_(train_for_env_steps=14 * 1024))
    r2.init()
    assert r2.learner.train_step == steps and r2.learner.env_steps == 12 * 1024
    assert torch.equal(r2.learner.actor_critic.flat_params, params) and r2.learner.best_performance == runner.learner.best_performance
    r2.run()
    assert r2.learner.env_steps == 14 * 1024 and r2.learner.train_step == steps + 4
    # best checkpoint instead of the latest
    cfg3, r3 = make_runner(cfg_(load_checkpoint_kind="best"))
    r3.init()
    cp = torch.load(Learner.get_checkpoints(d, "best_*")[-1], weights_only=False)
    assert r3.learner.train_step == cp["train_step"] and r3.learner.env_steps == cp["env_steps"]
    # restart: the old experiment directory is moved aside and training starts from scratch
    cfg4, r4 = make_runner(cfg_(restart_behavior="restart"))
    r4.init()
    assert r4.learner.train_step == 0 and os.path.isdir(os.path.join(str(tmp_path), "life_old0001"))
    assert not Learner.get_checkpoints(Learner.checkpoint_dir(cfg4, 0))


def test_example_script_written_against_sample_factory_runs(tmp_path):
    """examples/train_gym_env.py (imports only `sample_factory.*`, the reference's sf_examples/train_gym_env.py shape,
    BASELINE configs[0]) trains CartPole through run_rl and leaves a checkpoint + config.json behind"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "train_gym_env.py"), "--env=CartPole-v1",
                        "--use_rnn=False", "--serial_mode=True", "--async_rl=False", "--num_workers=1",
                        "--num_envs_per_worker=1", "--worker_num_splits=1", "--batch_size=512", "--rollout=32",
                        "--train_for_env_steps=4096", f"--train_dir={tmp_path}", "--experiment=example_gym_cartpole-v1",
                        "--seed=0", "--normalize_input=True", "--encoder_mlp_layers", "64", "64"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Collected {0: 4096}" in r.stdout
    d = os.path.join(str(tmp_path), "example_gym_cartpole-v1")
    assert os.path.exists(os.path.join(d, "config.json")) and glob.glob(os.path.join(d, "checkpoint_p0", "checkpoint_*.pth"))
    # ... and the evaluation script of the same shape (sf_examples/enjoy_gym_env.py) reads it back: saved config.json as
    # the base configuration, command-line flags on top
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "enjoy_gym_env.py"), "--env=CartPole-v1",
                        f"--train_dir={tmp_path}", "--experiment=example_gym_cartpole-v1", "--max_num_episodes=40",
                        "--eval_deterministic=True", "--env_agents=8"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    avg = float(r.stdout.split("Avg episode reward:")[-1].split()[0])
    assert 8.0 <= avg <= 500.0


def test_enjoy_deterministic_equals_host_argmax_loop(tmp_path):
    """enjoy(cfg) with --eval_deterministic (enjoy.py:170-172 argmax_actions) against an independent host loop over the
    same checkpoint and the same seeded env: logits from ActorCritic.forward, torch.argmax, episode returns summed on
    the host.  The average episode reward must be IDENTICAL (integer actions exact, same episode accounting), a
    sampled evaluation of the same policy must differ from it, and a missing experiment is reported as in the reference."""
    from sample_factory_amd.algo.learning.learner import Learner
    from sample_factory_amd.cfg.arguments import default_cfg, parse_full_cfg, parse_sf_args
    from sample_factory_amd.enjoy import enjoy
    from sample_factory_amd.envs.cartpole import CartPoleVecEnv, make_cartpole_env
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.model.model_factory import create_actor_critic
    from sample_factory_amd.train import make_runner
    register_env("cartpole_vec", make_cartpole_env)
    cfg = default_cfg(env="cartpole_vec", use_rnn=True, rnn_type="gru", rnn_size=32, recurrence=16, rollout=16,
                      encoder_mlp_layers=[32, 32], nonlinearity="tanh", normalize_input=True, batch_size=16 * 16,
                      num_batches_per_epoch=1, num_epochs=2, num_workers=1, num_envs_per_worker=1, worker_num_splits=1,
                      async_rl=False, serial_mode=True, seed=3, cartpole_num_agents=16, train_dir=str(tmp_path),
                      experiment="enjoy_me", train_for_env_steps=16 * 16 * 12, save_every_sec=100000)
    cfg, runner = make_runner(cfg)
    assert runner.init() == 0 and runner.run() == 0
    del runner
    argv = ["--env=cartpole_vec", f"--train_dir={tmp_path}", "--experiment=enjoy_me", "--eval_deterministic=True",
            "--max_num_frames=200", "--max_num_episodes=1000000", "--rollout=8"]
    parser, _ = parse_sf_args(argv, evaluation=True)
    parser.add_argument("--cartpole_num_agents", type=int, default=2)
    ecfg = parse_full_cfg(parser, argv)
    status, avg = enjoy(ecfg)
    assert status == 0
    # independent host loop: frames = 208 (the first multiple of rollout=8 past max_num_frames=200)
    ac = create_actor_critic(cfg, CartPoleVecEnv().observation_space, CartPoleVecEnv().action_space, torch.device("cuda", 0))
    ac.eval()
    ck = Learner.load_checkpoint(Learner.get_checkpoints(Learner.checkpoint_dir(cfg, 0)), "cpu")
    ac.load_state_dict(ck["model"])
    env = CartPoleVecEnv(num_agents=16, seed=3)
    o, _ = env.reset()
    h = torch.zeros((16, 32), device="cuda")
    ep_ret, total, episodes = np.zeros(16), 0.0, 0
    for _ in range(208):
        res = ac.forward({"obs": torch.from_numpy(o["obs"]).cuda()}, h)
        a = res["action_logits"].argmax(dim=1)
        h = res["new_rnn_states"].clone()
        o, rew, term, trunc, _ = env.step(a)
        ep_ret += rew
        done = term | trunc
        total += float(ep_ret[done].sum())
        episodes += int(done.sum())
        ep_ret[done] = 0.0
        h[torch.from_numpy(done).cuda()] = 0.0
    assert episodes > 0 and avg == total / episodes, (avg, total / episodes, episodes)
    # sampling instead of arg-max: the saved config is the base, the flag given here overrides it
    argv2 = [a for a in argv if not a.startswith("--eval_deterministic")] + ["--eval_deterministic=False"]
    parser, _ = parse_sf_args(argv2, evaluation=True)
    parser.add_argument("--cartpole_num_agents", type=int, default=2)
    _, avg_sampled = enjoy(parse_full_cfg(parser, argv2))
    assert avg_sampled != avg and avg_sampled > 0
    with pytest.raises(FileNotFoundError, match="Could not load saved parameters"):
        enjoy(default_cfg(env="cartpole_vec", train_dir=str(tmp_path), experiment="nope"))


@pytest.mark.parametrize("use_rnn", [False, True])
def test_losses_invariant_to_spliced_invalid_data(use_rnn):
    """The logic of the reference's tests/algo/test_learner.py:43-168 (TestValidMasks.test_losses_match) re-hosted on
    trajectories collected by this engine's sampler (the reference needs mujoco for its data): the reference-shaped
    `Learner._calculate_losses(dataset, num_invalids)` is deterministic, and splicing a rollout's worth of garbage
    marked policy_id = -1 into every trajectory leaves all four losses unchanged within the reference's tolerance."""
    import copy
    import random
    from sample_factory_amd.algo.utils.tensor_dict import TensorDict
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_synthetic_continuous_env
    from sample_factory_amd.train import make_runner
    register_env("synthetic_ant", make_synthetic_continuous_env)
    T = 8
    cfg = default_cfg(env="synthetic_ant", use_rnn=use_rnn, rnn_type="gru", rnn_size=32, recurrence=T if use_rnn else 1,
                      encoder_mlp_layers=[64, 64], nonlinearity="tanh", normalize_input=False, normalize_returns=False,
                      rollout=T, batch_size=32 * T, num_batches_per_epoch=1, num_epochs=1, num_workers=1,
                      num_envs_per_worker=1, worker_num_splits=1, async_rl=False, serial_mode=True, seed=0,
                      synthetic_num_agents=32, exploration_loss_coeff=0.001, kl_loss_coeff=0.1, adaptive_stddev=False)
    cfg, runner = make_runner(cfg)
    runner.init()
    learner = runner.learner
    runner._rollout_all(0.0)
    torch.cuda.synchronize()

    def clone(td):
        return TensorDict({k: clone(v) if isinstance(v, dict) else v.clone() for k, v in td.items()})

    og = clone(runner.traj)
    og["dones"][:] = False  # (the reference test data has no dones inside the spliced region either)

    def losses(dataset, invalids):
        _dist, policy_loss, exploration_loss, kl_old, kl_loss, value_loss, _summ = learner._calculate_losses(dataset, invalids)
        return dict(policy_loss=policy_loss, exploration_loss=exploration_loss, kl_old=kl_old, kl_loss=kl_loss,
                    value_loss=value_loss)

    dataset, experience_size, invalids = learner._prepare_batch(clone(og))
    assert invalids == 0 and experience_size == 32 * T
    res = prev = None
    for _ in range(3):  # sanity check: the same losses on the same batch
        res = losses(dataset, invalids)
        if prev is not None:
            for k in res:
                assert torch.equal(res[k], prev[k]), k
        prev = res
    # splice T steps of invalid data into every trajectory at a random position j
    random.seed(0)
    j = random.randint(0, T // 2)
    spliced = TensorDict()

    def splice(v, rollout_len):
        sh = v.shape
        extra = sh[1] - rollout_len  # 1 for [E, T+1, ...] tensors
        if v.dtype == torch.bool:
            junk = torch.randint(0, 2, (sh[0], T) + sh[2:], device=v.device).bool()
        else:
            junk = (torch.randint(-1, 1, (sh[0], T) + sh[2:], device=v.device) * 4242).to(v.dtype)
        return torch.cat([v[:, :j], junk, v[:, j:]], dim=1)

    for k, v in og.items():
        spliced[k] = TensorDict({kk: splice(vv, T) for kk, vv in v.items()}) if isinstance(v, dict) else splice(v, T)
    spliced["policy_id"][:, j:j + T] = -1
    spliced["dones"][:, j:j + T] = False
    spliced["time_outs"][:, j:j + T] = False
    inv_dataset, inv_size, invalids2 = learner._prepare_batch(spliced)
    assert inv_size == experience_size * 2 and invalids2 == experience_size
    inv = losses(inv_dataset, invalids2)
    for k in ("policy_loss", "exploration_loss", "kl_loss", "value_loss"):
        assert torch.allclose(res[k], inv[k], atol=0.02, rtol=0.02), (k, float(res[k]), float(inv[k]))
    # reference-typed returns (learner.py:586-669): a distribution OBJECT over the action parameters and the per-sample
    # KL(new || old) of the VALID samples, whose mean is the fused kernel's kl scalar
    dist, _pl, _el, kl_old, _kl, _vl, summ = learner._calculate_losses(dataset, invalids)
    assert dist.raw_logits.shape[0] == experience_size and dist.entropy().shape == (experience_size,)
    assert torch.isfinite(dist.log_prob(dataset.actions)).all() and dist.values.shape == (experience_size,)
    assert kl_old.shape == (experience_size,) and inv["kl_old"].shape == (experience_size,)   # invalid rows dropped
    assert abs(float(kl_old.mean()) - float(summ.kl_old_mean)) < 1e-6 + 1e-4 * abs(float(summ.kl_old_mean))
    assert abs(float(dist.entropy().mean()) - float(summ.entropy)) < 1e-5


def _mlp_ckpt_setup(tmp_path, golden):
    from sample_factory_amd.algo.utils.env_info import EnvInfo
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    g = golden("ref_checkpoint_mlp")
    E, T, A, nb = int(g["E"]), int(g["T"]), int(g["A"]), int(g["num_batches"])
    cfg = default_cfg(use_rnn=False, recurrence=1, nonlinearity="elu", normalize_input=True, encoder_mlp_layers=[32, 32],
                      rollout=T, batch_size=E * T // nb, num_batches_per_epoch=nb, num_epochs=2, seed=0, serial_mode=True,
                      train_dir=str(tmp_path), experiment="ckpt")
    env_info = EnvInfo(spaces.Dict({"obs": spaces.Box(-10, 10, (8,), np.float32)}), spaces.Discrete(A), E)
    return g, cfg, env_info, (E, T)


def test_resume_from_a_checkpoint_written_by_the_reference(tmp_path, golden):
    """tests/golden/ref_checkpoint_mlp.pth was written by the reference's Learner.save() (oracle/gen_golden.py ckpt):
    Learner.init() here resumes from it — progress counters, best_performance, weights, Adam moments, normaliser
    statistics — and the model reproduces the reference's outputs on a probe batch."""
    import shutil
    from sample_factory_amd.algo.learning.learner import Learner, ParameterServer
    g, cfg, env_info, _ = _mlp_ckpt_setup(tmp_path, golden)
    d = Learner.checkpoint_dir(cfg, 0)
    src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_checkpoint_mlp.pth")
    shutil.copy(src, os.path.join(d, str(g["file_name"])))
    pv = torch.zeros(1, dtype=torch.int32)
    learner = Learner(cfg, env_info, pv, 0, ParameterServer(0, pv))
    learner.init()
    assert learner.train_step == int(g["train_step"]) == 4 and learner.env_steps == int(g["env_steps"])
    assert learner.best_performance == 12.5 and int(pv[0]) == 4 and learner.adam_step_count == 4
    cp = torch.load(src, weights_only=False)
    ac = learner.actor_critic
    sd = ac.state_dict()
    for k, v in cp["model"].items():
        assert torch.equal(sd[k].cpu().reshape(v.shape), v), k
    m, v2 = ac.flat_to_ref(learner.exp_avg), ac.flat_to_ref(learner.exp_avg_sq)
    names = [n for n, _ in ac.ref_param_shapes()]
    for i, n in enumerate(names):
        assert torch.equal(m[n], cp["optimizer"]["state"][i]["exp_avg"]), n
        assert torch.equal(v2[n], cp["optimizer"]["state"][i]["exp_avg_sq"]), n
    ac.eval()
    res = ac.forward({"obs": torch.from_numpy(g["probe"]).cuda()}, None)
    np.testing.assert_allclose(res["action_logits"].cpu().numpy(), g["logits"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(res["values"].cpu().numpy(), g["values"], atol=2e-5, rtol=1e-4)
    np.testing.assert_array_equal(res["action_logits"].argmax(1).cpu().numpy(), g["logits"].argmax(1))


def test_write_a_checkpoint_for_the_reference(tmp_path, golden):
    """the other direction: train one dataset here, Learner.save(), and export the file + this model's outputs on the
    probe batch to gpurun_out/interop/ — committed as tests/golden/ours_checkpoint_mlp.{pth,npz}, which the CPU test
    test_reference_loads_a_checkpoint_written_here feeds to the REFERENCE's Learner.load_from_checkpoint."""
    import shutil
    from sample_factory_amd.algo.learning.learner import Learner, ParameterServer
    from sample_factory_amd.algo.utils.shared_buffers import alloc_trajectory_tensors
    g, cfg, env_info, (E, T) = _mlp_ckpt_setup(tmp_path, golden)
    pv = torch.zeros(1, dtype=torch.int32)
    learner = Learner(cfg, env_info, pv, 0, ParameterServer(0, pv))
    learner.init()
    batch = alloc_trajectory_tensors(env_info, E, T, 1, "cuda")
    for k in ["rnn_states", "actions", "action_logits", "log_prob_actions", "values", "policy_version", "rewards",
              "dones", "time_outs", "policy_id", "valids"]:
        batch[k].copy_(torch.from_numpy(g["in_" + k]))
    batch["obs"]["obs"].copy_(torch.from_numpy(g["in_obs_obs"]))
    learner.train(batch)
    assert learner.save()
    path = Learner.get_checkpoints(Learner.checkpoint_dir(cfg, 0))[-1]
    ac = learner.actor_critic
    ac.eval()
    res = ac.forward({"obs": torch.from_numpy(g["probe"]).cuda()}, None)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "interop")
    os.makedirs(out, exist_ok=True)
    shutil.copy(path, os.path.join(out, "ours_checkpoint_mlp.pth"))
    np.savez(os.path.join(out, "ours_checkpoint_mlp.npz"), file_name=os.path.basename(path), probe=g["probe"],
             logits=res["action_logits"].cpu().numpy(), values=res["values"].cpu().numpy(),
             train_step=learner.train_step, env_steps=learner.env_steps, argv=str(g["argv"]))
    cp = torch.load(path, weights_only=False)
    assert all(not t.is_cuda for t in cp["model"].values()) and cp["optimizer"]["state"][0]["exp_avg"].device.type == "cpu"


def test_host_env_async_with_sampler_thread_and_pitched_ingest():
    """BASELINE configs[2] shape at test size: a HOST vector env (numpy frames, as envpool returns them) in async mode ->
    the Runner samples on a thread of its own (the learner's launches are issued while the sampler waits for its env),
    frames reach the slab through ONE pitched H2D DMA per step (sf_h2d_rows): the slab holds exactly the env's frames."""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_host_frame_env
    from sample_factory_amd.train import make_runner
    register_env("host_atari", make_host_frame_env)
    cfg = default_cfg(env="host_atari", use_rnn=False, nonlinearity="relu", normalize_input=False, obs_scale=255.0,
                      encoder_conv_architecture="convnet_atari", rollout=8, batch_size=512, num_batches_per_epoch=2,
                      num_epochs=1, num_workers=1, num_envs_per_worker=2, worker_num_splits=2, async_rl=True,
                      serial_mode=False, seed=4, synthetic_num_agents=64, env_gpu_observations=False, env_gpu_actions=False,
                      env_workers_mode="inline")  # (this test: envs in this process, sampler THREAD; processes: below)
    cfg, runner = make_runner(cfg)
    runner.init()
    assert runner.threaded and all(sm.host_env for sm in runner.samplers) and len(runner.units) == 2
    trained = 0
    for _ in range(5):
        stats = runner.iteration()
        trained += stats is not None
    runner.stop_sampler_thread()
    torch.cuda.synchronize()
    assert trained == 5 and runner.learner.env_steps == 5 * 1024 and runner.sampling_rounds >= 5
    assert torch.isfinite(runner.learner.actor_critic.flat_params).all()
    assert runner.learner.last_summary["num_sgd_steps"] == 2
    # every frame in the slab is a frame of the env's ring, bit for bit, in step order (row block of instance 0)
    env = runner.envs[0]
    rows = runner._prev_rows[0]["obs"]["obs"]          # [64, T+1, 4, 84, 84] view of the slab
    host = rows.cpu().numpy()
    last = env.step_count                               # frames are ring[step % ring]
    for t in range(9):
        np.testing.assert_array_equal(host[:, t], env.ring[(last - 8 + t) % len(env.ring)])
    assert sum(sm.h2d_bytes for sm in runner.samplers) >= runner.sampling_rounds * 2 * 64 * 8 * 28224
    s = runner.episode_stats()
    assert s["episodes"] >= 0


@pytest.mark.parametrize("rnn_type", ["gru", "lstm"])
def test_host_env_async_sampler_thread_with_recurrent_core(rnn_type):
    """The default configuration class of the reference — async_rl + use_rnn with a HOST env — samples on a thread of
    its own while the learner thread runs `_prepare_batch`'s bootstrap forward through the SAME model object.  The
    one-step recurrent state a forward leaves behind is keyed by the caller's tag (`new_rnn_parts_of(tag)`): the sampler
    must store the state ITS forward produced, not the learner's bootstrap state ([128, H] here against the sampler's
    [64, H]: a mix-up raises in sf_rnn_store_state and kills the sampler thread).  The stored states obey the
    reference's carry rule (batched_sampling.py:332-335): zero after a done, |h| <= 1 otherwise."""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_host_frame_env
    from sample_factory_amd.train import make_runner
    register_env("host_atari", make_host_frame_env)
    cfg = default_cfg(env="host_atari", use_rnn=True, rnn_type=rnn_type, rnn_size=64, recurrence=8, nonlinearity="relu",
                      normalize_input=False, obs_scale=255.0, encoder_conv_architecture="convnet_atari", rollout=8,
                      batch_size=512, num_batches_per_epoch=2, num_epochs=1, num_workers=1, num_envs_per_worker=2,
                      worker_num_splits=2, async_rl=True, serial_mode=False, seed=5, synthetic_num_agents=64,
                      env_gpu_observations=False, env_gpu_actions=False, env_workers_mode="inline")
    cfg, runner = make_runner(cfg)
    runner.init()
    assert runner.threaded and runner.learner.actor_critic.rnn_kind is not None
    trained = 0
    for _ in range(8):
        trained += runner.iteration() is not None
    runner.stop_sampler_thread()
    torch.cuda.synchronize()
    assert trained == 8 and runner._thread_error is None
    ac = runner.learner.actor_critic
    assert set(ac._rnn_out) >= {"inf", "inf1", "boot"}          # sampler tags and the learner's tag kept apart
    # (one (h, c | None) pair per recurrent layer)
    assert ac._rnn_out["inf"][0][0].shape[0] == 64 and ac._rnn_out["boot"][0][0].shape[0] == 128
    assert torch.isfinite(ac.flat_params).all()
    H = 64
    for e in range(2):
        rows = runner._prev_rows[e]
        st, dn = rows["rnn_states"], rows["dones"]
        assert torch.isfinite(st).all() and float(st[:, :, :H].abs().max()) <= 1.0 + 1e-6
        nxt = st[:, 1:]                                          # state INPUT of step t+1
        assert float(nxt[dn].abs().max() if dn.any() else 0.0) == 0.0
        assert float(nxt[~dn].abs().max()) > 0.0


@pytest.mark.parametrize("async_rl", [False, True], ids=["sync", "async"])
def test_host_envs_in_worker_processes_double_buffered(async_rl):
    """The reference's default deployment for CPU envs (serial_mode=False): `num_workers` env worker PROCESSES x
    `num_envs_per_worker` instances, dealt to `worker_num_splits` = 2 splits that the Runner pipelines (while the workers
    step split A's envs the GPU runs split B's inference: rollout_worker.py:96-117).  The workers write their frames into
    shared pages which sf_h2d_rows DMAs into the slab in place: every frame in the slab must be the frame the env instance
    of that slab row produced at that step — rows ordered (split; worker, instance, agent)."""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import HostFrameVecEnv, make_host_frame_env
    from sample_factory_amd.train import make_runner
    register_env("host_atari", make_host_frame_env)
    W, K, n, T = 2, 4, 16, 8                                   # 2 workers x 4 instances x 16 agents = 128 envs, 2 splits of 64
    cfg = default_cfg(env="host_atari", use_rnn=False, nonlinearity="relu", normalize_input=False, obs_scale=255.0,
                      encoder_conv_architecture="convnet_atari", rollout=T, batch_size=512, num_batches_per_epoch=2,
                      num_epochs=1, num_workers=W, num_envs_per_worker=K, worker_num_splits=2, async_rl=async_rl,
                      serial_mode=False, seed=4, synthetic_num_agents=n, env_gpu_observations=False, env_gpu_actions=False)
    cfg, runner = make_runner(cfg)
    runner.init()
    try:
        assert runner.parallel_envs is not None and len(runner.envs) == 2 and runner.envs[0].num_agents == W * (K // 2) * n
        assert all(sm.async_env and sm.host_env for sm in runner.samplers)
        trained = 0
        for _ in range(4):
            trained += runner.iteration() is not None
        runner.stop_sampler_thread()
        torch.cuda.synchronize()
        assert trained >= 3 and torch.isfinite(runner.learner.actor_critic.flat_params).all()
        rounds = runner.sampling_rounds
        for split in range(2):
            rows = runner._prev_rows[split]["obs"]["obs"].cpu().numpy()   # [64, T+1, 4, 84, 84]: the split's last rollout
            per = K // 2
            for w in range(W):
                for j in range(per):
                    env_id = w * K + split * per + j
                    ring = HostFrameVecEnv(num_agents=n, seed=4 + env_id).ring
                    r0 = (w * per + j) * n
                    last = rounds * T                                   # env steps taken by every instance so far
                    for t in range(T + 1):
                        np.testing.assert_array_equal(rows[r0:r0 + n, t], ring[(last - T + t) % len(ring)],
                                                      err_msg=f"split {split} worker {w} instance {j} step {t}")
        direct = [sm._direct_ok.get("obs") for sm in runner.samplers]
        print("frames DMA'd from the workers' pages in place:", direct)
    finally:
        runner.close_envs()


def test_baseline_config0_two_single_agent_envs_serial_mode(tmp_path):
    """BASELINE.json configs[0] as written — "sf_examples/train_gym_env.py CartPole-v1, serial mode, 2 envs" — through the
    example script of the same shape: the env factory returns ONE gym-style env per instance (gym.make where gymnasium is
    installed, the bundled CartPoleEnv with the same API otherwise), `--num_envs_per_worker=2`, `--serial_mode=True`; the
    engine wraps the two single-agent envs as the reference does (one agent each, auto-reset on done).  The learner runs
    on the GPU: this engine has no CPU execution mode."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "train_gym_env.py"), "--env=CartPole-v1",
                        "--env_agents=0", "--use_rnn=False", "--serial_mode=True", "--async_rl=False", "--num_workers=1",
                        "--num_envs_per_worker=2", "--worker_num_splits=1", "--batch_size=64", "--rollout=32",
                        "--num_batches_per_epoch=1", "--train_for_env_steps=1280", f"--train_dir={tmp_path}",
                        "--experiment=config0", "--seed=0", "--encoder_mlp_layers", "64", "64"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Collected {0: 1280}" in r.stdout
    assert glob.glob(os.path.join(str(tmp_path), "config0", "checkpoint_p0", "checkpoint_*.pth"))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "enjoy_gym_env.py"), "--env=CartPole-v1",
                        f"--train_dir={tmp_path}", "--experiment=config0", "--max_num_episodes=5", "--env_agents=0"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert 8.0 <= float(r.stdout.split("Avg episode reward:")[-1].split()[0]) <= 500.0

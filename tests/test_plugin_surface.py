"""The Python half of the drop-in boundary: the `sample_factory.*` names a user's script imports resolve to
this engine, behave like the reference's, and the reference's OWN example scripts — staged unmodified by `make -C oracle
ref` — import, register their components and parse their arguments against it.  CPU only; the same scripts are EXECUTED
end to end on the GPU in tests/test_gpu_reference_scripts.py."""
import argparse
import ast
import importlib
import os
import sys
import zipfile

import numpy as np
import pytest
import torch

from conftest import EX_ZIP, staged_scripts

SCRIPTS = ["sf_examples/train_gym_env.py", "sf_examples/enjoy_gym_env.py",
           "sf_examples/train_custom_env_custom_model.py", "sf_examples/enjoy_custom_env_custom_model.py"]


def test_staged_scripts_are_the_reference_bytes():
    """where the reference is present (the build container) the staged archive equals it byte for byte"""
    if not (os.path.isdir("/root/reference/sf_examples") and os.path.isfile(EX_ZIP)):
        pytest.skip("needs /root/reference and the staged archive")
    with zipfile.ZipFile(EX_ZIP) as z:
        for name in SCRIPTS:
            assert z.read(name) == open(os.path.join("/root/reference", name), "rb").read(), name


def test_every_sample_factory_import_of_the_reference_scripts_resolves(tmp_path):
    d = staged_scripts(tmp_path)
    import sample_factory  # noqa: F401
    seen = 0
    for name in SCRIPTS:
        tree = ast.parse(open(os.path.join(d, name)).read())
        for node in ast.walk(tree):
            if isinstance(node, ast.ImportFrom) and node.module and node.module.startswith("sample_factory"):
                mod = importlib.import_module(node.module)
                for alias in node.names:
                    assert hasattr(mod, alias.name), f"{name}: from {node.module} import {alias.name}"
                    seen += 1
                assert mod.__name__.startswith("sample_factory_amd"), "the alias must hand out this engine's module"
    assert seen >= 15


# the plugin-surface names SURVEY.md §8b / VERDICT r4 list, wherever a user imports them from
SURFACE = {
    "sample_factory.model.encoder": ["Encoder", "MultiInputEncoder", "MlpEncoder", "ConvEncoder", "make_img_encoder",
                                     "default_make_encoder_func"],
    "sample_factory.model.core": ["ModelCore", "ModelCoreRNN", "ModelCoreIdentity", "default_make_core_func"],
    "sample_factory.model.decoder": ["Decoder", "MlpDecoder", "default_make_decoder_func"],
    "sample_factory.model.model_utils": ["nonlinearity", "fc_layer", "create_mlp", "ModelModule", "model_device",
                                         "get_rnn_size"],
    "sample_factory.algo.utils.torch_utils": ["calc_num_elements", "to_scalar", "masked_select", "init_torch_runtime",
                                              "inference_context", "to_torch_dtype", "synchronize"],
    "sample_factory.utils.typing": ["Config", "ObsSpace", "ActionSpace", "Env", "CreateEnvFunc", "PolicyID", "StatusCode"],
    "sample_factory.utils.utils": ["log", "str2bool", "is_module_available", "experiment_dir", "ensure_dir_exists",
                                   "cfg_file", "static_vars", "debug_log_every_n", "set_process_cpu_affinity"],
    "sample_factory.utils.gpu_utils": ["CUDA_ENVVAR", "set_global_cuda_envvars", "get_available_gpus", "gpus_for_process",
                                       "set_gpus_for_process"],
    "sample_factory.utils.attr_dict": ["AttrDict"],
    "sample_factory.utils.algo_version": ["ALGO_VERSION"],
    "sample_factory.algo.utils.gymnasium_utils": ["convert_space", "patch_non_gymnasium_env"],
    "sample_factory.algo.utils.make_env": ["make_env_func_batched", "is_multiagent_env", "get_multiagent_info"],
    "sample_factory.algo.utils.context": ["global_model_factory", "global_env_registry", "reset_global_context",
                                          "sf_global_context"],
    "sample_factory.algo.utils.misc": ["ExperimentStatus", "EPS", "EPISODIC", "TRAIN_STATS", "LEARNER_ENV_STEPS"],
    "sample_factory.envs.env_utils": ["register_env", "RewardShapingInterface", "TrainingInfoInterface",
                                      "find_training_info_interface", "find_wrapper_interface"],
    "sample_factory.envs.create_env": ["create_env"],
    "sample_factory.cfg.arguments": ["parse_sf_args", "parse_full_cfg", "load_from_checkpoint",
                                     "maybe_load_from_checkpoint", "checkpoint_override_defaults", "cfg_dict", "cfg_str"],
    "sample_factory.cfg.configurable": ["Configurable"],
    "sample_factory.train": ["run_rl", "make_runner"],
    "sample_factory.enjoy": ["enjoy"],
}


@pytest.mark.parametrize("module", sorted(SURFACE))
def test_plugin_surface_names(module):
    import sample_factory  # noqa: F401
    mod = importlib.import_module(module)
    for name in SURFACE[module]:
        assert hasattr(mod, name), f"{module}.{name}"


def test_model_building_helpers():
    from sample_factory.algo.utils.torch_utils import calc_num_elements, masked_select, to_scalar, to_torch_dtype
    from sample_factory.model.model_utils import create_mlp, fc_layer, model_device, nonlinearity
    from torch import nn
    cfg = argparse.Namespace(nonlinearity="elu")
    assert isinstance(nonlinearity(cfg), nn.ELU) and isinstance(nonlinearity(argparse.Namespace(nonlinearity="relu")), nn.ReLU)
    assert isinstance(nonlinearity(argparse.Namespace(nonlinearity="tanh")), nn.Tanh)
    with pytest.raises(Exception):
        nonlinearity(argparse.Namespace(nonlinearity="gelu"))
    mlp = create_mlp([16, 8], 5, nn.ReLU())
    assert [type(m) for m in mlp] == [nn.Linear, nn.ReLU, nn.Linear, nn.ReLU] and mlp(torch.zeros(3, 5)).shape == (3, 8)
    assert isinstance(create_mlp([], 5, nn.ReLU()), nn.Identity)
    assert fc_layer(3, 4, bias=False).bias is None
    conv = nn.Sequential(nn.Conv2d(1, 8, 3, stride=2), nn.ELU(), nn.Conv2d(8, 16, 2, stride=1), nn.ELU())
    assert calc_num_elements(conv, (1, 10, 10)) == 16 * 3 * 3     # the custom-model example's conv head
    assert model_device(conv) == torch.device("cpu") and model_device(nn.ReLU()) is None
    assert to_scalar(torch.tensor(2.5)) == 2.5 and to_scalar(3) == 3
    x, mask = torch.arange(6.0), torch.tensor([1, 0, 1, 0, 0, 1], dtype=torch.bool)
    assert torch.equal(masked_select(x, mask, 3), torch.tensor([0.0, 2.0, 5.0])) and masked_select(x, mask, 0) is x
    assert to_torch_dtype(np.uint8) == torch.uint8 and to_torch_dtype(np.float32) == torch.float32


def test_str2bool_and_logging_helpers(tmp_path):
    from sample_factory.utils.utils import (cfg_file, cores_for_worker_process, experiment_dir, is_module_available, log,
                                            str2bool)
    assert str2bool("True") is True and str2bool("false") is False and str2bool(True) is True
    with pytest.raises(argparse.ArgumentTypeError):
        str2bool("yes")     # the reference accepts true / false only
    assert is_module_available("numpy") and not is_module_available("surely_not_a_module_xyz")
    cfg = argparse.Namespace(train_dir=str(tmp_path / "td"), experiment="e1")
    assert experiment_dir(cfg, mkdir=False) == str(tmp_path / "td" / "e1") and not os.path.isdir(tmp_path / "td")
    assert cfg_file(cfg).endswith("e1/config.json") and os.path.isdir(tmp_path / "td" / "e1")
    assert log.name == "rl" and not log.propagate
    assert cores_for_worker_process(1, 4, 16) == [4, 5, 6, 7] and cores_for_worker_process(5, 8, 4) == [1]
    assert cores_for_worker_process(0, 3, 16) is None


def test_gpu_visibility_maps_to_hip_visible_devices(monkeypatch):
    from sample_factory.utils import gpu_utils
    assert gpu_utils.CUDA_ENVVAR == "HIP_VISIBLE_DEVICES"
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)
    monkeypatch.delenv("CUDA_VISIBLE_DEVICES", raising=False)
    gpu_utils.set_global_cuda_envvars(argparse.Namespace(device="cpu"))
    assert os.environ["HIP_VISIBLE_DEVICES"] == "" and gpu_utils.get_available_gpus() == []
    assert gpu_utils.gpus_for_process(0, 1) == []
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.setenv("CUDA_VISIBLE_DEVICES", "2,5")       # a user who set the CUDA name: the HIP name follows it
    gpu_utils.set_global_cuda_envvars(argparse.Namespace(device="gpu"))
    assert os.environ["HIP_VISIBLE_DEVICES"] == "2,5"
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "1,2,3")
    assert gpu_utils.get_available_gpus() == [1, 2, 3]
    # indices are relative to the visible set and wrap around (the reference's arithmetic)
    assert gpu_utils.gpus_for_process(0, 1) == [0] and gpu_utils.gpus_for_process(1, 1) == [1]
    assert gpu_utils.gpus_for_process(3, 1) == [0] and gpu_utils.gpus_for_process(1, 2) == [2, 0]
    assert gpu_utils.set_gpus_for_process(2, 1, "learner") == [2] and os.environ["HIP_VISIBLE_DEVICES"] == "3"
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "4,5,6,7")
    assert gpu_utils.gpus_for_process(0, 1, gpu_mask=[2, 3]) == [0]
    gpu_utils.set_gpus_for_process(1, 1, "inference", gpu_mask=[2, 3])
    assert os.environ["HIP_VISIBLE_DEVICES"] == "7"


def test_convert_space_duck_types():
    from sample_factory.algo.utils.gymnasium_utils import convert_space, patch_non_gymnasium_env
    from sample_factory_amd.envs import spaces

    class OldBox:       # another library's Box: only the fields
        low, high, shape, dtype = -1.0, 1.0, (3,), np.dtype(np.float32)

    class OldDiscrete:
        n = 5

    class OldDict:
        spaces = {"obs": OldBox(), "goal": OldDiscrete()}

    b = convert_space(OldBox())
    assert tuple(b.shape) == (3,) and np.dtype(b.dtype) == np.float32
    assert convert_space(OldDiscrete()).n == 5
    d = convert_space(OldDict())
    assert sorted(d.keys()) == ["goal", "obs"] and d["goal"].n == 5
    own = spaces.Box(0, 1, (2,))
    assert convert_space(own) is own or tuple(convert_space(own).shape) == (2,)
    with pytest.raises(ValueError):
        convert_space(object())

    class E:
        observation_space, action_space = OldBox(), OldDiscrete()
    e = patch_non_gymnasium_env(E())
    assert e.action_space.n == 5 and tuple(e.observation_space.shape) == (3,)


def _custom_cfg(**kw):
    from sample_factory_amd.cfg.arguments import default_cfg
    return default_cfg(**kw)


def test_user_encoder_subclass_in_the_default_actor_critic():
    """the reference's custom-model pattern (sf_examples/train_custom_env_custom_model.py:99-136): an `Encoder` subclass
    built from `nonlinearity(cfg)` and `calc_num_elements`, registered with the model factory — the engine builds the
    default core / decoder / heads around it with the reference's parameter paths and initialisation"""
    from sample_factory.algo.utils.context import global_model_factory
    from sample_factory.algo.utils.torch_utils import calc_num_elements
    from sample_factory.model.encoder import Encoder
    from sample_factory.model.model_utils import nonlinearity
    from sample_factory_amd.envs import spaces
    from sample_factory_amd.model.torch_policy import build_torch_actor_critic
    from torch import nn

    class CustomEncoder(Encoder):
        def __init__(self, cfg, obs_space):
            super().__init__(cfg)
            self.conv_head = nn.Sequential(nn.Conv2d(1, 8, 3, stride=2), nonlinearity(cfg), nn.Conv2d(8, 16, 2, stride=1),
                                           nonlinearity(cfg))
            self.norm = nn.LayerNorm(144)
            with torch.no_grad():
                self.norm.bias.fill_(0.5)
            self.conv_head_out_size = calc_num_elements(self.conv_head, obs_space["obs"].shape)

        def forward(self, obs_dict):
            return self.norm(self.conv_head(obs_dict["obs"]).view(-1, self.conv_head_out_size))

        def get_out_size(self):
            return self.conv_head_out_size

    f = global_model_factory()
    f.register_encoder_factory(lambda cfg, obs_space: CustomEncoder(cfg, obs_space))
    try:
        for init in ("orthogonal", "torch_default"):
            cfg = _custom_cfg(use_rnn=True, rnn_size=128, rnn_type="gru", policy_initialization=init)
            obs_space = spaces.Dict({"obs": spaces.Box(0, 1, (1, 10, 10))})
            m = build_torch_actor_critic(cfg, obs_space, spaces.Discrete(10), f)
            names = [n for n, _ in m.named_parameters()]
            assert "encoder.conv_head.0.weight" in names and "core.core.weight_ih_l0" in names
            assert "critic_linear.weight" in names and "action_parameterization.distribution_linear.bias" in names
            enc = m.encoder
            assert isinstance(enc, Encoder) and enc.cfg is cfg and enc.get_out_size() == 144
            assert enc.device_for_input_tensor("obs") == torch.device("cpu") and enc.type_for_input_tensor("obs") == torch.float32
            # actor_critic.py:73-96: every `.bias` Parameter is zeroed whatever the scheme, user layers included
            assert float(enc.norm.bias.detach().abs().max()) == 0.0 and float(enc.conv_head[0].bias.detach().abs().max()) == 0.0
            assert float(m.critic_linear.bias.detach().abs().max()) == 0.0
            w = enc.conv_head[2].weight.detach().reshape(16, -1)
            orth = torch.allclose(w @ w.T, torch.eye(16), atol=1e-5)
            assert orth == (init == "orthogonal")
            out = m({"obs": torch.rand(5, 1, 10, 10)}, torch.zeros(5, 128))
            assert out["values"].shape == (5,) and out["action_logits"].shape == (5, 10) and out["new_rnn_states"].shape == (5, 128)
    finally:
        f.reset()


def test_reference_scripts_import_register_and_parse(ref_scripts):
    """`import sf_examples.train_custom_env_custom_model` — the reference's file, unmodified — against this engine: module
    import, register_custom_components(), the two-pass argument parse with the script's extra flags and default overrides,
    the env factory and the encoder factory"""
    from sample_factory.algo.utils.context import global_env_registry, global_model_factory
    from sample_factory.model.encoder import Encoder
    mod = importlib.import_module("sf_examples.train_custom_env_custom_model")
    assert os.path.dirname(mod.__file__).startswith(ref_scripts)
    mod.register_custom_components()
    assert "my_custom_env_v1" in global_env_registry()
    assert global_model_factory().make_model_encoder_func is mod.make_custom_encoder
    cfg = mod.parse_custom_args(argv=["--algo=APPO", "--env=my_custom_env_v1", "--experiment=test_example"])
    assert cfg.rnn_size == 128 and cfg.custom_env_num_actions == 10 and cfg.custom_env_episode_len == 1000
    assert cfg.cli_args == dict(algo="APPO", env="my_custom_env_v1", experiment="test_example")
    ecfg = mod.parse_custom_args(argv=["--algo=APPO", "--env=my_custom_env_v1", "--experiment=test_example"], evaluation=True)
    assert ecfg.max_num_frames == int(1e9) and hasattr(ecfg, "eval_deterministic")
    env = global_env_registry()["my_custom_env_v1"]("my_custom_env_v1", cfg, None, None)
    obs, info = env.reset()
    assert obs.shape == (1, 10, 10) and obs.dtype == np.float32 and env.action_space.n == 10
    _, rew, term, trunc, _ = env.step(7)
    assert abs(rew - 0.07) < 1e-9 and not term and not trunc
    enc = mod.make_custom_encoder(cfg, env_obs_dict(env))
    assert isinstance(enc, Encoder) and enc.get_out_size() == 144

    gym_mod = importlib.import_module("sf_examples.train_gym_env")
    gym_mod.register_custom_components()
    assert "CartPole-v1" in global_env_registry()
    env = global_env_registry()["CartPole-v1"]("CartPole-v1", cfg, None, None)
    assert env.action_space.n == 2 and env.reset()[0].shape == (4,)
    importlib.import_module("sf_examples.enjoy_custom_env_custom_model")
    importlib.import_module("sf_examples.enjoy_gym_env")


def env_obs_dict(env):
    from sample_factory_amd.envs import spaces
    return spaces.Dict({"obs": env.observation_space})


def test_run_rl_reports_unsupported_devices_as_status(ref_scripts, tmp_path):
    """`--device=cpu` (BASELINE configs[0] says "CPU only") is parsed and then REFUSED loudly: run_rl returns
    ExperimentStatus.FAILURE like the reference's Runner.init() does for a configuration it cannot run
    (algo/runners/runner.py:521-541) — this engine has no CPU execution path by design.  Without a GPU the answer is
    the same status, not an exception."""
    from sample_factory.algo.utils.misc import ExperimentStatus
    from sample_factory.train import make_runner, run_rl
    mod = importlib.import_module("sf_examples.train_gym_env")
    mod.register_custom_components()
    argv = ["--algo=APPO", "--env=CartPole-v1", "--experiment=cpu_only", f"--train_dir={tmp_path}", "--device=cpu",
            "--serial_mode=True", "--num_workers=1", "--num_envs_per_worker=2"]
    cfg = mod.parse_custom_args(argv=argv)
    assert cfg.device == "cpu"
    assert run_rl(cfg) == ExperimentStatus.FAILURE
    if not torch.cuda.is_available():
        cfg = mod.parse_custom_args(argv=[a for a in argv if a != "--device=cpu"])
        before = dict(vars(cfg))
        cfg2, runner = make_runner(cfg)
        assert runner.init() == ExperimentStatus.FAILURE
        assert dict(vars(cfg)) == before, "the user's cfg object must not be mutated"


class _CurriculumEnv:
    """single-agent env implementing TrainingInfoInterface through a wrapper chain (module level: picklable for spawn)"""

    def __init__(self, full_env_name, cfg, env_config, render_mode=None):
        from sample_factory_amd.envs import spaces
        from sample_factory_amd.envs.env_utils import TrainingInfoInterface

        class Inner(TrainingInfoInterface):
            observation_space, action_space = spaces.Box(-1, 1, (3,)), spaces.Discrete(2)

            def __init__(self):
                TrainingInfoInterface.__init__(self)

            def reset(self, **kw):
                return np.zeros(3, np.float32), {}

            def step(self, a):
                # the observation reports what the env was told: visible to the parent through the shared pages
                steps = float(self.training_info.get("approx_total_training_steps", -1))
                return np.array([steps, 0, 0], np.float32), 0.0, False, False, {}

            def close(self):
                pass

        self.env = Inner()      # `.env` chain: find_training_info_interface has to unwrap one layer
        self.observation_space, self.action_space = self.env.observation_space, self.env.action_space

    def reset(self, **kw):
        return self.env.reset(**kw)

    def step(self, a):
        return self.env.step(a)

    def close(self):
        pass


def _make_curriculum_env(full_env_name, cfg=None, env_config=None, render_mode=None):
    return _CurriculumEnv(full_env_name, cfg, env_config, render_mode)


@pytest.mark.parametrize("inline", [True, False])
def test_training_info_reaches_envs_behind_the_parallel_view(inline):
    """envs behind a ParallelVecEnvView (worker processes or inline) receive approx_total_training_steps as
    the reference's rollout workers forward it (batched_sampling.py:352-355)"""
    from sample_factory_amd.algo.sampling.parallel_env import ParallelHostEnvs
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import set_training_info
    cfg = default_cfg(env="curriculum")
    pe = ParallelHostEnvs(cfg, "curriculum", _make_curriculum_env, 2, 2, num_splits=2, inline=inline)
    try:
        assert pe._training_info_instances == 4
        for v in pe.views:
            v.reset()
        for v in pe.views:      # what Runner._rollout_all does with every view before a rollout
            set_training_info(v, dict(approx_total_training_steps=12345))
        for v in pe.views:
            obs, *_ = v.step(np.zeros(v.num_agents, np.int32))
            assert np.all(obs["obs"][:, 0] == 12345.0)
    finally:
        pe.close()


def test_make_runner_resume_loads_saved_config_with_cli_overrides(tmp_path):
    """train.py:12-30 + cfg/arguments.py:227-275 of the reference: restart_behavior=resume (default) -> config.json of
    the experiment is the base, explicit command-line flags override it, new flags are added; a cfg built without the
    parser (no cli_args) is used as is"""
    import json
    from sample_factory_amd.cfg.arguments import default_cfg, parse_full_cfg, parse_sf_args
    from sample_factory_amd.train import make_runner
    d = tmp_path / "exp1"
    d.mkdir()
    json.dump(dict(env="e", experiment="exp1", train_dir=str(tmp_path), gamma=0.9, rollout=16, batch_size=256), open(d / "config.json", "w"))
    argv = ["--env=e", "--experiment=exp1", f"--train_dir={tmp_path}", "--rollout=8"]
    parser, _ = parse_sf_args(argv)
    cfg = parse_full_cfg(parser, argv)
    cfg2, runner = make_runner(cfg)
    assert cfg2.gamma == 0.9 and cfg2.rollout == 8 and cfg2.batch_size == 256 and cfg2.num_epochs == 1
    assert cfg.gamma == 0.99, "the caller's object is untouched"
    assert runner.cfg is cfg2
    cfg3 = default_cfg(env="e", experiment="exp1", train_dir=str(tmp_path))
    assert make_runner(cfg3)[0] is cfg3
    argv = ["--env=e", "--experiment=fresh", f"--train_dir={tmp_path}"]
    parser, _ = parse_sf_args(argv)
    cfg4, _ = make_runner(parse_full_cfg(parser, argv))
    assert cfg4.gamma == 0.99 and cfg4["experiment"] == "fresh"


# ---- the whole example tree of the reference (sf_examples/**: 72 names from 30 modules, inventory generated by
# oracle/gen_import_surface.py): every `from sample_factory... import ...` an env / model integration contains resolves here,
# except the modules below — subsystems DESIGN.md §7 / SURVEY.md §8 put outside the hot-path scope, each absent as a whole
OUT_OF_SCOPE_MODULES = {
    "sample_factory.launcher.launcher_utils": "experiment launcher (grid search over processes / slurm)",
    "sample_factory.launcher.run_description": "experiment launcher",
    "sample_factory.export_onnx": "ONNX export",
    "sample_factory.eval": "stand-alone evaluation front end (multi-process sampler without a learner)",
    "sample_factory.algo.sampling.sync_sampling_api": "sampler-only API (the reference's event-loop sampler used without a learner)",
    "sample_factory.envs.env_wrappers": "gym wrapper zoo (Atari / VizDoom / DMLab preprocessing; needs gymnasium)",
    "sample_factory.envs.pettingzoo_envs": "PettingZoo adapter (needs pettingzoo)",
}


def test_import_surface_against_the_reference_example_tree():
    import json
    import sample_factory  # noqa: F401
    inv = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_example_imports.json")))
    assert len(inv) >= 30 and sum(len(v) for v in inv.values()) >= 72
    resolved, skipped, missing = 0, 0, []
    for module, names in inv.items():
        if module in OUT_OF_SCOPE_MODULES:
            with pytest.raises(ModuleNotFoundError):  # absent as a whole: a clear error, not a half-working stand-in
                importlib.import_module(module)
            skipped += len(names)
            continue
        mod = importlib.import_module(module)
        assert mod.__name__.startswith("sample_factory_amd"), module
        for name, files in names.items():
            if name and not hasattr(mod, name):
                missing.append(f"from {module} import {name}   ({files[0]})")
            resolved += 1
    assert not missing, "\n".join(missing)
    assert resolved >= 45 and set(OUT_OF_SCOPE_MODULES) <= set(inv)

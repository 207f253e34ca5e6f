"""CPU suite: the C-ABI library loads without a GPU and exports every symbol include/sf_hip.h declares; host-side
logic (config surface, TensorDict, layouts, LR schedulers, minibatch planning)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "sf_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sf_[a-z0-9_]+)\s*\(", txt)))


def test_library_loads_and_exports_every_declared_symbol():
    from sample_factory_amd import build, lib
    build.build()
    L = lib.load()
    syms = header_symbols()
    assert len(syms) >= 26
    assert sorted(lib.SYMBOLS) == syms, "sample_factory_amd.lib.SYMBOLS out of date with include/sf_hip.h"
    for s in syms:
        assert hasattr(L, s), f"libsf_hip.so does not export {s}"
    assert L.sf_abi_version() == 19
    assert L.sf_selftest_host() == 0          # exact integer division used by every im2col address
    assert isinstance(L.sf_last_error(), bytes)


def test_argument_errors_are_reported_not_thrown():
    from sample_factory_amd import lib
    L = lib.load()
    rc = L.sf_gae_returns(None, None, None, None, None, None, 0, 0, ctypes.c_float(0.99), ctypes.c_float(0.95), 0, None, None, None)
    assert rc == -1 and b"sf_gae_returns" in L.sf_last_error()
    with pytest.raises(lib.SfHipError):      # CPU tensors are refused: there is no fallback path
        lib.grad_sumsq(torch.zeros(8), torch.zeros(1, dtype=torch.float64))
    # the data-parallel exchange: argument checks come before librccl is even looked for
    comm = ctypes.c_void_p()
    assert L.sf_dp_comm_create(None, 2, 0, ctypes.byref(comm)) == -1 and b"sf_dp_comm_create" in L.sf_last_error()
    assert L.sf_dp_comm_create(b"x" * 128, 2, 2, ctypes.byref(comm)) == -1 and b"rank=2" in L.sf_last_error()
    assert L.sf_allreduce_grads(None, None, ctypes.c_int64(4), None) == -1 and b"sf_allreduce_grads" in L.sf_last_error()
    assert L.sf_dp_allreduce_f64(ctypes.c_void_p(1), ctypes.c_void_p(1), ctypes.c_int64(4), 7, None) == -1
    with pytest.raises(lib.SfHipError):
        lib.dp_comm_create(b"short", 1, 0)


def test_struct_layouts_match_header():
    from sample_factory_amd import lib
    assert ctypes.sizeof(lib.sf_loss_cfg) == (8 + 1 + 8 + 1) * 4   # ... dense_adv, num_heads, head_n[8], old_values_T
    assert ctypes.sizeof(lib.sf_conv_desc) == 14 * 4
    assert [f[0] for f in lib.sf_conv_desc._fields_] == ["Cin", "H", "W", "Cout", "KH", "KW", "stride", "OH", "OW",
                                                          "in_u8", "relu", "traj_T", "sub_mean", "inv_scale"]


def test_cfg_defaults_equal_reference(golden_json):
    from sample_factory_amd.cfg.arguments import default_cfg, parse_full_cfg, parse_sf_args
    ref = golden_json("cfg_defaults")
    cfg = vars(default_cfg())
    from sample_factory_amd.cfg.arguments import ENGINE_FLAGS
    skip = {"train_dir", "command_line", "env", "experiment", "help"} | {f[0] for f in ENGINE_FLAGS}
    assert not {f[0] for f in ENGINE_FLAGS} & set(ref), "an engine flag shadows a reference flag"
    for k, v in cfg.items():
        if k in skip:
            continue
        assert k in ref, f"flag {k} does not exist in the reference"
        assert ref[k] == v, (k, ref[k], v)
    argv = ["--env=x", "--rollout=64", "--use_rnn=False", "--encoder_conv_mlp_layers", "256", "128", "--gamma=0.9"]
    p, _ = parse_sf_args(argv)
    c = parse_full_cfg(p, argv)
    assert (c.rollout, c.use_rnn, c.encoder_conv_mlp_layers, c.gamma) == (64, False, [256, 128], 0.9)


def test_verify_cfg_rules():
    from sample_factory_amd.algo.utils.env_info import EnvInfo
    from sample_factory_amd.cfg.arguments import default_cfg, preprocess_cfg
    ei = EnvInfo(None, None, 4096)
    ok = default_cfg(use_rnn=False, async_rl=False, num_workers=1, num_envs_per_worker=1, worker_num_splits=1, rollout=32,
                     batch_size=32768, num_batches_per_epoch=4)
    assert preprocess_cfg(ok, ei) and ok.recurrence == 1
    # a dataset of k rollouts (the Batcher accumulates them) is fine in sync mode, a non-multiple is not
    assert preprocess_cfg(default_cfg(use_rnn=False, async_rl=False, num_workers=1, num_envs_per_worker=1,
                                      worker_num_splits=1, rollout=32, batch_size=65536, num_batches_per_epoch=4), ei)
    # cfg/arguments.py:124-128: envs per worker must split evenly into the double-buffered sampling groups
    assert not preprocess_cfg(default_cfg(use_rnn=False, async_rl=True, num_workers=1, num_envs_per_worker=1,
                                          worker_num_splits=2), ei)
    assert not preprocess_cfg(default_cfg(use_rnn=False, async_rl=True, num_workers=1, num_envs_per_worker=2,
                                          worker_num_splits=2, num_policies=2), ei)
    bad = default_cfg(use_rnn=False, with_vtrace=True, normalize_returns=True, num_envs_per_worker=2)
    assert not preprocess_cfg(bad, ei)
    bad2 = default_cfg(use_rnn=False, async_rl=False, num_workers=1, num_envs_per_worker=1, worker_num_splits=1,
                       batch_size=1024)
    assert not preprocess_cfg(bad2, ei)


def test_tensor_dict_semantics():
    from sample_factory_amd.algo.utils.tensor_dict import TensorDict, clone_tensordict
    d = TensorDict(a=torch.zeros(4, 3), obs=TensorDict(x=torch.zeros(4, 3, 2)))
    step = d[:, 1]
    assert step["a"].shape == (4,) and step["obs"]["x"].shape == (4, 2)
    step[:] = dict(a=torch.ones(4), obs=dict(x=np.full((4, 2), 2.0, np.float32)))
    assert d["a"][:, 1].eq(1).all() and d["obs"]["x"][:, 1].eq(2).all() and d["a"][:, 0].eq(0).all()
    c = clone_tensordict(d)
    c["a"].fill_(5)
    assert d["a"].max() == 1


def test_trajectory_layout_matches_reference():
    from sample_factory_amd.algo.utils.env_info import EnvInfo
    from sample_factory_amd.algo.utils.shared_buffers import alloc_trajectory_tensors
    from sample_factory_amd.envs import spaces
    ei = EnvInfo(spaces.Dict({"obs": spaces.Box(0, 255, (4, 84, 84), np.uint8)}), spaces.Discrete(6), 8)
    t = alloc_trajectory_tensors(ei, 8, 32, 1, "cpu")
    shapes = {k: (tuple(v.shape), v.dtype) for k, v in t.items() if k != "obs"}
    assert t["obs"]["obs"].shape == (8, 33, 4, 84, 84) and t["obs"]["obs"].dtype == torch.uint8
    assert shapes == {"rnn_states": ((8, 33, 1), torch.float32), "actions": ((8, 32, 1), torch.float32),
                      "action_logits": ((8, 32, 6), torch.float32), "log_prob_actions": ((8, 32), torch.float32),
                      "values": ((8, 33), torch.float32), "policy_version": ((8, 32), torch.float32),
                      "rewards": ((8, 32), torch.float32), "dones": ((8, 32), torch.bool),
                      "time_outs": ((8, 32), torch.bool), "policy_id": ((8, 32), torch.int32),
                      "valids": ((8, 33), torch.bool)}
    assert (t["policy_id"] == -1).all() and t["dones"].all() and not t["valids"].any()


def test_weight_layout_round_trips():
    from sample_factory_amd import lib
    from sample_factory_amd.model.actor_critic import _Layer, _linear_desc
    g = torch.Generator().manual_seed(0)
    d1 = lib.sf_conv_desc(Cin=4, H=84, W=84, Cout=32, KH=8, KW=8, stride=4, OH=20, OW=20, in_u8=1, relu=1)
    L1 = _Layer("c1", d1, (32, 4, 8, 8), "conv_u8")
    w = torch.randn((32, 4, 8, 8), generator=g)
    k = L1.w_from_ref(w)
    assert k.shape == (256, 32) and torch.equal(L1.w_to_ref(k), w)
    assert k[(2 * 8 + 3) * 8 + 5, 7] == w[7, 2, 3, 5]                       # k = (c*KH + kh)*KW + kw
    d2 = lib.sf_conv_desc(Cin=32, H=20, W=20, Cout=64, KH=4, KW=4, stride=2, OH=9, OW=9, in_u8=0, relu=1)
    L2 = _Layer("c2", d2, (64, 32, 4, 4), "conv")
    w = torch.randn((64, 32, 4, 4), generator=g)
    k = L2.w_from_ref(w)
    assert k.shape == (512, 64) and torch.equal(L2.w_to_ref(k), w)
    assert k[(1 * 4 + 2) * 32 + 9, 11] == w[11, 9, 1, 2]                    # k = (kh*KW + kw)*Cin + c
    L3 = _Layer("fc", _linear_desc(3136, 512, True), (512, 3136), "linear_after_conv", first_fc_chw=(64, 7, 7))
    w = torch.randn((512, 3136), generator=g)
    k = L3.w_from_ref(w)
    assert torch.equal(L3.w_to_ref(k), w)
    assert k[(3 * 7 + 4) * 64 + 10, 99] == w[99, 10 * 49 + 3 * 7 + 4]       # NHWC flatten vs the reference's NCHW


def test_lr_schedulers():
    from sample_factory_amd.algo.learning.learner import get_lr_scheduler
    from sample_factory_amd.cfg.arguments import default_cfg
    s = get_lr_scheduler(default_cfg(lr_schedule="kl_adaptive_minibatch"))
    assert s.invoke_after_each_minibatch() and not s.invoke_after_each_epoch()
    assert s.update(1e-4, [0.1]) == pytest.approx(1e-4 / 1.5) and s.update(1e-4, [0.0]) == pytest.approx(1.5e-4)
    assert s.update(1e-4, [0.008]) == 1e-4 and s.update(1e-6, [1.0]) == 1e-6 and s.update(1e-2, [0.0]) == 1e-2
    e = get_lr_scheduler(default_cfg(lr_schedule="kl_adaptive_epoch", num_batches_per_epoch=2))
    assert e.invoke_after_each_epoch() and e.update(1e-4, [1.0, 0.0, 0.0]) == pytest.approx(1.5e-4)
    assert get_lr_scheduler(default_cfg()).update(3e-4, []) == 3e-4
    with pytest.raises(RuntimeError):
        get_lr_scheduler(default_cfg(lr_schedule="nope"))


def test_env_registry():
    from sample_factory_amd.envs.env_utils import create_env, register_env
    calls = []
    register_env("dummy_env", lambda name, cfg, env_config, render_mode: calls.append((name, cfg, env_config, render_mode)) or "ENV")
    assert create_env("dummy_env", 1, 2, None) == "ENV" and calls == [("dummy_env", 1, 2, None)]
    with pytest.raises(ValueError):
        create_env("not_registered")


def test_oracle_is_not_imported_by_the_product():
    """the product package must never reach into oracle/ (tests, smoke and bench's cpu_baseline leg only)"""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sample_factory_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "sf_oracle" in src or "/root/reference" in src:
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_slice_merger_and_buffer_mgr_match_reference(golden_json):
    """host-side slab bookkeeping vs traces recorded from the reference's SliceMerger / BufferMgr"""
    from sample_factory_amd.algo.learning.batcher import Batcher, RowLedger
    from sample_factory_amd.algo.utils.env_info import EnvInfo
    from sample_factory_amd.algo.utils.shared_buffers import BufferMgr
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    g = golden_json("host_logic")
    # the occupancy-map ledger replays the op traces recorded from the reference's SliceMerger: same rows handed out in
    # the same order, same number of rows held, same run starts after every operation (row granularity and, where the
    # trace never splits a unit, the coarser granule the Batcher itself would use)
    for ops in g["slice_merger_traces"]:
        sizes = [op[2] - op[1] for op in ops if op[0] == "merge"] + [op[1] for op in ops if op[0] != "merge"]
        for granule in {1, int(np.gcd.reduce(sizes))}:
            led = RowLedger(256, granule)
            for op in ops:
                if op[0] == "merge":
                    led.add(op[1], op[2])
                    assert led.total_num == op[3] and led.run_starts == op[4]
                else:
                    got = led.take(op[1], exact=op[0] == "exactly")
                    assert (None if got is None else [got.start, got.stop]) == op[2] and led.total_num == op[3]
    led = RowLedger(64, 8)
    led.add(8, 16)
    with pytest.raises(AssertionError):
        led.add(8, 16)      # rows cannot be handed in twice
    with pytest.raises(AssertionError):
        led.add(4, 12)      # nor in pieces that are not whole granules
    obs = spaces.Dict({"obs": spaces.Box(-1, 1, (4,), np.float32)})
    for m in g["buffer_mgr"]:
        kv = {}
        for a in m["argv"]:
            k, v = a[2:].split("=")
            kv[k] = {"True": True, "False": False}.get(v, int(v) if v.lstrip("-").isdigit() else v)
        cfg = default_cfg(**kv)
        bm = BufferMgr(cfg, EnvInfo(obs, spaces.Discrete(3), m["num_agents"]), "cpu", allocate=False)
        assert bm.buffers_per_device["cpu"] == m["buffers_for_device"] and bm.num_buffers == m["allocated"]
        assert bm.trajectories_per_training_iteration == m["trajectories_per_training_iteration"]
        assert bm.sampling_trajectories_per_iteration == m["sampling_trajectories_per_iteration"]
        assert bm.max_batches_to_accumulate == m["max_batches_to_accumulate"]
        q = [[s.start, s.stop] if isinstance(s, slice) else s for s in bm.traj_buffer_queue]
        assert q == m["queue"]
    # the slab protocol end to end: sampler slices -> datasets for the learner -> back to the free queue
    cfg = default_cfg(num_workers=1, num_envs_per_worker=1, worker_num_splits=2, batched_sampling=True, async_rl=True,
                      batch_size=1024, num_batches_per_epoch=2, rollout=8, num_batches_to_accumulate=2)
    bm = BufferMgr(cfg, EnvInfo(obs, spaces.Discrete(3), 256), "cpu", allocate=False)
    assert bm.num_buffers == 512 and bm.sampling_trajectories_per_iteration == 128 and len(bm.traj_buffer_queue) == 4
    b = Batcher(bm, cfg)
    s = [bm.get_free_slice() for _ in range(4)]
    assert bm.get_free_slice() is None                                   # every row in flight: the sampler must pause
    assert b.on_new_trajectories(s[1]) == [] and b.on_new_trajectories(s[3]) == []
    assert b.on_new_trajectories(s[0]) == [slice(0, 256)]                # adjacent slices merged into one dataset
    assert b.on_new_trajectories(s[2]) == [slice(256, 512)]
    assert b.on_training_batch_released(slice(0, 256)) == 2 and len(bm.traj_buffer_queue) == 2
    assert bm.get_free_slice() == slice(0, 128)


def test_env_plugin_interfaces_follow_the_wrapper_chain():
    """envs/env_utils.py:60-133 of the reference: the optional interfaces are looked up through the gym-style `.env`
    wrapper chain down to `.unwrapped`."""
    from sample_factory_amd.envs.env_utils import (RewardShapingInterface, TrainingInfoInterface, find_training_info_interface,
                                                   find_wrapper_interface, get_default_reward_shaping, set_reward_shaping,
                                                   set_training_info)

    class Base:
        def __init__(self):
            self.unwrapped = self

    class Shaped(Base, RewardShapingInterface):
        def __init__(self):
            Base.__init__(self)
            self.scheme, self.idx = dict(kill=1.0), None

        def get_default_reward_shaping(self):
            return dict(self.scheme)

        def set_reward_shaping(self, reward_shaping, agent_idx):
            self.scheme, self.idx = dict(reward_shaping), agent_idx

    class Wrapper:
        def __init__(self, env):
            self.env, self.unwrapped = env, env.unwrapped

    class Curriculum(Wrapper, TrainingInfoInterface):
        def __init__(self, env):
            Wrapper.__init__(self, env)
            TrainingInfoInterface.__init__(self)

    inner = Shaped()
    env = Wrapper(Curriculum(Wrapper(inner)))
    assert find_wrapper_interface(env, RewardShapingInterface) is inner
    assert get_default_reward_shaping(env) == dict(kill=1.0)
    set_reward_shaping(env, dict(kill=2.0), slice(0, 4))
    assert inner.scheme == dict(kill=2.0) and inner.idx == slice(0, 4)
    set_reward_shaping(env, None, 0)  # no-op
    assert inner.idx == slice(0, 4)
    iface = find_training_info_interface(env)
    assert isinstance(iface, Curriculum)
    set_training_info(iface, dict(approx_total_training_steps=123))
    assert iface.training_info["approx_total_training_steps"] == 123
    plain = Wrapper(Base())
    assert find_training_info_interface(plain) is None and get_default_reward_shaping(plain) is None
    set_training_info(None, dict(approx_total_training_steps=1))  # tolerated, as in the reference


def test_multi_input_architecture_matches_reference_on_cpu():
    """model/encoder.py:33-69 (MultiInputEncoder) as built for observation dicts with several keys: parameter names,
    shapes and — with the same seeded weights — logits / values of the REFERENCE model (tests/golden/model_fwd_multi.npz).
    The network of this path is plain torch, so the check runs without a GPU; per-key normaliser rules included."""
    import os
    import types
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    from sample_factory_amd.model.model_factory import create_actor_critic
    from sample_factory_amd.model.torch_policy import TorchObsNormalizer, TorchPolicyAdapter
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "model_fwd_multi.npz"), allow_pickle=True)
    cfg = default_cfg(encoder_conv_architecture="convnet_impala", nonlinearity="relu", obs_scale=255.0,
                      normalize_input=False, encoder_conv_mlp_layers=[32], encoder_mlp_layers=[16, 16], use_rnn=False,
                      normalize_returns=False)
    cfg.dp_world = 1
    obs_space = spaces.Dict({"obs": spaces.Box(0, 255, (4, 36, 36), np.uint8),
                             "measurements": spaces.Box(-1, 1, (5,), np.float32)})
    ac = create_actor_critic(cfg, obs_space, spaces.Discrete(6), torch.device("cpu"))
    assert isinstance(ac, TorchPolicyAdapter) and ac.multi_key and ac.obs_keys == ["measurements", "obs"]
    assert [(n, tuple(s)) for n, s in ac.ref_param_shapes()] == \
        [(str(n), tuple(eval(str(s)))) for n, s in zip(g["param_names"], g["param_shapes"])]
    from oracle.weights import seeded_state  # the deterministic weight generator the golden was made with
    st = seeded_state([(str(n), eval(str(s))) for n, s in zip(g["param_names"], g["param_shapes"])], int(g["param_seed"]))
    sd = {k: torch.from_numpy(v) for k, v in st.items()}
    ac.load_state_dict(sd, strict=True)
    ac.eval()
    res = ac.forward({"obs": torch.from_numpy(g["obs"]), "measurements": torch.from_numpy(g["measurements"])}, None)
    np.testing.assert_allclose(res["action_logits"].numpy(), g["action_logits"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(res["values"].numpy(), g["values"], atol=2e-5, rtol=1e-4)
    # normaliser rules: scale / mean shift only on "obs"; running statistics only for the keys listed
    c2 = types.SimpleNamespace(obs_subtract_mean=3.0, obs_scale=255.0, normalize_input=True, normalize_input_keys=["measurements"])
    n_obs, n_meas = TorchObsNormalizer(c2, (2,), "cpu", key="obs"), TorchObsNormalizer(c2, (2,), "cpu", key="measurements")
    assert n_obs.scale == 255.0 and n_obs.sub_mean == 3.0 and not n_obs.running
    assert n_meas.scale == 1.0 and n_meas.sub_mean == 0.0 and n_meas.running


@pytest.mark.parametrize("tag", ["ff", "gru", "lstm"])
def test_separate_weights_architecture_matches_reference_on_cpu(tag):
    """cfg.actor_critic_share_weights=False (model/actor_critic.py:198-334, ActorCriticSeparateWeights): parameter names,
    shapes and — with the same seeded weights — head / core / new state / logits / values of the REFERENCE model
    (tests/golden/model_fwd_separate_*.npz).  This path's network is plain torch (model/torch_policy.py), so the check runs
    without a GPU; the recurrent state is [actor | critic], twice the shared-weights width (model_utils.py:20-22)."""
    import os
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    from sample_factory_amd.model.actor_critic import get_rnn_size
    from sample_factory_amd.model.model_factory import create_actor_critic
    from sample_factory_amd.model.torch_policy import TorchPolicyAdapter
    from oracle.weights import seeded_state
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"model_fwd_separate_{tag}.npz"), allow_pickle=True)
    kw = dict(ff=dict(use_rnn=False), gru=dict(use_rnn=True, rnn_type="gru", rnn_size=12, recurrence=4),
              lstm=dict(use_rnn=True, rnn_type="lstm", rnn_size=10, recurrence=4, decoder_mlp_layers=[14]))[tag]
    cfg = default_cfg(actor_critic_share_weights=False, encoder_mlp_layers=[16, 12], nonlinearity="tanh",
                      normalize_input=False, normalize_returns=False, **kw)
    cfg.dp_world = 1
    obs_space = spaces.Dict({"obs": spaces.Box(-10, 10, (8,), np.float32)})
    ac = create_actor_critic(cfg, obs_space, spaces.Discrete(5), torch.device("cpu"))
    assert isinstance(ac, TorchPolicyAdapter)
    assert [(n, tuple(s)) for n, s in ac.ref_param_shapes()] == \
        [(str(n), tuple(eval(str(s)))) for n, s in zip(g["param_names"], g["param_shapes"])]
    st = seeded_state([(str(n), eval(str(s))) for n, s in zip(g["param_names"], g["param_shapes"])], int(g["param_seed"]))
    ac.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=True)
    ac.eval()
    rnn = torch.from_numpy(g["rnn_states"])
    assert rnn.shape[1] == get_rnn_size(cfg)
    m = ac.module
    with torch.no_grad():
        head = m.forward_head({"obs": torch.from_numpy(g["obs"])})
        core, new_rnn = m.forward_core(head, rnn)
        res = m.forward_tail(core)
    np.testing.assert_allclose(head.numpy(), g["head"], atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(core.numpy(), g["core"], atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(new_rnn.numpy(), g["new_rnn_states"], atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(res["action_logits"].numpy(), g["action_logits"], atol=2e-6, rtol=1e-5)
    np.testing.assert_allclose(res["values"].numpy(), g["values"], atol=2e-6, rtol=1e-5)
    # ... and through the adapter's interface (what the rollout runner / evaluation call)
    out = ac.forward({"obs": torch.from_numpy(g["obs"])}, rnn if cfg.use_rnn else None)
    np.testing.assert_allclose(out["action_logits"].numpy(), g["action_logits"], atol=2e-6, rtol=1e-5)
    if cfg.use_rnn:
        np.testing.assert_allclose(out["new_rnn_states"].numpy(), g["new_rnn_states"], atol=1e-6, rtol=1e-5)


def test_every_module_imports_without_a_gpu():
    """the whole host package must import on a CPU-only box (the HIP library is only dlopen'ed on first use)"""
    import importlib
    import pkgutil
    import sample_factory_amd
    names = [m.name for m in pkgutil.walk_packages(sample_factory_amd.__path__, "sample_factory_amd.")
             if not m.name.endswith("libsf_hip")]  # (the C-ABI shared object sits in the package directory)
    assert len(names) > 20
    for name in names:
        importlib.import_module(name)


def test_sample_factory_import_surface_resolves_to_this_engine():
    """a script written against `sample_factory.*` (examples/train_gym_env.py, the shape of the reference's
    sf_examples/train_gym_env.py) imports and parses its arguments here; without a GPU run_rl fails LOUDLY (no CPU
    fallback).  Subprocess: the oracle tooling imports the real reference under the same package name."""
    import subprocess
    import sys
    code = ("import sys; sys.argv=['x','--env=CartPole-v1','--use_rnn=False','--rollout=16'];"
            "import examples.train_gym_env as m, sample_factory, sample_factory_amd;"
            "from sample_factory.algo.learning.learner import Learner;"
            "from sample_factory.algo.utils.context import global_model_factory, global_env_registry;"
            "from sample_factory.algo.runners.runner import Runner, AlgoObserver;"
            "import sample_factory.train as t, sample_factory_amd.train as t2; assert t is t2;"
            "m.register_custom_components(); assert 'CartPole-v1' in global_env_registry();"
            "cfg = m.parse_custom_args(); assert cfg.rollout == 16 and cfg.env_agents == 16;"
            "print('OK', Learner.__module__)")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "OK sample_factory_amd.algo.learning.learner" in r.stdout, r.stderr[-1500:]
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train_gym_env.py"), "--env=CartPole-v1",
                            "--use_rnn=False", "--num_workers=1", "--num_envs_per_worker=1", "--worker_num_splits=1",
                            "--async_rl=False", "--batch_size=512", f"--train_dir={ROOT}/gpurun_out/td_cpu"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "no GPU visible" in r.stderr


def test_three_term_bf16_split_is_exact():
    """the operand split of csrc/sf_nn_u8.h (split3_bf16) restated in numpy: truncating an f32 to its upper 16 bits three
    times (value, then the two residuals) yields three bf16 numbers whose sum IS the f32 — for every sign, exponent and
    mantissa pattern; and every u8 value (and u8 - integer mean) is itself a bf16 number."""
    rng = np.random.default_rng(0)
    bits = rng.integers(0, 2 ** 32, size=200000, dtype=np.uint64).astype(np.uint32)
    w = bits.view(np.float32)
    w = w[np.isfinite(w) & (np.abs(w) > 1e-30) & (np.abs(w) < 1e30)]       # (terms below 2^-126 would be flushed by the MFMA)
    w = np.concatenate([w, np.float32([1.0, -1.0, 1 / 255.0, 0.1, 3.1415927, 1e-8, 65504.0, 0.0])])

    def trunc(v):
        return (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)

    hi = trunc(w)
    r1 = w - hi                      # exact: same exponent range, low 16 mantissa bits
    mid = trunc(r1)
    r2 = r1 - mid
    lo = trunc(r2)
    assert np.array_equal(r2, lo), "the third term is already a bf16 number"
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), w.astype(np.float64))
    for t in (hi, mid, lo):          # bf16 = an f32 whose low 16 bits are zero
        assert not np.any(t.view(np.uint32) & np.uint32(0xFFFF))
    px = np.arange(256, dtype=np.float32)
    for mean in (0.0, 128.0, 255.0):
        assert np.array_equal(trunc(px - np.float32(mean)), px - np.float32(mean))


def test_sample_factory_alias_keeps_the_real_module_specs():
    """`import sample_factory.x` returns the sample_factory_amd.x module object; the import machinery must not leave
    the alias spec on it (importlib.reload, __package__ == __spec__.parent), and `python -m sample_factory.<module>`
    finds the real module's code"""
    import importlib
    import subprocess
    import sys
    import sample_factory.cfg.arguments as a
    import sample_factory_amd.cfg.arguments as b
    import sample_factory.algo.learning as pkg
    assert a is b and a.__spec__.name == "sample_factory_amd.cfg.arguments" and a.__package__ == a.__spec__.parent
    assert pkg.__spec__.name == "sample_factory_amd.algo.learning" and list(pkg.__path__)
    assert importlib.reload(a) is a
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "sample_factory.cfg.arguments"], cwd=root, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_row_ledger_conserves_rows_under_random_traffic():
    """RowLedger against a brute-force model (a set of held rows) under random add / take traffic: no row is handed out
    twice or lost, a take stays inside ONE maximal run and prefers the oldest one that fits, runs() are maximal and
    disjoint, total_num is the model's size"""
    from hypothesis import given, settings, strategies as st
    from sample_factory_amd.algo.learning.batcher import RowLedger

    @settings(max_examples=150, deadline=None)
    @given(st.integers(1, 4), st.lists(st.tuples(st.booleans(), st.integers(0, 63), st.integers(1, 16), st.booleans()), max_size=60))
    def run(granule, ops):
        rows = 64 * granule
        led, held = RowLedger(rows, granule), set()
        for is_add, a, n, exact in ops:
            if is_add:
                start, stop = a * granule, min(rows, (a + n) * granule)
                if any(r in held for r in range(start, stop)):
                    continue  # (the ledger asserts on double adds; the callers never do that)
                led.add(start, stop)
                held.update(range(start, stop))
            else:
                before = led.runs()
                got = led.take(n * granule, exact=exact)
                fits = [r for r in before if (r[1] - r[0] >= n * granule) or not exact]
                if not fits:
                    assert got is None
                    continue
                oldest = min(fits, key=lambda r: r[2])
                assert got is not None and got.start == oldest[0] and got.stop == min(oldest[1], oldest[0] + n * granule)
                taken = set(range(got.start, got.stop))
                assert taken <= held
                held -= taken
            runs = led.runs()
            assert led.total_num == len(held) == sum(b - a for a, b, _ in runs)
            for (a0, b0, _), (a1, b1, _) in zip(runs, runs[1:]):
                assert b0 < a1, "runs must be maximal (no two touching) and in row order"
            assert all(set(range(a, b)) <= held for a, b, _ in runs)

    run()


def test_bench_numa_plan_splits_a_node_between_the_ranks_on_it():
    """bench.py pins every rank (and the env worker processes it starts) to its GPU's NUMA node; ranks whose GPUs share a
    node split its cores in local-rank order; an unknown node leaves the rank unpinned"""
    import bench
    node_cpus = {0: list(range(0, 32)) + list(range(64, 96)), 1: list(range(32, 64)) + list(range(96, 128))}
    nodes = [0, 0, 0, 0, 1, 1, 1, 1]
    shares = [bench.numa_plan(r, nodes, range(128), node_cpus) for r in range(8)]
    assert all(len(s) == 16 for s in shares)
    assert sorted(sum(shares[:4], [])) == sorted(node_cpus[0]) and sorted(sum(shares[4:], [])) == sorted(node_cpus[1])
    assert bench.numa_plan(0, [0, 1], [0, 1, 2, 3, 40], node_cpus) == [0, 1, 2, 3]      # the process's own mask is respected
    assert bench.numa_plan(1, [0, None], range(128), node_cpus) is None and bench.numa_plan(0, [-1], range(8), {}) is None
    assert bench._cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    # SMT siblings (cpu c and c + 64 share a core on this host's numbering) stay with one rank
    core_of = {c: (0, c % 32) for c in node_cpus[0]}
    a, b = (bench.numa_plan(r, [0, 0], range(128), node_cpus, core_of) for r in range(2))
    assert a == list(range(0, 16)) + list(range(64, 80)) and b == list(range(16, 32)) + list(range(80, 96))


def test_launch_program_recorder_on_a_stand_in_library(monkeypatch):
    """lib.record_launches / LaunchProgram without a GPU: a stand-in for the CDLL shows that launches are logged with the
    arguments they were called with AND executed, queries pass through unlogged, a ctypes cell is read again at every
    replay, calls that read host state mark the program, and a failing replayed call raises like the wrapper would."""
    import ctypes as C
    from sample_factory_amd import lib

    class Fake:
        def __init__(self):
            self.log, self.fail = [], False

        def sf_thing(self, *a):
            self.log.append(("thing",) + tuple(x.value if hasattr(x, "value") else x for x in a))
            return 7 if self.fail else 0

        def sf_thing_supported(self, *a):
            self.log.append(("query",))
            return 1

        def sf_h2d_rows(self, *a):
            return 0

        def sf_last_error(self):
            return b"stand-in failure"

    fake = Fake()
    monkeypatch.setattr(lib, "_lib", fake)
    cell = C.c_uint32(4)
    assert lib.u32(cell) is cell and lib.f(C.c_float(2.0)).value == 2.0 and lib.u32(2 ** 32 + 3).value == 3
    with lib.record_launches() as p:
        assert lib.load().sf_thing_supported(1) == 1
        assert lib.load().sf_thing(lib.u32(cell), 9) == 0
    assert lib.load() is fake
    assert fake.log == [("query",), ("thing", 4, 9)] and [c[2] for c in p.calls] == ["sf_thing"] and p.unsafe is None
    cell.value = 5
    p.replay()
    assert fake.log[-1] == ("thing", 5, 9) and len(fake.log) == 3
    fake.fail = True
    with pytest.raises(lib.SfHipError, match="sf_thing failed \\(7\\): stand-in failure"):
        p.replay()
    with lib.record_launches() as q:
        lib.load().sf_h2d_rows(0)
    assert q.unsafe == "sf_h2d_rows" and q.calls == []

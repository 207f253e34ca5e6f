"""Launch programs of the rollout step (lib.LaunchProgram, BatchedVectorEnvRunner._program): the library calls of step t are
recorded once and replayed with one foreign call per launch.  A replayed run must be the SAME run: every slab leaf, the
parameters and the episode statistics bit for bit equal to a run that goes through the wrappers every step — for the data
path of sample_factory/algo/sampling/batched_sampling.py:298-388 nothing but the host time may change."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cfg(kind, **over):
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs import synthetic
    register_env("synthetic_atari", synthetic.make_synthetic_env)
    register_env("synthetic_ant", synthetic.make_synthetic_continuous_env)
    register_env("synthetic_tuple", synthetic.make_synthetic_tuple_env)
    register_env("dict_bandit", synthetic.make_dict_obs_bandit_env)
    common = dict(rollout=8, num_epochs=1, num_workers=1, num_envs_per_worker=1, worker_num_splits=1, async_rl=False, seed=3,
                  serial_mode=True, num_batches_per_epoch=2)
    if kind == "conv_discrete":        # BASELINE configs[1] in miniature
        base = dict(env="synthetic_atari", use_rnn=False, nonlinearity="relu", normalize_input=False, obs_scale=255.0,
                    encoder_conv_architecture="convnet_atari", synthetic_num_agents=64, batch_size=256)
    elif kind == "conv_normalized":    # normalize_input=True: conv1's loader reads the published moment tables
        base = dict(env="synthetic_atari", use_rnn=False, nonlinearity="relu", normalize_input=True, obs_scale=255.0,
                    encoder_conv_architecture="convnet_atari", synthetic_num_agents=64, batch_size=256)
    elif kind in ("mlp_lstm_box", "mlp_gru_box", "mlp_lstm2_box"):  # BASELINE configs[4] in miniature (+ GRU, stacked)
        base = dict(env="synthetic_ant", use_rnn=True, rnn_type="gru" if "gru" in kind else "lstm", rnn_size=64,
                    rnn_num_layers=2 if "lstm2" in kind else 1, nonlinearity="tanh", normalize_input=True,
                    encoder_mlp_layers=[64, 64], recurrence=8, synthetic_num_agents=128, batch_size=512, kl_loss_coeff=0.1,
                    with_vtrace=True, normalize_returns=False, shuffle_minibatches=False, adaptive_stddev=False)
    elif kind == "conv_tuple_mixed":   # Tuple(Discrete(6), Box(2), Discrete(3)): the env reads the slab's action row
        base = dict(env="synthetic_tuple", use_rnn=False, nonlinearity="relu", normalize_input=False, obs_scale=255.0,
                    encoder_conv_architecture="convnet_atari", synthetic_num_agents=64, batch_size=256,
                    synthetic_head_sizes=(6, -2, 3))
    elif kind == "separate_gru_box":   # actor / critic towers on one flat buffer; state rows [actor | critic]
        base = dict(env="synthetic_ant", use_rnn=True, rnn_type="gru", rnn_size=64, nonlinearity="tanh", normalize_input=True,
                    encoder_mlp_layers=[64, 64], recurrence=8, synthetic_num_agents=128, batch_size=512,
                    actor_critic_share_weights=False, normalize_returns=True, adaptive_stddev=False)
    elif kind == "dict_multikey_gru":  # image + vector keys on the native towers, a torch-stepped device env (no step_into)
        base = dict(env="dict_bandit", use_rnn=True, rnn_type="gru", rnn_size=32, recurrence=8, nonlinearity="relu",
                    normalize_input=True, normalize_input_keys=["measurements"], obs_scale=255.0,
                    encoder_conv_architecture="convnet_impala", encoder_conv_mlp_layers=[32], encoder_mlp_layers=[32],
                    synthetic_num_agents=64, batch_size=256, normalize_returns=False)
    elif kind == "dict_multikey_separate":  # separate actor / critic weights x several keys: towers of towers
        base = dict(env="dict_bandit", use_rnn=True, rnn_type="lstm", rnn_size=32, recurrence=8, nonlinearity="relu",
                    normalize_input=True, normalize_input_keys=["measurements"], obs_scale=255.0,
                    encoder_conv_architecture="convnet_impala", encoder_conv_mlp_layers=[32], encoder_mlp_layers=[32],
                    synthetic_num_agents=64, batch_size=256, normalize_returns=True, actor_critic_share_weights=False)
    else:
        raise KeyError(kind)
    base.update(common)
    base.update(over)
    return default_cfg(**base)


def _run(kind, programs: bool, iters: int, **over):
    from sample_factory_amd import lib
    from sample_factory_amd.train import make_runner
    old = lib.LAUNCH_PROGRAMS
    lib.LAUNCH_PROGRAMS = programs
    try:
        cfg, runner = make_runner(_cfg(kind, **over))
        runner.init()
        slabs = []
        for _ in range(iters):
            runner.iteration()
            torch.cuda.synchronize()
            slabs.append({k: v.clone() for k, v in _leaves(runner.traj)})
        ac = runner.learner.actor_critic
        out = dict(slabs=slabs, params=ac.flat_params.clone(), samplers=runner.samplers,
                   ep=[s.ep_stats.clone() for s in runner.samplers], steps=[s.global_step for s in runner.samplers])
    finally:
        lib.LAUNCH_PROGRAMS = old
    return out


def _leaves(tr, prefix=""):
    for k in tr.keys():
        v = tr[k]
        if isinstance(v, torch.Tensor):
            yield prefix + k, v
        else:
            yield from _leaves(v, prefix + k + ".")


def _same_run(a, b):
    for i, (sa, sb) in enumerate(zip(a["slabs"], b["slabs"])):
        assert sa.keys() == sb.keys()
        for k in sa:
            assert torch.equal(sa[k], sb[k]), f"iteration {i}: slab leaf {k} differs"
    assert torch.equal(a["params"], b["params"]), "parameters differ"
    for ea, eb in zip(a["ep"], b["ep"]):
        assert torch.equal(ea, eb), "episode statistics differ"
    assert a["steps"] == b["steps"]


@pytest.mark.parametrize("kind", ["conv_discrete", "conv_normalized", "mlp_lstm_box", "mlp_gru_box", "mlp_lstm2_box",
                                  "conv_tuple_mixed", "dict_multikey_gru", "separate_gru_box",
                                  "dict_multikey_separate"])
def test_replayed_rollouts_equal_the_wrapper_path(kind):
    """5 iterations (rollout + train each): first sight, recording, then replays — against the same run with programs off.
    The sampler's Philox step and the policy version travel through ctypes cells: a stale value would repeat actions /
    stamp an old version into the slab."""
    from sample_factory_amd import lib
    plain = _run(kind, False, 5)
    prog = _run(kind, True, 5)
    _same_run(plain, prog)
    for s in plain["samplers"]:
        assert s.program_replays == 0 and not s._progs
    for s in prog["samplers"]:
        T = s.T
        progs = [p for p in s._progs.values() if isinstance(p, lib.LaunchProgram)]
        # a policy and a record program per step (and slab slice); env outputs in fresh tensors: the policy program only
        per_step = 2 if (s.zero_copy or s.host_env) else 1
        assert len(progs) >= per_step * T and len(progs) % (per_step * T) == 0, (len(progs), sorted(k[:2] for k in s._progs))
        assert all(p.unsafe is None for p in progs)
        assert s.program_replays >= T, s.program_replays  # (the layout settles after the first training pass)
        pol = [p for k, p in s._progs.items() if k[0] == "policy"]
        names = [[c[2] for c in p.calls] for p in pol]
        assert all(n == names[0] for n in names) and names[0][-1].startswith("sf_sample_write_step"), names[0]
        # policy_version in the slab follows the learner: the replayed sampler wrote the CURRENT version
        assert (prog["slabs"][-1]["policy_version"] == plain["slabs"][-1]["policy_version"]).all()
        assert float(prog["slabs"][-1]["policy_version"].max()) > 0


def test_programs_follow_the_published_snapshot_in_async_mode():
    """async mode: inference reads weight snapshot `snap_read`, which flips with every publish; a program recorded against
    slot 0 must not run when slot 1 is current (launch_key carries the slot), and two slab slices alternate"""
    from sample_factory_amd import lib
    kw = dict(async_rl=True, serial_mode=False, num_batches_to_accumulate=2)
    plain = _run("conv_discrete", False, 8, **kw)
    prog = _run("conv_discrete", True, 8, **kw)
    # the rollout stream runs beside the learner: the SAME schedule of publishes is not guaranteed bit for bit between two
    # runs (host timing decides which snapshot a round reads), so compare what is schedule-independent
    for s in prog["samplers"]:
        progs = [(k, p) for k, p in s._progs.items() if isinstance(p, lib.LaunchProgram)]
        assert progs and s.program_replays > 0
        assert len({k[5] for k, _ in progs}) <= 2  # weights slot of the key
    assert prog["steps"] == plain["steps"]
    for k, v in prog["slabs"][-1].items():
        assert torch.isfinite(v.float()).all(), k


def test_recorder_mechanics():
    """lib.record_launches: launches are logged with converted arguments and run; queries are not logged; a ctypes cell is
    read at replay time; host-state calls poison the program; another thread keeps talking to the real library"""
    import ctypes as C
    import threading
    from sample_factory_amd import lib
    lib.load()
    B, H = 16, 8
    h = torch.arange(B * H, dtype=torch.float32, device="cuda").view(B, H)
    dones = torch.zeros(B, dtype=torch.bool, device="cuda")
    out = torch.zeros(B, H, device="cuda")
    with lib.record_launches() as p:
        assert lib.lstm_seq_supported(64, 512) in (True, False)      # a query: runs, is not recorded
        lib.rnn_store_state(h, None, dones, out)
        seen = []
        th = threading.Thread(target=lambda: seen.append(type(lib.load()).__name__))
        th.start(); th.join()
    assert seen == ["CDLL"] and [c[2] for c in p.calls] == ["sf_rnn_store_state"] and p.unsafe is None
    assert torch.equal(out, h)
    out.zero_(); dones[3] = True
    p.replay()
    torch.cuda.synchronize()
    want = h.clone(); want[3] = 0
    assert torch.equal(out, want)
    assert any(t is h for t in p.keep) and any(t is out for t in p.keep)
    # a cell: the sampler's step counter
    logits = torch.randn(B, 8, device="cuda"); vals = torch.zeros(B, device="cuda")
    T = 2
    tr = dict(a=torch.zeros(B, T, 1, device="cuda"), lg=torch.zeros(B, T, 6, device="cuda"), lp=torch.zeros(B, T, device="cuda"),
              v=torch.zeros(B, T + 1, device="cuda"), pv=torch.zeros(B, T, device="cuda"))
    env_a = torch.zeros(B, dtype=torch.int32, device="cuda")
    step, ver = C.c_uint32(5), C.c_float(1.0)
    with lib.record_launches() as q:
        lib.sample_write_step(logits, 8, vals, 1, B, 6, T, 0, 11, step, 0, ver, False, tr["a"], tr["lg"], tr["lp"], tr["v"],
                              tr["pv"], env_a)
    a5 = tr["a"][:, 0, 0].clone()
    step.value, ver.value = 6, 3.0
    q.replay()
    a6 = tr["a"][:, 0, 0].clone()
    lib.sample_write_step(logits, 8, vals, 1, B, 6, T, 1, 11, 6, 0, 3.0, False, tr["a"], tr["lg"], tr["lp"], tr["v"], tr["pv"], env_a)
    torch.cuda.synchronize()
    assert torch.equal(a6, tr["a"][:, 1, 0]) and (tr["pv"] == 3.0).all() and not torch.equal(a5, a6)
    # host-state calls and explicit marks poison a recording
    with lib.record_launches() as r:
        lib.recording_unsafe("a torch op")
    assert r.unsafe == "a torch op"
    with pytest.raises(lib.SfHipError):
        with lib.record_launches():
            with lib.record_launches():
                pass
    assert type(lib.load()).__name__ == "CDLL"

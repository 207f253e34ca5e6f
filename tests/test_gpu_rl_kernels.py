"""GPU parity tests for the HBM/latency-bound kernels, through the C ABI (sample_factory_amd.lib -> libsf_hip.so),
against (a) golden vectors produced by the reference itself and (b) the CPU oracle on seeded inputs at full sizes.
Tolerances: integers / masks / indices exact; advantages & returns <= 1e-5 (north star: 1e-4); losses/grads fp32."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def dev(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


@pytest.fixture(scope="module")
def lib():
    from sample_factory_amd import lib as L
    L.load()
    return L


def run_gae(lib, rewards, dones, values, valids, gamma, lam, rms=None, time_outs=None, bootstrap=False):
    r, d, v, va = dev(rewards, torch.float32), dev(dones, torch.bool), dev(values, torch.float32), dev(valids, torch.bool)
    to = dev(time_outs, torch.bool) if time_outs is not None else None
    st = dev(rms, torch.float64) if rms is not None else None
    adv, ret = torch.empty_like(r), torch.empty_like(r)
    lib.gae_returns(r, d, to, v, va, st, gamma, lam, bootstrap, adv, ret)
    torch.cuda.synchronize()
    return adv.cpu().numpy(), ret.cpu().numpy(), r.cpu().numpy()


def test_gae_golden(lib, golden):
    g = golden("gae")
    for i in range(int(g["num_cases"])):
        adv, _, _ = run_gae(lib, g[f"c{i}_rewards"], g[f"c{i}_dones"], g[f"c{i}_values"], g[f"c{i}_valids"],
                            float(g[f"c{i}_gamma"]), float(g[f"c{i}_lambda"]))
        np.testing.assert_allclose(adv, g[f"c{i}_adv"], rtol=0, atol=1e-5)


@pytest.mark.parametrize("E,T", [(4096, 32), (77, 5), (130, 128), (1, 1), (64, 33), (4096, 128)])
def test_gae_vs_oracle(lib, E, T):
    rng = np.random.default_rng(E * 1000 + T)
    rewards = rng.standard_normal((E, T)).astype(np.float32)
    dones = rng.random((E, T)) < 0.05
    values = rng.standard_normal((E, T + 1)).astype(np.float32)
    valids = rng.random((E, T + 1)) > 0.1
    valids[:, -1] = valids[:, -2]
    adv, ret, _ = run_gae(lib, rewards, dones, values, valids, 0.99, 0.95)
    ref = oracle.gae(rewards, dones, values, valids, 0.99, 0.95)
    np.testing.assert_allclose(adv, ref, rtol=0, atol=1e-6)
    np.testing.assert_allclose(ret, ref + valids[:, :-1] * values[:, :-1], rtol=0, atol=1e-6)
    # the wrapper with the reference's signature
    from sample_factory_amd.algo.utils.rl_utils import gae_advantages
    a2 = gae_advantages(dev(rewards), dev(dones), dev(values), dev(valids), 0.99, 0.95).cpu().numpy()
    np.testing.assert_array_equal(a2, adv)


def _kv(argv):
    return {t[2:].split("=", 1)[0]: t[2:].split("=", 1)[1] for t in str(argv).split() if t.startswith("--") and "=" in t}


@pytest.mark.parametrize("case", ["ff_default", "ff_invalids", "ff_bootstrap_nonorm", "ff_continuous"])
def test_prepare_batch_golden(lib, golden, case):
    """K13 + K10 + K11 + K12 chained exactly as Learner._prepare_batch chains them, vs the reference's output."""
    g = golden("learner_" + case)
    kv = _kv(g["argv"])
    norm = kv.get("normalize_returns", "True") == "True"
    boot = kv.get("value_bootstrap", "False") == "True"
    E, T = g["in_rewards"].shape
    values = g["in_values"].copy()
    values[:, -1] = g["bootstrap_values"]
    pid, pver = dev(g["in_policy_id"], torch.int32), dev(g["in_policy_version"], torch.float32)
    valids = torch.zeros((E, T + 1), dtype=torch.bool, device="cuda")
    actions, logp = dev(g["in_actions"], torch.float32), dev(g["in_log_prob_actions"], torch.float32)
    ninv = torch.zeros(1, dtype=torch.int32, device="cuda")
    na = actions.numel() // (E * T)
    lib.valid_mask(pid, pver, valids, actions, na, logp, 0, int(g["train_step"]), int(kv.get("max_policy_lag", 1000)), ninv)
    assert int(ninv.item()) == int(g["pb_num_invalids"])
    np.testing.assert_array_equal(valids.cpu().numpy(), g["out_valids_full"])
    np.testing.assert_array_equal(actions.cpu().numpy().reshape(g["pb_actions"].shape), g["pb_actions"])
    np.testing.assert_array_equal(logp.cpu().numpy().reshape(-1), g["pb_log_prob_actions"])
    rms = g["in_rms"] if norm else None
    adv, ret, rew = run_gae(lib, g["in_rewards"], g["in_dones"], values, valids.cpu().numpy(), 0.99, 0.95, rms=rms,
                            time_outs=g["in_time_outs"], bootstrap=boot)
    np.testing.assert_allclose(rew, g["out_rewards"], atol=1e-6)
    np.testing.assert_allclose(adv.reshape(-1), g["pb_advantages"], atol=1e-5)
    if norm:
        from sample_factory_amd.algo.utils.running_mean_std import RunningMeanStdInPlace
        r = RunningMeanStdInPlace((1,), "cuda")
        r.stats.copy_(torch.as_tensor(g["in_rms"]))
        rt = dev(ret.reshape(-1))
        r(rt)
        np.testing.assert_allclose(r.stats.cpu().numpy(), g["out_rms"], rtol=1e-6)
        ret = rt.cpu().numpy()
    np.testing.assert_allclose(ret.reshape(-1), g["pb_returns"], atol=1e-5)


def test_rms_golden(lib, golden):
    from sample_factory_amd.algo.utils.running_mean_std import RunningMeanStdInPlace
    g = golden("rms")
    r = RunningMeanStdInPlace((1,), "cuda")
    for i in range(int(g["num_steps"])):
        x = dev(g[f"s{i}_x"])
        r(x)
        np.testing.assert_allclose(r.stats.cpu().numpy(), g[f"s{i}_stats"], rtol=2e-6)
        np.testing.assert_allclose(x.cpu().numpy(), g[f"s{i}_normalized"], atol=2e-6, rtol=1e-6)
        z = dev(g[f"s{i}_z"])
        r(z, denormalize=True)
        np.testing.assert_allclose(z.cpu().numpy(), g[f"s{i}_denormalized"], atol=1e-5, rtol=1e-6)
    r.eval()
    x = dev(g["eval_x"])
    r(x)
    np.testing.assert_allclose(x.cpu().numpy(), g["eval_normalized"], atol=2e-6)
    np.testing.assert_allclose(r.stats.cpu().numpy(), g["eval_stats"], rtol=2e-6)  # eval mode: stats frozen
    sd = r.state_dict("returns_normalizer.")
    assert set(sd) == {"returns_normalizer.running_mean", "returns_normalizer.running_var", "returns_normalizer.count"}
    assert sd["returns_normalizer.count"].dtype == torch.float64


def _loss_cfg(lib, kv, continuous, dense_adv=False):
    expl = kv.get("exploration_loss", "entropy")
    coeff = float(kv.get("exploration_loss_coeff", 0.003))
    return lib.sf_loss_cfg(clip_ratio=0.1, clip_value=1.0, value_loss_coeff=0.5, exploration_coeff=coeff,
                           kl_coeff=float(kv.get("kl_loss_coeff", 0.0)),
                           exploration_kind=0 if coeff == 0 else (1 if expl == "entropy" else 2),
                           action_kind=1 if continuous else 0, dense_adv=int(dense_adv))


def run_loss(lib, cfg, params, values, actions, old_logp, old_params, old_values, adv, targets, valids, index=None,
             offset=0, n=None, fused_heads=False):
    """Runs sf_moments + sf_ppo_loss + sf_loss_scalars; dataset arrays may be larger than the minibatch."""
    n = params.shape[0] if n is None else n
    A = params.shape[1]
    if fused_heads:  # params/values as strided columns of one [n, 1+A] matrix (how the model produces them)
        heads = torch.cat([dev(values, torch.float32)[:, None], dev(params, torch.float32)], dim=1).contiguous()
        p, v, ld = heads[:, 1:], heads[:, 0], 1 + A
        g = torch.zeros_like(heads)
        gp, gv = g[:, 1:], g[:, 0]
    else:
        p, v, ld = dev(params, torch.float32), dev(values, torch.float32), A
        gp, gv = torch.zeros_like(p), torch.zeros_like(v)
    d = dict(actions=dev(actions, torch.float32), old_logp=dev(old_logp, torch.float32),
             old_params=dev(old_params, torch.float32), old_values=dev(old_values, torch.float32),
             adv=dev(adv, torch.float32), targets=dev(targets, torch.float32), valids=dev(valids, torch.bool))
    idx = dev(index, torch.int32) if index is not None else None
    mom = torch.zeros(3, dtype=torch.float64, device="cuda")
    sums = torch.zeros(8, dtype=torch.float64, device="cuda")
    out = torch.zeros(16, dtype=torch.float32, device="cuda")
    if cfg.dense_adv:
        vd = d["valids"][idx.long()] if idx is not None else d["valids"][offset:offset + n]
        lib.moments(d["adv"], vd.contiguous(), None, n, mom)
    elif idx is not None:
        lib.moments(d["adv"], d["valids"], idx, n, mom)
    else:
        lib.moments(d["adv"][offset:offset + n], d["valids"][offset:offset + n], None, n, mom)
    lib.ppo_loss(p, ld, v, ld if fused_heads else 1, d["actions"], d["old_logp"], d["old_params"], d["old_values"],
                 d["adv"], d["targets"], d["valids"], idx, offset, n, A, cfg, mom, sums, gp, gv)
    lib.loss_scalars(sums, mom, cfg, out)
    torch.cuda.synchronize()
    o = out.cpu().numpy()
    res = {k: float(o[i]) for i, k in enumerate(oracle.SCALAR_NAMES)}
    res["grad_params"], res["grad_values"] = gp.cpu().numpy(), gv.cpu().numpy()
    return res


@pytest.mark.parametrize("case", ["ff_default", "ff_invalids", "ff_bootstrap_nonorm", "ff_continuous", "ff_vtrace",
                                  "ff_tuple", "ff_tuple_symkl", "ff_tuple_mixed"])
@pytest.mark.parametrize("fused_heads", [False, True])
def test_ppo_loss_golden(lib, golden, case, fused_heads):
    g = golden("learner_" + case)
    kv = _kv(g["argv"])
    n = int(g["mb_size"])
    continuous = case == "ff_continuous"
    cfg = _loss_cfg(lib, kv, continuous, dense_adv=case == "ff_vtrace")
    if "head_sizes" in g:  # Tuple space: n > 0 = Discrete(n) member, -D = Box(D) member
        cfg.num_heads = len(g["head_sizes"])
        for i, nh in enumerate(g["head_sizes"]):
            cfg.head_n[i] = int(nh)
    A = g["l_params"].shape[1]
    if case == "ff_vtrace":
        vs = torch.zeros(n, device="cuda")
        adv = torch.zeros(n, device="cuda")
        lib.vtrace(dev(g["l_params"]), A, dev(g["l_values"]), 1, dev(g["pb_actions"], torch.float32),
                   dev(g["pb_log_prob_actions"]), dev(g["pb_rewards"]), dev(g["pb_dones"], torch.bool), None, 0, n, A,
                   0, int(kv["recurrence"]), 0.99, float(kv["vtrace_rho"]), float(kv["vtrace_c"]), vs, adv)
        np.testing.assert_allclose(vs.cpu().numpy(), g["l_targets"], atol=1e-5)
        adv_arr, tgt_arr = adv.cpu().numpy(), vs.cpu().numpy()
    else:
        adv_arr, tgt_arr = g["pb_advantages"], g["pb_returns"]
    out = run_loss(lib, cfg, g["l_params"], g["l_values"], g["pb_actions"], g["pb_log_prob_actions"],
                   g["pb_action_logits"], g["pb_values"], adv_arr, tgt_arr, g["pb_valids"], n=n,
                   fused_heads=fused_heads)
    assert abs(out["adv_mean"] - float(g["l_adv_mean"])) < 1e-6
    assert abs(out["adv_std"] - float(g["l_adv_std"])) < 2e-6
    for k in ["policy_loss", "exploration_loss", "kl_loss", "value_loss"]:
        ref = float(g["l_" + k])
        assert abs(out[k] - ref) < 2e-6 + 1e-5 * abs(ref), (k, out[k], ref)
    np.testing.assert_allclose(out["grad_params"], g["l_grad_params"], atol=2e-7, rtol=2e-4)
    np.testing.assert_allclose(out["grad_values"], g["l_grad_values"], atol=2e-7, rtol=2e-4)


@pytest.mark.parametrize("A,kind,expl,klc", [(6, 0, 1, 0.0), (18, 0, 2, 0.3), (6, 1, 1, 0.1), (40, 0, 1, 0.2)])
def test_ppo_loss_vs_oracle_full_size(lib, A, kind, expl, klc):
    """config-2 minibatch size, read through a shuffled index into a 4x larger dataset."""
    rng = np.random.default_rng(A * 7 + kind)
    N, n = 131072, 32768
    nact = 1 if kind == 0 else A // 2
    old_params = rng.standard_normal((N, A)).astype(np.float32)
    actions = (rng.integers(0, A, (N, 1)) if kind == 0 else rng.standard_normal((N, nact))).astype(np.float32)
    old_logp = (-rng.random(N) * 2 - 0.3).astype(np.float32)
    old_values = rng.standard_normal(N).astype(np.float32)
    adv = rng.standard_normal(N).astype(np.float32) * 3 + 0.5
    targets = rng.standard_normal(N).astype(np.float32)
    valids = rng.random(N) > 0.07
    index = rng.permutation(N)[:n].astype(np.int32)
    params = (old_params[index] + 0.3 * rng.standard_normal((n, A))).astype(np.float32)
    values = (old_values[index] + rng.standard_normal(n) * 0.7).astype(np.float32)
    cfg = lib.sf_loss_cfg(clip_ratio=0.1, clip_value=0.5, value_loss_coeff=0.5, exploration_coeff=0.01, kl_coeff=klc,
                          exploration_kind=expl, action_kind=kind, dense_adv=0)
    out = run_loss(lib, cfg, params, values, actions, old_logp, old_params, old_values, adv, targets, valids,
                   index=index, n=n, fused_heads=True)
    ref = oracle.ppo_loss(params, values, actions[index], old_logp[index], old_params[index], old_values[index],
                          adv[index], targets[index], valids[index], action_kind=kind, clip_ratio=0.1, clip_value=0.5,
                          value_loss_coeff=0.5, exploration_coeff=0.01, exploration_kind=expl, kl_coeff=klc)
    assert out["n_valid"] == ref["n_valid"]
    for k in ["policy_loss", "exploration_loss", "kl_loss", "value_loss", "adv_mean", "adv_std", "kl_mean"]:
        assert abs(out[k] - ref[k]) < 1e-6 + 2e-5 * abs(ref[k]), (k, out[k], ref[k])
    assert abs(out["kl_max"] - ref["kl_max"]) < 1e-4 * max(1.0, abs(ref["kl_max"]))
    np.testing.assert_allclose(out["grad_params"], ref["grad_params"], atol=1e-9, rtol=5e-4)
    np.testing.assert_allclose(out["grad_values"], ref["grad_values"], atol=1e-9, rtol=5e-4)
    # size-independent property: gradients of invalid samples are exactly zero
    assert np.all(out["grad_params"][~valids[index]] == 0) and np.all(out["grad_values"][~valids[index]] == 0)
    # categorical: every gradient row sums to ~0 (softmax Jacobian annihilates constants)
    if kind == 0:
        assert np.abs(out["grad_params"].sum(1)).max() < 1e-8


def test_ppo_loss_offset_equals_index(lib):
    rng = np.random.default_rng(5)
    N, n, A = 4096, 1024, 6
    a = dict(old_params=rng.standard_normal((N, A)), actions=rng.integers(0, A, (N, 1)), old_logp=-rng.random(N) - 0.2,
             old_values=rng.standard_normal(N), adv=rng.standard_normal(N), targets=rng.standard_normal(N),
             valids=rng.random(N) > 0.1)
    params, values = rng.standard_normal((n, A)), rng.standard_normal(n)
    cfg = lib.sf_loss_cfg(clip_ratio=0.1, clip_value=1.0, value_loss_coeff=0.5, exploration_coeff=0.003, kl_coeff=0.0,
                          exploration_kind=1, action_kind=0, dense_adv=0)
    args = (params, values, a["actions"], a["old_logp"], a["old_params"], a["old_values"], a["adv"], a["targets"], a["valids"])
    o1 = run_loss(lib, cfg, *args, offset=2048, n=n)
    o2 = run_loss(lib, cfg, *args, index=np.arange(2048, 2048 + n), n=n)
    np.testing.assert_array_equal(o1["grad_params"], o2["grad_params"])  # integer indexing: bit-exact
    np.testing.assert_array_equal(o1["grad_values"], o2["grad_values"])


@pytest.mark.parametrize("N,rec", [(4096, 8), (640, 32), (96, 1)])
def test_vtrace_vs_oracle(lib, N, rec):
    rng = np.random.default_rng(N + rec)
    A = 6
    params = rng.standard_normal((N, A)).astype(np.float32)
    actions = rng.integers(0, A, (N, 1)).astype(np.float32)
    _, logp_a, _ = oracle.categorical(params, actions.reshape(-1))
    old_logp = (logp_a + rng.standard_normal(N) * 0.3).astype(np.float32)
    ratio = np.clip(np.exp(logp_a - old_logp), 0.05, 20.0).astype(np.float32)
    values, rewards = rng.standard_normal(N).astype(np.float32), rng.standard_normal(N).astype(np.float32)
    dones = rng.random(N) < 0.1
    vs, adv = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
    lib.vtrace(dev(params), A, dev(values), 1, dev(actions), dev(old_logp), dev(rewards), dev(dones, torch.bool), None,
               0, N, A, 0, rec, 0.99, 1.0, 1.0, vs, adv)
    rvs, radv = oracle.vtrace(ratio, values, rewards, dones.astype(np.float32), rec, 0.99)
    np.testing.assert_allclose(vs.cpu().numpy(), rvs, atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(adv.cpu().numpy(), radv, atol=2e-5, rtol=1e-5)


@pytest.mark.parametrize("P", [1687719 + 25, 1000, 7])
def test_clip_and_adam_vs_oracle(lib, P):
    rng = np.random.default_rng(P)
    Pp = (P + 3) // 4 * 4
    p = rng.standard_normal(Pp).astype(np.float32)
    m, v = np.zeros(Pp, np.float32), np.zeros(Pp, np.float32)
    tp, tm, tv = dev(p), dev(m), dev(v)
    sumsq = torch.zeros(1, dtype=torch.float64, device="cuda")
    for step in range(1, 4):
        g = (rng.standard_normal(Pp) * (10.0 if step == 2 else 0.001)).astype(np.float32)
        tg = dev(g)
        lib.grad_sumsq(tg, sumsq)
        lib.adam_step(tp, tg, tm, tv, step, 1e-4, 0.9, 0.999, 1e-6, 4.0, sumsq)
        gc, total = oracle.clip_grad_norm(g, 4.0)
        assert abs(float(sumsq.sqrt().item()) - total) < 1e-5 * max(1.0, total)
        p, m, v = oracle.adam_step(p, gc, m, v, step, 1e-4, 0.9, 0.999, 1e-6)
        np.testing.assert_allclose(tm.cpu().numpy(), m, rtol=2e-5, atol=1e-10)
        np.testing.assert_allclose(tv.cpu().numpy(), v, rtol=4e-5, atol=1e-14)
        np.testing.assert_allclose(tp.cpu().numpy(), p, rtol=2e-7, atol=2e-7)  # <= 1 ulp


@pytest.mark.parametrize("N,rec", [(131072, 1), (4096, 32), (96, 4), (64, 64)])
def test_minibatch_indices_properties(lib, N, rec):
    out = torch.empty(N, dtype=torch.int32, device="cuda")
    lib.minibatch_indices(out, N, rec, False, 0, 0)
    np.testing.assert_array_equal(out.cpu().numpy(), np.arange(N))          # default: contiguous slices
    lib.minibatch_indices(out, N, rec, True, 123, 0)
    a = out.cpu().numpy().astype(np.int64)
    np.testing.assert_array_equal(np.sort(a), np.arange(N))                 # a permutation of the dataset
    ch = a.reshape(-1, rec)
    assert np.all(ch[:, 0] % rec == 0)                                      # chunk starts are recurrence-aligned
    np.testing.assert_array_equal(ch, ch[:, :1] + np.arange(rec)[None, :])  # chunks stay intact (learner.py:512-515)
    if N // rec > 8:
        assert not np.array_equal(a, np.arange(N))
        lib.minibatch_indices(out, N, rec, True, 123, 1)
        assert not np.array_equal(out.cpu().numpy(), a)                     # new epoch -> new permutation
        lib.minibatch_indices(out, N, rec, True, 123, 0)
        np.testing.assert_array_equal(out.cpu().numpy(), a)                 # stateless / reproducible


def test_minibatch_expansion_golden(lib, golden):
    """index expansion of the reference (learner.py:512-519): our layout of a chunk == theirs"""
    g = golden("minibatch_indices")
    rec = int(g["recurrence"])
    starts = g["minibatches"].reshape(-1, rec)[:, 0]
    np.testing.assert_array_equal(g["minibatches"].reshape(-1, rec), starts[:, None] + np.arange(rec)[None])


def test_shuffled_minibatches_equal_the_reference_index_sets(lib, golden, tmp_path):
    """Learner._get_minibatches with shuffle_minibatches=True returns, element for element, the index sets the
    reference's Learner._get_minibatches returned under the same np.random seed (golden minibatch_indices.npz:
    learner.py:498-526 — seeded host permutation of chunk starts, expansion, np.split)."""
    from sample_factory_amd.algo.learning.learner import Learner, ParameterServer
    from sample_factory_amd.algo.utils.env_info import EnvInfo
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    g = golden("minibatch_indices")
    N, B, rec = int(g["experience_size"]), int(g["batch_size"]), int(g["recurrence"])
    cfg = default_cfg(use_rnn=False, recurrence=rec, nonlinearity="tanh", normalize_input=False, encoder_mlp_layers=[32, 32],
                      rollout=8, batch_size=B, num_batches_per_epoch=N // B, shuffle_minibatches=True, seed=0,
                      serial_mode=True, train_dir=str(tmp_path), experiment="t")
    env_info = EnvInfo(spaces.Dict({"obs": spaces.Box(-10, 10, (8,), np.float32)}), spaces.Discrete(3), 16)
    pv = torch.zeros(1, dtype=torch.int32)
    learner = Learner(cfg, env_info, pv, 0, ParameterServer(0, pv))
    learner.init()
    np.random.seed(123)
    mbs = learner._get_minibatches(B, N)
    assert len(mbs) == len(g["minibatches"])
    for (idx, offset, n), want in zip(mbs, g["minibatches"]):
        assert offset == 0 and n == B and idx.dtype == torch.int32
        np.testing.assert_array_equal(idx.cpu().numpy(), want)
    # a second epoch continues the SAME host stream, as the reference does
    second = np.concatenate([m[0].cpu().numpy() for m in learner._get_minibatches(B, N)])
    np.random.seed(123)
    np.random.permutation(np.arange(0, N, rec))
    want2 = np.random.permutation(np.arange(0, N, rec))
    np.testing.assert_array_equal(second.reshape(-1, rec)[:, 0], want2)
    # cfg.device_shuffle: stateless on-device permutation instead (still a permutation of whole chunks)
    cfg.device_shuffle = True
    dev_idx = np.concatenate([m[0].cpu().numpy() for m in learner._get_minibatches(B, N)])
    np.testing.assert_array_equal(np.sort(dev_idx), np.arange(N))
    assert np.all(dev_idx.reshape(-1, rec)[:, 0] % rec == 0)


def test_synthetic_env_bit_exact(lib):
    from sample_factory_amd.envs.synthetic import SyntheticVecEnv
    env = SyntheticVecEnv(num_agents=96, obs_shape=(4, 84, 84), num_actions=6, seed=3, env0=1000)
    T = 3
    slab = torch.zeros((96, T + 1, 4, 84, 84), dtype=torch.uint8, device="cuda")
    env.reset_into(slab[:, 0])
    np.testing.assert_array_equal(slab[:, 0].cpu().numpy().reshape(96, -1), oracle.synth_obs(96, 1000, 28224, 3, 0))
    rng = np.random.default_rng(0)
    for t in range(T):
        a = rng.integers(0, 6, 96).astype(np.int32)
        rew, term, trunc = env.step_into(dev(a, torch.int32), slab[:, t + 1])
        r_ref, t_ref = oracle.synth_step(a, 1000, 6, 3, t)
        np.testing.assert_array_equal(rew.cpu().numpy(), r_ref)
        np.testing.assert_array_equal(term.cpu().numpy(), t_ref)
        np.testing.assert_array_equal(slab[:, t + 1].cpu().numpy().reshape(96, -1), oracle.synth_obs(96, 1000, 28224, 3, t + 1))
    assert not trunc.any()
    # gymnasium-style surface
    obs, info = env.reset()
    o2, rew, term, trunc, info = env.step(torch.zeros(96, dtype=torch.int64))
    assert o2["obs"].shape == (96, 4, 84, 84) and o2["obs"].dtype == torch.uint8 and rew.shape == (96,)


@pytest.mark.parametrize("B,A", [(4096, 6), (100, 18), (7, 2)])
def test_sample_write_step_vs_oracle(lib, B, A):
    rng = np.random.default_rng(B + A)
    T, t = 5, 2
    logits = (rng.standard_normal((B, A)) * 2).astype(np.float32)
    values = rng.standard_normal(B).astype(np.float32)
    heads = dev(np.concatenate([values[:, None], logits], 1))
    tr = dict(actions=torch.full((B, T, 1), -7.0, device="cuda"), logits=torch.full((B, T, A), -7.0, device="cuda"),
              logp=torch.full((B, T), -7.0, device="cuda"), values=torch.full((B, T + 1), -7.0, device="cuda"),
              ver=torch.full((B, T), -7.0, device="cuda"))
    env_a = torch.zeros(B, dtype=torch.int32, device="cuda")
    lib.sample_write_step(heads[:, 1:], 1 + A, heads[:, 0], 1 + A, B, A, T, t, 11, 77, 5, 42.0, False, tr["actions"],
                          tr["logits"], tr["logp"], tr["values"], tr["ver"], env_a)
    act_ref, lp_ref = oracle.sample_categorical(logits, 11, 77, row0=5)
    np.testing.assert_array_equal(tr["actions"][:, t, 0].cpu().numpy(), act_ref)       # integer action: exact
    np.testing.assert_array_equal(env_a.cpu().numpy(), act_ref.astype(np.int32))
    np.testing.assert_allclose(tr["logp"][:, t].cpu().numpy(), lp_ref, atol=2e-6)
    np.testing.assert_array_equal(tr["logits"][:, t].cpu().numpy(), logits)
    np.testing.assert_array_equal(tr["values"][:, t].cpu().numpy(), values)
    assert torch.all(tr["ver"][:, t] == 42.0)
    # untouched steps keep their sentinel
    assert torch.all(tr["actions"][:, t + 1] == -7.0) and torch.all(tr["values"][:, t + 1] == -7.0)
    # log-prob is the gather at the recorded integer action (action_distributions.py:145-148)
    lp_all, _, _ = oracle.categorical(logits, act_ref)
    np.testing.assert_allclose(tr["logp"][:, t].cpu().numpy(), lp_all[np.arange(B), act_ref.astype(int)], atol=2e-6)
    # deterministic = argmax (enjoy.py:177-182)
    lib.sample_write_step(heads[:, 1:], 1 + A, heads[:, 0], 1 + A, B, A, T, t, 11, 77, 5, 42.0, True, tr["actions"],
                          tr["logits"], tr["logp"], tr["values"], tr["ver"], env_a)
    np.testing.assert_array_equal(env_a.cpu().numpy(), logits.argmax(1).astype(np.int32))


def test_vtrace_tuple_heads_vs_oracle(lib):
    """V-trace with a Tuple of Discrete heads: the importance ratio is exp(sum_h log_softmax(z_h)[a_h] - old_logp)"""
    rng = np.random.default_rng(21)
    hs, rec, ntraj = [4, 3, 5], 8, 50
    A, H, n = sum(hs), len(hs), rec * ntraj
    params = (rng.standard_normal((n, A)) * 1.2).astype(np.float32)
    values = rng.standard_normal(n).astype(np.float32)
    actions = np.stack([rng.integers(0, nh, n) for nh in hs], 1).astype(np.float32)
    old_logp = (-rng.random(n) * 3 - 0.5).astype(np.float32)
    rewards = rng.standard_normal(n).astype(np.float32)
    dones = rng.random(n) < 0.1
    off = np.cumsum([0] + hs)
    lp = np.zeros(n, np.float32)
    for h in range(H):
        z = params[:, off[h]:off[h + 1]].astype(np.float64)
        ls = z - z.max(1, keepdims=True)
        ls = ls - np.log(np.exp(ls).sum(1, keepdims=True))
        lp += ls[np.arange(n), actions[:, h].astype(int)].astype(np.float32)
    ratio = np.clip(np.exp(lp - old_logp), 0.05, 20.0).astype(np.float32)
    vs_ref, adv_ref = oracle.vtrace(ratio, values, rewards, dones.astype(np.float32), rec, 0.99, 0.9, 0.8)
    vs, adv = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    lib.vtrace(dev(params), A, dev(values), 1, dev(actions), dev(old_logp), dev(rewards), dev(dones, torch.bool), None, 0,
               n, A, 0, rec, 0.99, 0.9, 0.8, vs, adv, head_sizes=hs)
    np.testing.assert_allclose(vs.cpu().numpy(), vs_ref, atol=3e-5, rtol=1e-5)
    np.testing.assert_allclose(adv.cpu().numpy(), adv_ref, atol=3e-5, rtol=1e-5)


def test_lamb_step_vs_oracle(lib):
    """Lamb (optimizers.py:14-189): per-tensor trust ratios over a flat buffer with a segment-id map, padding skipped,
    gradient clipping folded in; three consecutive steps vs the C oracle."""
    rng = np.random.default_rng(5)
    sizes = [4096, 64, 100000, 512, 3, 1, 777]
    seg = np.concatenate([np.full(s, i, np.uint8) for i, s in enumerate(sizes)] + [np.full(13, 255, np.uint8)])
    rng.shuffle(seg[:5000])      # ids need not be contiguous (the fused heads matrix interleaves two tensors)
    P = seg.size
    p = (rng.standard_normal(P) * np.where(seg == 2, 3.0, 0.05)).astype(np.float32)
    p[seg == 4] = 0.0            # a zero tensor: trust ratio 1
    m, v = np.zeros(P, np.float32), np.zeros(P, np.float32)
    pd, md, vd = dev(p), dev(m), dev(v)
    segd = dev(seg, torch.uint8)
    scratch, sums = torch.empty(P, device="cuda"), torch.zeros(128, dtype=torch.float64, device="cuda")
    sumsq = torch.zeros(1, dtype=torch.float64, device="cuda")
    for step in (1, 2, 3):
        g = (rng.standard_normal(P) * 0.3).astype(np.float32)
        gd = dev(g)
        lib.grad_sumsq(gd, sumsq)
        lib.lamb_step(pd, gd, md, vd, scratch, segd, sums, len(sizes), step, 3e-3, 0.9, 0.999, 1e-6, 1e-4, 0.01, 4.0, sumsq)
        gc, _ = oracle.clip_grad_norm(g, 4.0)
        p, m, v = oracle.lamb_step(p, gc, m, v, seg, len(sizes), step, lr=3e-3)
        # the clip coefficient is an f32 function of an f64 (device) / f32 (oracle) norm: ~1e-5 relative in g^2
        np.testing.assert_allclose(md.cpu().numpy()[seg < 255], m[seg < 255], rtol=3e-5, atol=1e-9)
        np.testing.assert_allclose(vd.cpu().numpy()[seg < 255], v[seg < 255], rtol=6e-5, atol=1e-12)
        np.testing.assert_allclose(pd.cpu().numpy(), p, rtol=3e-5, atol=3e-7)
    assert np.array_equal(pd.cpu().numpy()[seg == 255], p[seg == 255])   # padding untouched


def test_sample_masked_vs_oracle(lib, golden):
    """action masks: kernel == oracle (actions exact), log-prob == the reference's masked_log_softmax gather, the
    deterministic action is the argmax of the reference's masked probabilities; mask read through a row stride."""
    g = golden("action_dist")
    z, mask, probs, lps = g["mask_logits"], g["mask_mask"], g["mask_probs"], g["mask_log_probs"]
    B, A = z.shape
    T, t = 3, 1
    heads = dev(np.concatenate([np.zeros((B, 1), np.float32), z], 1))
    slab_mask = torch.zeros((B, T + 1, A), dtype=torch.uint8, device="cuda")
    slab_mask[:, t] = dev(mask, torch.uint8)
    zz = lambda *s: torch.full(s, -7.0, device="cuda")
    ta, tl, tp, tv, tver = zz(B, T, 1), zz(B, T, A), zz(B, T), zz(B, T + 1), zz(B, T)
    env_a = torch.zeros(B, dtype=torch.int32, device="cuda")
    mview = slab_mask[:, t]
    lib.sample_write_step_masked(heads[:, 1:], 1 + A, heads[:, 0], 1 + A, mview, mview.stride(0), B, A, T, t, 9, 4, 2,
                                 1.0, False, ta, tl, tp, tv, tver, env_a)
    a_ref, lp_ref = oracle.sample_masked(z, mask, 9, 4, row0=2)
    np.testing.assert_array_equal(env_a.cpu().numpy(), a_ref.astype(np.int32))
    np.testing.assert_allclose(tp[:, t].cpu().numpy(), lp_ref, atol=3e-6)
    ok = mask.sum(1) > 0
    ai = a_ref.astype(int)
    np.testing.assert_allclose(tp[:, t].cpu().numpy()[ok], lps[np.arange(B)[ok], ai[ok]], atol=3e-6)
    np.testing.assert_array_equal(tl[:, t].cpu().numpy(), z)             # RAW logits are recorded
    lib.sample_write_step_masked(heads[:, 1:], 1 + A, heads[:, 0], 1 + A, mview, mview.stride(0), B, A, T, t, 9, 4, 2,
                                 1.0, True, ta, tl, tp, tv, tver, env_a)
    np.testing.assert_array_equal(env_a.cpu().numpy()[ok], probs[ok].argmax(1).astype(np.int32))


def test_sample_tuple_vs_oracle(lib):
    """Tuple of Discrete heads (TupleActionDistribution): per-head inverse-CDF sampling, log-prob = sum over heads;
    head 0 draws from the same Philox stream as the single-head sampler."""
    rng = np.random.default_rng(17)
    B, hs, T, t = 3000, [6, 3, 4], 4, 2
    A, H = sum(hs), len(hs)
    logits = (rng.standard_normal((B, A)) * 1.5).astype(np.float32)
    values = rng.standard_normal(B).astype(np.float32)
    heads = dev(np.concatenate([values[:, None], logits, np.zeros((B, 2), np.float32)], 1))
    ld = heads.shape[1]
    z = lambda *s: torch.full(s, -7.0, device="cuda")
    ta, tl, tp, tv, tver = z(B, T, H), z(B, T, A), z(B, T), z(B, T + 1), z(B, T)
    env_a = torch.zeros((B, H), dtype=torch.int32, device="cuda")
    lib.sample_write_step_tuple(heads[:, 1:], ld, heads[:, 0], ld, B, hs, T, t, 11, 77, 5, 9.0, False, ta, tl, tp, tv,
                                tver, env_a)
    a_ref, lp_ref = oracle.sample_tuple(logits, hs, 11, 77, row0=5)
    np.testing.assert_array_equal(ta[:, t].cpu().numpy(), a_ref)                         # integer actions: exact
    np.testing.assert_array_equal(env_a.cpu().numpy(), a_ref.astype(np.int32))
    np.testing.assert_allclose(tp[:, t].cpu().numpy(), lp_ref, atol=4e-6)
    np.testing.assert_array_equal(tl[:, t].cpu().numpy(), logits)
    np.testing.assert_array_equal(tv[:, t].cpu().numpy(), values)
    assert torch.all(tver[:, t] == 9.0) and torch.all(ta[:, t + 1] == -7.0)
    a0, _ = oracle.sample_categorical(logits[:, :hs[0]].copy(), 11, 77, row0=5)
    np.testing.assert_array_equal(a_ref[:, 0], a0)
    lib.sample_write_step_tuple(heads[:, 1:], ld, heads[:, 0], ld, B, hs, T, t, 11, 77, 5, 9.0, True, ta, tl, tp, tv,
                                tver, env_a)
    off = np.cumsum([0] + hs)
    am = np.stack([logits[:, off[i]:off[i + 1]].argmax(1) for i in range(H)], 1)
    np.testing.assert_array_equal(env_a.cpu().numpy(), am.astype(np.int32))



def test_sample_mixed_tuple_vs_oracle(lib):
    """Tuple with a Box member — hs = [6, -2, 4] = (Discrete(6), Box(2), Discrete(4)): the Discrete members as above, the Box
    member a = mu + clamp(exp(log_std)) * eps with eps from Box-Muller on Philox counter (step, dim / 2, 3, member index);
    actions [B, 1 + 2 + 1] f32, no int32 env_actions; log-prob = sum over the members; deterministic: arg-max / the mean"""
    rng = np.random.default_rng(23)
    B, hs, T, t = 3000, [6, -2, 4], 4, 1
    A, NA = 6 + 4 + 4, 4
    logits = (rng.standard_normal((B, A)) * 1.2).astype(np.float32)
    values = rng.standard_normal(B).astype(np.float32)
    heads = dev(np.concatenate([values[:, None], logits, np.zeros((B, 1), np.float32)], 1))
    ld = heads.shape[1]
    z = lambda *s: torch.full(s, -7.0, device="cuda")
    ta, tl, tp, tv, tver = z(B, T, NA), z(B, T, A), z(B, T), z(B, T + 1), z(B, T)
    lib.sample_write_step_tuple(heads[:, 1:], ld, heads[:, 0], ld, B, hs, T, t, 11, 77, 5, 9.0, False, ta, tl, tp, tv,
                                tver, None)
    a_ref, lp_ref = oracle.sample_tuple(logits, hs, 11, 77, row0=5)
    got = ta[:, t].cpu().numpy()
    np.testing.assert_array_equal(got[:, [0, 3]], a_ref[:, [0, 3]])                      # integer members: exact
    np.testing.assert_allclose(got[:, 1:3], a_ref[:, 1:3], atol=2e-5, rtol=1e-5)         # libm vs device exp / log / cos
    np.testing.assert_allclose(tp[:, t].cpu().numpy(), lp_ref, atol=3e-5)
    np.testing.assert_array_equal(tl[:, t].cpu().numpy(), logits)
    assert torch.all(ta[:, t + 1] == -7.0) and torch.all(tver[:, t] == 9.0)
    # the Box member is a standard normal in (a - mu) / sd
    sd = np.clip(np.exp(logits[:, 8:10]), 1e-4, 1e4)
    zs = (got[:, 1:3] - logits[:, 6:8]) / sd
    assert abs(zs.mean()) < 0.05 and abs(zs.std() - 1.0) < 0.05
    # a tuple with a Box member has no int32 action buffer
    with pytest.raises(lib.SfHipError):
        lib.sample_write_step_tuple(heads[:, 1:], ld, heads[:, 0], ld, B, hs, T, t, 11, 77, 5, 9.0, False, ta, tl, tp, tv,
                                    tver, torch.zeros((B, 3), dtype=torch.int32, device="cuda"))
    lib.sample_write_step_tuple(heads[:, 1:], ld, heads[:, 0], ld, B, hs, T, t, 11, 77, 5, 9.0, True, ta, tl, tp, tv,
                                tver, None)
    got = ta[:, t].cpu().numpy()
    np.testing.assert_array_equal(got[:, 0], logits[:, :6].argmax(1))
    np.testing.assert_array_equal(got[:, 1:3], logits[:, 6:8])
    np.testing.assert_array_equal(got[:, 3], logits[:, 10:].argmax(1))


def test_vtrace_mixed_tuple_vs_oracle(lib):
    """V-trace importance ratios of a Tuple(Discrete(3), Box(2), Discrete(4)) policy: exp(sum of the members' log-probs -
    old log-prob), against the oracle's loss head evaluated on the same parameters (its ratio is the same expression)"""
    rng = np.random.default_rng(29)
    hs, rec, ntraj = [3, -2, 4], 8, 64
    n, A, NA = rec * ntraj, 3 + 4 + 4, 4
    params = (rng.standard_normal((n, A)) * 0.7).astype(np.float32)
    values = rng.standard_normal(n).astype(np.float32)
    actions = np.concatenate([rng.integers(0, 3, (n, 1)), rng.standard_normal((n, 2)), rng.integers(0, 4, (n, 1))], 1).astype(np.float32)
    # reference log-prob in float64 from the definition
    def logp(p):
        lp = np.log(np.exp(p[:, :3] - p[:, :3].max(1, keepdims=True)) / np.exp(p[:, :3] - p[:, :3].max(1, keepdims=True)).sum(1, keepdims=True))
        out = lp[np.arange(n), actions[:, 0].astype(int)].astype(np.float64)
        mu, sd = p[:, 3:5].astype(np.float64), np.clip(np.exp(p[:, 5:7].astype(np.float64)), 1e-4, 1e4)
        out += (-((actions[:, 1:3] - mu) ** 2) / (2 * sd * sd) - np.log(sd) - 0.5 * np.log(2 * np.pi)).sum(1)
        l2 = p[:, 7:] - p[:, 7:].max(1, keepdims=True)
        l2 = l2 - np.log(np.exp(l2).sum(1, keepdims=True))
        return out + l2[np.arange(n), actions[:, 3].astype(int)]
    old_logp = (logp(params) + rng.standard_normal(n) * 0.2).astype(np.float32)
    rewards = rng.standard_normal(n).astype(np.float32)
    dones = rng.random(n) < 0.1
    ratio = np.clip(np.exp(logp(params) - old_logp), 0.05, 20.0).astype(np.float32)
    vs_ref, adv_ref = oracle.vtrace(ratio, values, rewards, dones.astype(np.float32), rec, 0.99, 0.9, 0.8)
    vs, adv = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    lib.vtrace(dev(params), A, dev(values), 1, dev(actions), dev(old_logp), dev(rewards), dev(dones, torch.bool), None, 0, n, A,
               0, rec, 0.99, 0.9, 0.8, vs, adv, head_sizes=hs)
    np.testing.assert_allclose(vs.cpu().numpy(), vs_ref, atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(adv.cpu().numpy(), adv_ref, atol=2e-5, rtol=1e-5)

def test_sample_continuous_vs_oracle(lib):
    rng = np.random.default_rng(3)
    B, D, T, t = 1000, 8, 4, 1
    A = 2 * D
    params = (rng.standard_normal((B, A)) * 0.7).astype(np.float32)
    values = rng.standard_normal(B).astype(np.float32)
    heads = dev(np.concatenate([values[:, None], params, np.zeros((B, 3), np.float32)], 1))   # ld = 20
    ld = heads.shape[1]
    z = lambda *s: torch.zeros(s, device="cuda")
    ta, tl, tp, tv, tver = z(B, T, D), z(B, T, A), z(B, T), z(B, T + 1), z(B, T)
    lib.sample_write_step(heads[:, 1:], ld, heads[:, 0], ld, B, A, T, t, 5, 9, 100, 3.0, False, ta, tl, tp, tv, tver,
                          None, action_kind=1)
    a_ref, lp_ref = oracle.sample_normal(params, 5, 9, row0=100)
    np.testing.assert_allclose(ta[:, t].cpu().numpy(), a_ref, atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(tp[:, t].cpu().numpy(), lp_ref, atol=2e-4, rtol=1e-5)
    np.testing.assert_array_equal(tl[:, t].cpu().numpy(), params)
    np.testing.assert_array_equal(tv[:, t].cpu().numpy(), values)
    lib.sample_write_step(heads[:, 1:], ld, heads[:, 0], ld, B, A, T, t, 5, 9, 100, 3.0, True, ta, tl, tp, tv, tver,
                          None, action_kind=1)
    np.testing.assert_array_equal(ta[:, t].cpu().numpy(), params[:, :D])      # deterministic: the mean


def test_sampler_distribution(lib):
    """sampling parity is distributional (torch.multinomial's stream cannot be reproduced): chi-square-ish check"""
    B, A, T = 200000, 6, 1
    logits = np.tile(np.array([[0.0, 1.0, 2.0, -1.0, 0.5, 0.2]], np.float32), (B, 1))
    heads = dev(np.concatenate([np.zeros((B, 1), np.float32), logits], 1))
    z = lambda *s: torch.zeros(s, device="cuda")
    env_a = torch.zeros(B, dtype=torch.int32, device="cuda")
    lib.sample_write_step(heads[:, 1:], 1 + A, heads[:, 0], 1 + A, B, A, T, 0, 1, 0, 0, 0.0, False, z(B, T, 1),
                          z(B, T, A), z(B, T), z(B, T + 1), z(B, T), env_a)
    freq = np.bincount(env_a.cpu().numpy(), minlength=A) / B
    p = np.exp(logits[0]) / np.exp(logits[0]).sum()
    assert np.abs(freq - p).max() < 4 * np.sqrt(p.max() / B) + 1e-3


def test_traj_write_env_step(lib):
    B, T, t = 300, 4, 1
    rng = np.random.default_rng(1)
    rew = (rng.standard_normal(B) * 5).astype(np.float32)
    term, trunc = rng.random(B) < 0.3, rng.random(B) < 0.1
    tr = dict(r=torch.zeros((B, T), device="cuda"), d=torch.zeros((B, T), dtype=torch.bool, device="cuda"),
              to=torch.zeros((B, T), dtype=torch.bool, device="cuda"), pid=torch.full((B, T), -1, dtype=torch.int32, device="cuda"))
    ep_ret, ep_len = dev(np.ones(B, np.float32)), dev(np.full(B, 3, np.int32))
    ep_stats = torch.zeros(3, dtype=torch.float64, device="cuda")
    lib.traj_write_env_step(dev(rew), dev(term, torch.bool), dev(trunc, torch.bool), T, t, 0.5, 2.0, 0, tr["r"], tr["d"],
                            tr["to"], tr["pid"], ep_ret, ep_len, ep_stats)
    np.testing.assert_allclose(tr["r"][:, t].cpu().numpy(), np.clip(rew * np.float32(0.5), -2, 2), atol=0)
    done = term | trunc
    np.testing.assert_array_equal(tr["d"][:, t].cpu().numpy(), done)
    np.testing.assert_array_equal(tr["to"][:, t].cpu().numpy(), trunc)
    assert torch.all(tr["pid"][:, t] == 0) and torch.all(tr["pid"][:, 0] == -1)
    s = ep_stats.cpu().numpy()
    assert s[2] == done.sum() and s[1] == 4 * done.sum()
    np.testing.assert_allclose(s[0], (1.0 + rew[done]).sum(), rtol=1e-6)
    np.testing.assert_allclose(ep_ret.cpu().numpy(), np.where(done, 0, 1.0 + rew), atol=1e-6)


def test_no_cpu_fallback(lib):
    """the product path refuses CPU tensors instead of silently computing elsewhere"""
    with pytest.raises(lib.SfHipError):
        lib.grad_sumsq(torch.zeros(16), torch.zeros(1, dtype=torch.float64, device="cuda"))


def test_tanh_scale_fwd_bwd_vs_torch(lib):
    """continuous_tanh_scale (model/action_parameterization.py:62-66): y = tanh(x/s)*s on the mean columns of the
    heads matrix, in place; backward g *= 1 - (y/s)^2 — against torch autograd; untouched columns stay bit-identical."""
    g = torch.Generator().manual_seed(3)
    n, ld, col0, D, s = 1000, 8, 1, 3, 2.0
    x = (torch.randn((n, ld), generator=g) * 3).cuda()
    xr = x.clone().requires_grad_(True)
    yr = torch.tanh(xr[:, col0:col0 + D] / s) * s
    gy = torch.randn((n, ld), generator=g).cuda()
    yr.backward(gy[:, col0:col0 + D])
    y = x.clone()
    lib.tanh_scale_fwd(y, ld, n, col0, D, s)
    assert (y[:, col0:col0 + D] - yr.detach()).abs().max().item() < 1e-6
    assert torch.equal(y[:, :col0], x[:, :col0]) and torch.equal(y[:, col0 + D:], x[:, col0 + D:])
    gx = gy.clone()
    lib.tanh_scale_bwd(gx, y, ld, n, col0, D, s)
    assert (gx[:, col0:col0 + D] - xr.grad[:, col0:col0 + D]).abs().max().item() < 1e-5
    assert torch.equal(gx[:, :col0], gy[:, :col0]) and torch.equal(gx[:, col0 + D:], gy[:, col0 + D:])
    with pytest.raises(lib.SfHipError):
        lib.tanh_scale_fwd(y, ld, n, col0, ld, s)  # columns past the row


@pytest.mark.parametrize("Cn,R,H", [(512, 6, 512), (200, 4, 512), (2048, 3, 512), (16, 5, 512), (512, 5, 256),
                                    (200, 3, 256), (2048, 2, 256), (1024, 3, 512), (1000, 3, 256),
                                    (256, 4, 512), (384, 33, 512)])  # (two-groups-per-work-group forward: 4 / 6 pairs)
def test_fused_lstm_sequence_passes_vs_torch_fp64(lib, Cn, R, H):
    """sf_lstm_seq_fwd / sf_lstm_seq_bwd (ONE persistent launch per BPTT pass, W_hh slices resident in LDS, per-step
    write-through hand-offs between the work-groups of a row group) against a float64 torch LSTM loop with the same
    masking (state zeroed after done/invalid steps): all saved tensors and the gate-pre-activation gradient.  Widths
    with a compiled instantiation: 512 (BASELINE configs[4], the reference's default rnn_size) and 256."""
    assert lib.lstm_seq_supported(Cn, H) and not lib.lstm_seq_supported(Cn, 64) and not lib.lstm_seq_supported(Cn, 1024)
    g = torch.Generator().manual_seed(Cn + R)
    gx = torch.randn((R, Cn, 4 * H), generator=g) * 0.7
    whh = torch.randn((H, 4 * H), generator=g) / np.sqrt(H)
    bhh = torch.randn((4 * H,), generator=g) * 0.1
    keep = (torch.rand((R, Cn), generator=g) > 0.15).float()
    h0, c0 = torch.randn((Cn, H), generator=g) * 0.5, torch.randn((Cn, H), generator=g) * 0.5
    dout = torch.randn((R, Cn, H), generator=g)
    # ---- float64 reference with autograd
    gx64 = gx.double().requires_grad_(True)
    W, b = whh.double(), bhh.double()
    h, c = h0.double(), c0.double()
    ref = dict(gates=[], hout=[], cout=[], hprev=[h], cprev=[c])
    for t in range(R):
        pre = gx64[t] + (h @ W + b)
        i_, f_, g_, o_ = pre.split(H, dim=1)
        ig, fg, gg, og = torch.sigmoid(i_), torch.sigmoid(f_), torch.tanh(g_), torch.sigmoid(o_)
        c = fg * c + ig * gg
        h = og * torch.tanh(c)
        ref["gates"].append(torch.cat([ig, fg, gg, og], 1)); ref["hout"].append(h); ref["cout"].append(c)
        h, c = h * keep[t].double()[:, None], c * keep[t].double()[:, None]
        ref["hprev"].append(h); ref["cprev"].append(c)
    (torch.stack(ref["hout"]) * dout.double()).sum().backward()
    # ---- fused kernels
    d = lambda x: x.cuda().contiguous()
    gates, hout, cout = (torch.full(s_, 7.0, device="cuda") for s_ in [(R, Cn, 4 * H), (R, Cn, H), (R, Cn, H)])
    hprev, cprev = torch.full((R + 1, Cn, H), 7.0, device="cuda"), torch.full((R + 1, Cn, H), 7.0, device="cuda")
    hprev[0], cprev[0] = d(h0), d(c0)
    sync = torch.zeros(192, dtype=torch.int32, device="cuda")
    lib.lstm_seq_fwd(d(gx), d(whh), d(bhh), d(keep), gates, hprev, hout, cprev, cout, sync, R, Cn, H)
    torch.cuda.synchronize()
    assert int(sync[128]) == 0, "forward pass aborted"
    tol = dict(atol=3e-6, rtol=2e-5)
    for name, got in [("gates", gates), ("hout", hout), ("cout", cout), ("hprev", hprev), ("cprev", cprev)]:
        want = torch.stack([x.detach() for x in ref[name]])
        np.testing.assert_allclose(got.cpu().double().numpy(), want.numpy(), err_msg=name, **tol)
    dgx = torch.full((R, Cn, 4 * H), 7.0, device="cuda")
    lib.lstm_seq_bwd(d(dout), gates, cprev, cout, d(keep), d(whh), dgx, sync, R, Cn, H)
    torch.cuda.synchronize()
    assert int(sync[128]) == 0, "backward pass aborted"
    want = gx64.grad
    scale = float(want.abs().max())
    np.testing.assert_allclose(dgx.cpu().double().numpy(), want.numpy(), atol=2e-6 * scale, rtol=5e-5)
    # env_major: hout / dout in the minibatch's own row order [Cn][R][H] — same numbers, no transpose copies around the pass
    hout_em = torch.full((Cn, R, H), 7.0, device="cuda")
    hprev2, cprev2 = hprev.clone(), cprev.clone()
    lib.lstm_seq_fwd(d(gx), d(whh), d(bhh), d(keep), gates, hprev2, hout_em, cprev2, cout, sync, R, Cn, H, env_major=True)
    assert torch.equal(hout_em, hout.transpose(0, 1).contiguous()) and torch.equal(hprev2, hprev)
    dgx2 = torch.full_like(dgx, 7.0)
    lib.lstm_seq_bwd(d(dout).transpose(0, 1).contiguous(), gates, cprev, cout, d(keep), d(whh), dgx2, sync, R, Cn, H, env_major=True)
    assert torch.equal(dgx2, dgx)


@pytest.mark.parametrize("Cn,R,H", [(512, 6, 512), (200, 4, 512), (2048, 3, 512), (16, 5, 512), (512, 5, 256),
                                    (200, 3, 256), (2048, 2, 256), (1024, 3, 512), (1000, 3, 256)])
def test_fused_gru_sequence_passes_vs_torch_fp64(lib, Cn, R, H):
    """sf_gru_seq_fwd / sf_gru_seq_bwd (the reference's default core: GRU-512) against a float64 torch GRU loop with the
    same masking; torch.nn.GRUCell's equations (r, z, n gate order, n = tanh(x_n + r * (h W_hn + b_hn))) are checked
    against the loop first, so the reference here IS torch's GRU.  Widths 512 and 256."""
    g = torch.Generator().manual_seed(3 * Cn + R)
    gx = torch.randn((R, Cn, 3 * H), generator=g) * 0.7
    whh = torch.randn((H, 3 * H), generator=g) / np.sqrt(H)
    bhh = torch.randn((3 * H,), generator=g) * 0.1
    keep = (torch.rand((R, Cn), generator=g) > 0.15).float()
    h0 = torch.randn((Cn, H), generator=g) * 0.5
    dout = torch.randn((R, Cn, H), generator=g)
    gx64 = gx.double().requires_grad_(True)
    W, b = whh.double(), bhh.double()
    h = h0.double()
    ref = dict(gates=[], hout=[], hprev=[h])
    gh_probe = torch.zeros((R, Cn, 3 * H), dtype=torch.float64, requires_grad=True)  # its gradient = dL/d(h W_hh + b_hh)
    for t in range(R):
        gh = h @ W + b + gh_probe[t]
        xr, xz, xn = gx64[t].split(H, dim=1)
        hr, hz, hn = gh.split(H, dim=1)
        r, z = torch.sigmoid(xr + hr), torch.sigmoid(xz + hz)
        n = torch.tanh(xn + r * hn)
        hnew = (1 - z) * n + z * h
        if t == 0:  # the loop is torch's GRUCell (weights: W_ih = I-free form -> compare through a cell with zero W_ih)
            cell = torch.nn.GRUCell(1, H).double()
            with torch.no_grad():
                cell.weight_ih.zero_(); cell.bias_ih.zero_(); cell.weight_hh.copy_(W.t()); cell.bias_hh.copy_(b)
                want0 = cell(torch.zeros((Cn, 1), dtype=torch.float64), h)
                r0, z0 = torch.sigmoid(hr), torch.sigmoid(hz)
                mine0 = (1 - z0) * torch.tanh(r0 * hn) + z0 * h
            assert (want0 - mine0).abs().max().item() < 1e-12
        ref["gates"].append(torch.cat([r, z, n, hn], 1)); ref["hout"].append(hnew)
        h = hnew * keep[t].double()[:, None]
        ref["hprev"].append(h)
    (torch.stack(ref["hout"]) * dout.double()).sum().backward()
    d = lambda x: x.cuda().contiguous()
    gates, hout = torch.full((R, Cn, 4 * H), 7.0, device="cuda"), torch.full((R, Cn, H), 7.0, device="cuda")
    hprev = torch.full((R + 1, Cn, H), 7.0, device="cuda")
    hprev[0] = d(h0)
    sync = torch.zeros(192, dtype=torch.int32, device="cuda")
    lib.gru_seq_fwd(d(gx), d(whh), d(bhh), d(keep), gates, hprev, hout, sync, R, Cn, H)
    torch.cuda.synchronize()
    assert int(sync[128]) == 0, "forward pass aborted"
    tol = dict(atol=3e-6, rtol=2e-5)
    for name, got in [("gates", gates), ("hout", hout), ("hprev", hprev)]:
        want = torch.stack([x.detach() for x in ref[name]])
        np.testing.assert_allclose(got.cpu().double().numpy(), want.numpy(), err_msg=name, **tol)
    dgx, dgh = torch.full((R, Cn, 3 * H), 7.0, device="cuda"), torch.full((R, Cn, 3 * H), 7.0, device="cuda")
    lib.gru_seq_bwd(d(dout), gates, hprev, d(keep), d(whh), dgx, dgh, sync, R, Cn, H)
    torch.cuda.synchronize()
    assert int(sync[128]) == 0, "backward pass aborted"
    for name, got, want in [("dgx", dgx, gx64.grad), ("dgh", dgh, gh_probe.grad)]:
        scale = float(want.abs().max())
        np.testing.assert_allclose(got.cpu().double().numpy(), want.numpy(), atol=2e-6 * scale, rtol=5e-5, err_msg=name)
    hout_em = torch.full((Cn, R, H), 7.0, device="cuda")
    hprev2 = hprev.clone()
    lib.gru_seq_fwd(d(gx), d(whh), d(bhh), d(keep), gates, hprev2, hout_em, sync, R, Cn, H, env_major=True)
    assert torch.equal(hout_em, hout.transpose(0, 1).contiguous()) and torch.equal(hprev2, hprev)
    dgx2, dgh2 = torch.full_like(dgx, 7.0), torch.full_like(dgh, 7.0)
    lib.gru_seq_bwd(d(dout).transpose(0, 1).contiguous(), gates, hprev, d(keep), d(whh), dgx2, dgh2, sync, R, Cn, H, env_major=True)
    assert torch.equal(dgx2, dgx) and torch.equal(dgh2, dgh)


def test_synthetic_continuous_env_kernel_rules(lib):
    """sf_synth_vec_step: reward / termination / dynamics rules of the Ant-shaped stand-in env, zero-copy into a strided
    slab slot, reproducible from (seed, global env id, step)"""
    from sample_factory_amd.envs.synthetic import SyntheticContinuousEnv
    N, D, A, T = 300, 27, 8, 3
    env = SyntheticContinuousEnv(num_agents=N, seed=5, env0=1000)
    slab = torch.zeros((N, T + 1, D), device="cuda")
    env.reset_into(slab[:, 0])
    o0 = slab[:, 0].clone()
    assert torch.equal(env.state, o0) and abs(float(o0.mean())) < 0.05 and 0.9 < float(o0.std()) < 1.1
    acts = torch.randn((N, T, A), device="cuda")
    for t in range(T):
        prev = slab[:, t].clone()
        rew, term, trunc = env.step_into(acts[:, t], slab[:, t + 1])
        want = -(acts[:, t] ** 2).mean(1) + 0.1 * prev[:, 0]
        np.testing.assert_allclose(rew.cpu().numpy(), want.cpu().numpy(), atol=1e-6, rtol=1e-6)
        nxt = slab[:, t + 1]
        noise = (nxt - 0.9 * prev) / 0.1                     # non-terminated envs: obs' = 0.9 obs + 0.1 noise
        keep = ~term
        assert float(noise[keep].std()) > 0.8 and float(noise[keep].std()) < 1.2 and not trunc.any()
        assert torch.equal(env.state, nxt)
    # same (seed, env ids) -> same stream; a shard of envs reproduces its slice of the full env set
    env2 = SyntheticContinuousEnv(num_agents=100, seed=5, env0=1100)
    s2 = torch.zeros((100, 2, D), device="cuda")
    env2.reset_into(s2[:, 0])
    assert torch.equal(s2[:, 0], o0[100:200])
    r2, t2, _ = env2.step_into(acts[100:200, 0], s2[:, 1])
    assert torch.equal(s2[:, 1], slab[100:200, 1])


def test_rnn_store_state_masks_and_packs(lib):
    B, H = 77, 24
    g = torch.Generator().manual_seed(2)
    h, c = torch.randn((B, H), generator=g).cuda(), torch.randn((B, H), generator=g).cuda()
    dones = (torch.rand((B, 5), generator=g) < 0.3).cuda()
    slab = torch.full((B, 4, 2 * H), 9.0, device="cuda")
    lib.rnn_store_state(h, c, dones[:, 2], slab[:, 3])
    want = torch.cat([h, c], 1) * (~dones[:, 2]).float()[:, None]
    assert torch.equal(slab[:, 3], want) and (slab[:, :3] == 9.0).all()
    lib.rnn_store_state(h, None, dones[:, 0], slab[:, 1, :H])
    assert torch.equal(slab[:, 1, :H], h * (~dones[:, 0]).float()[:, None]) and (slab[:, 1, H:] == 9.0).all()


@pytest.mark.parametrize("n,D,H1,H2,act,norm", [(2048, 27, 64, 64, 2, True), (77, 8, 32, 32, 1, False), (300, 27, 64, 64, 3, True)])
def test_fused_mlp_encoder_inference_kernel(lib, n, D, H1, H2, act, norm):
    """sf_mlp2_fwd (normalisation + two encoder layers in one launch, rollout on vector observations) against the layer
    kernels it replaces (sf_obsnorm_apply-style normalisation in torch + sf_linear_fwd twice) and float64"""
    g = torch.Generator().manual_seed(n)
    slab = torch.randn((n, 3, D), generator=g).cuda() * 3.0          # strided rows as in slab obs[:, t]
    x = slab[:, 1]
    w1, b1 = (torch.randn((D, H1), generator=g) / np.sqrt(D)).cuda(), (torch.randn(H1, generator=g) * 0.1).cuda()
    w2, b2 = (torch.randn((H1, H2), generator=g) / np.sqrt(H1)).cuda(), (torch.randn(H2, generator=g) * 0.1).cuda()
    mu = (torch.randn(D, generator=g) * 0.5).cuda() if norm else None
    rstd = (torch.rand(D, generator=g) + 0.5).cuda() if norm else None
    out = torch.full((n, H2), 7.0, device="cuda")
    assert lib.mlp2_supported(D, H1, H2) and not lib.mlp2_supported(64, 512, 512)
    lib.mlp2_fwd(x, x.stride(0), n, D, 0.0, 1.0, mu, rstd, w1, b1, w2, b2, act, out)
    xn = x.contiguous()
    if norm:
        xn = ((xn - mu) * rstd).clamp(-5.0, 5.0)
    from sample_factory_amd.model.actor_critic import _linear_desc
    h1, h2 = torch.empty((n, H1), device="cuda"), torch.empty((n, H2), device="cuda")
    lib.conv_fwd_raw(xn.contiguous(), D, None, 0, w1, b1, h1, n, _linear_desc(D, H1, act))   # the layer kernels (any act)
    lib.conv_fwd_raw(h1, H1, None, 0, w2, b2, h2, n, _linear_desc(H1, H2, act))
    fn = {1: torch.relu, 2: torch.tanh, 3: torch.nn.functional.elu}[act]
    ref = fn(fn(xn.double() @ w1.double() + b1.double()) @ w2.double() + b2.double())
    np.testing.assert_allclose(out.cpu().numpy(), ref.cpu().numpy(), atol=3e-6, rtol=2e-5)
    np.testing.assert_allclose(out.cpu().numpy(), h2.cpu().numpy(), atol=1e-6, rtol=1e-6)


def test_optimizer_skip_flag_and_sticky_abort_word(lib):
    """The abort word of the fused recurrent passes is STICKY (a launch clears its hand-off counters, never word 128),
    and sf_adam_step / sf_lamb_step leave weights and moments untouched while the word they are handed is non-zero —
    an aborted pass in minibatch k can neither be wiped by the launch of minibatch k+1 nor reach the weights."""
    P = 4096
    g = torch.Generator().manual_seed(3)
    p, gr = torch.randn(P, generator=g).cuda(), torch.randn(P, generator=g).cuda()
    m, v = torch.zeros(P, device="cuda"), torch.zeros(P, device="cuda")
    sumsq = torch.zeros(1, dtype=torch.float64, device="cuda")
    flag = torch.ones(1, dtype=torch.int32, device="cuda")
    p0 = p.clone()
    lib.grad_sumsq(gr, sumsq)
    lib.adam_step(p, gr, m, v, 1, 1e-3, 0.9, 0.999, 1e-6, 4.0, sumsq, skip_flag=flag)
    seg = torch.zeros(P, dtype=torch.uint8, device="cuda")
    lib.lamb_step(p, gr, m, v, torch.empty_like(p), seg, torch.zeros(128, dtype=torch.float64, device="cuda"), 1, 1, 1e-3,
                  0.9, 0.999, 1e-6, 1e-4, 0.01, 4.0, sumsq, skip_flag=flag)
    assert torch.equal(p, p0) and not m.any() and not v.any()
    flag.zero_()
    lib.adam_step(p, gr, m, v, 1, 1e-3, 0.9, 0.999, 1e-6, 4.0, sumsq, skip_flag=flag)
    assert not torch.equal(p, p0) and m.any() and v.any()
    # the sequence passes reset counters only
    H, Cn, R = 512, 64, 2
    z = lambda *s: torch.zeros(s, device="cuda")
    sync = torch.full((192,), 5, dtype=torch.int32, device="cuda")
    sync[128] = 0
    hprev, cprev = z(R + 1, Cn, H), z(R + 1, Cn, H)
    lib.lstm_seq_fwd(z(R, Cn, 4 * H), z(H, 4 * H), z(4 * H), torch.ones(R, Cn, device="cuda"), z(R, Cn, 4 * H), hprev,
                     z(R, Cn, H), cprev, z(R, Cn, H), sync, R, Cn, H)
    torch.cuda.synchronize()
    assert int(sync[128]) == 0 and int(sync[129]) == 5      # finished normally; words past the abort word untouched
    sync[128] = 1                                            # "an earlier pass aborted"
    lib.lstm_seq_fwd(z(R, Cn, 4 * H), z(H, 4 * H), z(4 * H), torch.ones(R, Cn, device="cuda"), z(R, Cn, 4 * H), hprev,
                     z(R, Cn, H), cprev, z(R, Cn, H), sync, R, Cn, H)
    torch.cuda.synchronize()
    assert int(sync[128]) == 1                               # still set after the next launch


def test_learner_raises_and_keeps_weights_after_an_aborted_recurrent_pass(lib):
    """Learner.train with the abort word set in the FIRST minibatch's forward pass (simulated by patching the clear):
    every optimiser step of the call is skipped, the epoch-end readback raises, the weights are the ones before."""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_synthetic_continuous_env
    from sample_factory_amd.train import make_runner
    register_env("synthetic_ant", make_synthetic_continuous_env)
    cfg = default_cfg(env="synthetic_ant", use_rnn=True, rnn_type="lstm", rnn_size=512, nonlinearity="tanh",
                      normalize_input=True, encoder_mlp_layers=[64, 64], rollout=8, recurrence=8, batch_size=512,
                      num_batches_per_epoch=2, num_epochs=2, num_workers=1, num_envs_per_worker=1, async_rl=False, seed=2,
                      serial_mode=True, synthetic_num_agents=128, with_vtrace=True, normalize_returns=False)
    cfg, runner = make_runner(cfg)
    runner.init()
    assert runner.iteration() is not None                      # a clean iteration first
    ac = runner.learner.actor_critic
    assert ac.rnn_abort_word() is not None and not ac.rnn_pass_aborted()
    p0, m0, step0 = ac.flat_params.clone(), runner.learner.exp_avg.clone(), runner.learner.train_step
    ac.rnn_abort_clear = lambda: ac._seq_sync_buf()[128:129].fill_(1)   # the next call starts out "aborted"
    with pytest.raises(lib.SfHipError, match="aborted"):
        runner.iteration()
    torch.cuda.synchronize()
    assert torch.equal(ac.flat_params, p0) and torch.equal(runner.learner.exp_avg, m0)
    assert runner.learner.train_step == step0 + 2               # one epoch of (skipped) steps was issued, then the raise


def test_slab_views_without_compaction_copies(lib):
    """The [E, T+1] arrays of the slab are consumed IN PLACE: sf_valid_mask emits the flat [E*T] mask next to the
    [E, T+1] one, sf_ppo_loss reads old values through `old_values_T`, sf_moments takes (index | offset) with dense or
    dataset-indexed values — each bit-equal to the path through compacted copies (integer indexing: exact)."""
    rng = np.random.default_rng(17)
    E, T, A, n = 96, 32, 6, 1024
    N = E * T
    pid = dev(np.where(rng.random((E, T)) < 0.1, 3, 0).astype(np.int32))
    pver = dev(rng.integers(-2000, 5, (E, T)).astype(np.float32))
    mk = lambda: (torch.zeros((E, T + 1), dtype=torch.bool, device="cuda"), dev(rng.integers(0, A, (E, T, 1)).astype(np.float32)),
                  dev(-rng.random((E, T)).astype(np.float32)), torch.zeros(1, dtype=torch.int32, device="cuda"))
    v1, a1, l1, c1 = mk()
    v2, a2, l2, c2 = v1.clone(), a1.clone(), l1.clone(), c1.clone()
    flat = torch.zeros(N, dtype=torch.bool, device="cuda")
    lib.valid_mask(pid, pver, v1, a1, 1, l1, 0, 7, 1000, c1)
    lib.valid_mask(pid, pver, v2, a2, 1, l2, 0, 7, 1000, c2, valids_flat=flat)
    assert torch.equal(v1, v2) and torch.equal(a1, a2) and torch.equal(l1, l2) and int(c1) == int(c2) > 0
    assert torch.equal(flat, v1[:, :T].reshape(N))
    with pytest.raises(lib.SfHipError):
        lib.valid_mask(pid, pver, v2, a2, 1, l2, 0, 7, 1000, c2, valids_flat=flat[:-1])
    # ---- loss: old values as the slab's [E, T+1] array vs its compacted copy
    slab_values = dev(rng.standard_normal((E, T + 1)).astype(np.float32))
    old_params = rng.standard_normal((N, A)).astype(np.float32)
    index = rng.permutation(N)[:n].astype(np.int32)
    params = (old_params[index] + 0.3 * rng.standard_normal((n, A))).astype(np.float32)
    values = rng.standard_normal(n).astype(np.float32)
    adv, tgt = rng.standard_normal(N).astype(np.float32), rng.standard_normal(N).astype(np.float32)
    old_logp = (-rng.random(N) - 0.3).astype(np.float32)
    actions = rng.integers(0, A, (N, 1)).astype(np.float32)

    def loss(old_values, ov_T, idx, off):
        cfg = lib.sf_loss_cfg(clip_ratio=0.1, clip_value=0.5, value_loss_coeff=0.5, exploration_coeff=0.01, kl_coeff=0.2,
                              exploration_kind=1, action_kind=0, dense_adv=0, old_values_T=ov_T)
        mom = torch.zeros(3, dtype=torch.float64, device="cuda")
        sums = torch.zeros(8, dtype=torch.float64, device="cuda")
        gp, gv = torch.zeros((n, A), device="cuda"), torch.zeros(n, device="cuda")
        lib.moments(dev(adv), flat, idx, n, mom, offset=off)
        lib.ppo_loss(dev(params), A, dev(values), 1, dev(actions), dev(old_logp), dev(old_params), old_values, dev(adv),
                     dev(tgt), flat, idx, off, n, A, cfg, mom, sums, gp, gv)
        return mom.clone(), sums.clone(), gp, gv
    compact = slab_values[:, :T].reshape(N).contiguous()
    for idx, off in ((dev(index), 0), (None, 2048)):
        m1, s1, gp1, gv1 = loss(compact, 0, idx, off)
        m2, s2, gp2, gv2 = loss(slab_values, T, idx, off)
        assert torch.equal(gp1, gp2) and torch.equal(gv1, gv2) and gv1.abs().max() > 0
        # the reductions are double-precision atomics: same terms, launch-dependent order
        assert torch.allclose(m1, m2, rtol=1e-13, atol=0) and torch.allclose(s1, s2, rtol=1e-12, atol=1e-300)
    # ---- moments: offset == slice views, dense values with a dataset-indexed mask == gathered mask
    mom_a, mom_b = torch.zeros(3, dtype=torch.float64, device="cuda"), torch.zeros(3, dtype=torch.float64, device="cuda")
    lib.moments(dev(adv), flat, None, n, mom_a, offset=2048)
    lib.moments(dev(adv)[2048:2048 + n], flat[2048:2048 + n], None, n, mom_b)
    assert torch.allclose(mom_a, mom_b, rtol=1e-13, atol=0) and mom_a[2] == mom_b[2]
    dense = dev(rng.standard_normal(n).astype(np.float32))
    lib.moments(dense, flat, dev(index), n, mom_a, dense_x=True)
    lib.moments(dense, flat[dev(index).long()].contiguous(), None, n, mom_b)
    assert torch.allclose(mom_a, mom_b, rtol=1e-13, atol=0) and mom_a[2] == mom_b[2] and float(mom_a[2]) < n


@pytest.mark.parametrize("shape,dtype", [((64, 9, 4, 84, 84), torch.uint8), ((300, 33), torch.float32),
                                         ((50, 5, 7), torch.uint8), ((128, 33, 1024), torch.float32),
                                         ((17, 3, 6), torch.bool)])
def test_copy_rows_equals_torch_copy(lib, shape, dtype):
    """sf_copy_rows (the slab's column copies) against torch's strided copy_: 16-byte, 4-byte and byte units"""
    g = torch.Generator().manual_seed(len(shape))
    mk = lambda: (torch.randint(0, 255, shape, generator=g).to(dtype) if dtype != torch.float32
                  else torch.randn(shape, generator=g)).cuda()
    a, b = mk(), mk()
    want = a.clone()
    want[:, 0].copy_(b[:, shape[1] - 1])
    lib.copy_rows(a[:, 0], b[:, shape[1] - 1])           # strided <- strided
    assert torch.equal(a, want)
    col = mk()[:, 1].contiguous()
    want[:, 2].copy_(col)
    lib.copy_rows(a[:, 2], col)                           # strided <- contiguous
    assert torch.equal(a, want)
    out = torch.empty_like(col)
    lib.copy_rows(out, a[:, 2])                           # contiguous <- strided
    assert torch.equal(out, col)
    with pytest.raises(lib.SfHipError):
        lib.copy_rows(a[:, 0], b[:, 0].float() if dtype != torch.float32 else b[:, 0].double())


@pytest.mark.parametrize("vtrace,use_index", [(False, True), (False, False), (True, True)])
def test_train_summaries_kernel_vs_torch(lib, vtrace, use_index):
    """sf_train_summaries (learner.py:843-923 in one pass) against the torch expressions it replaces"""
    rng = np.random.default_rng(33)
    E, T, A, n = 64, 16, 6, 512
    N = E * T
    valids = dev(rng.random(N) > 0.2, torch.bool)
    pid = dev(np.where(rng.random(N) < 0.15, 2, 0).astype(np.int32))
    pver = dev(rng.integers(0, 9, N).astype(np.float32))
    ratio = dev((rng.random(n) * 1.5 + 0.3).astype(np.float32))
    heads = dev(rng.standard_normal((n, 8)).astype(np.float32))
    values = heads[:, 0]
    slab_values = dev(rng.standard_normal((E, T + 1)).astype(np.float32))
    actions = dev(rng.integers(0, A, (N, 1)).astype(np.float32))
    adv = dev(rng.standard_normal(n if vtrace else N).astype(np.float32))
    logits = dev((rng.standard_normal((N, A)) * 3).astype(np.float32))
    m2 = dev(rng.random(5000).astype(np.float32))
    index = dev(rng.permutation(N)[:n].astype(np.int32)) if use_index else None
    off = 0 if use_index else 256
    out = torch.empty(24, dtype=torch.float64, device="cuda")
    lib.train_summaries(valids, ratio, values, 8, slab_values, T, actions, 1, adv, vtrace, pid, pver, logits, A, index, off,
                        n, 0, 11, 0.1, m2, out)
    o = out.cpu().numpy()
    rows = index.long() if use_index else torch.arange(off, off + n, device="cuda")
    v, same = valids[rows], pid[rows] == 0
    vr = ratio[v]
    old_v = slab_values[:, :T].reshape(-1)[rows]
    dv = (values - old_v).abs()
    vd = (11.0 - pver[rows])[same]
    ad = adv if vtrace else adv[rows]
    want = [n, v.sum(), same.sum(), values.double().sum(), (1 - vr).abs().double().sum(),
            ((vr < 1 / 1.1).sum() + (vr > 1.1).sum()), dv.double().sum(), vd.double().sum(), vr.min(), vr.max(), dv.max(),
            actions[rows].min(), actions[rows].max(), ad.min(), ad.max(), logits[rows].abs().max(), vd.min(), vd.max(),
            m2.max()]
    np.testing.assert_allclose(o[:19], [float(x) for x in want], rtol=1e-6, atol=1e-6)
    # no valid / no same-policy row: the minima / maxima stay at +-inf (the Learner substitutes the reference's defaults)
    lib.train_summaries(torch.zeros_like(valids), ratio, values, 8, slab_values, T, actions, 1, adv, vtrace, pid + 5, pver,
                        logits, A, index, off, n, 0, 11, 0.1, None, out)
    o = out.cpu().numpy()
    assert o[1] == 0 and o[2] == 0 and np.isposinf(o[8]) and np.isneginf(o[9]) and np.isposinf(o[16]) and np.isneginf(o[18])


@pytest.mark.parametrize("kind", ["lstm", "gru"])
@pytest.mark.parametrize("Cn,R,H", [(512, 6, 512), (200, 4, 512), (1024, 3, 512), (16, 5, 256), (1000, 2, 256), (320, 7, 512)])
def test_fused_sequence_forward_with_input_projection(lib, kind, Cn, R, H):
    """sf_lstm_seq_fwd_x / sf_gru_seq_fwd_x: the pass computes gx_t = x_t W_ih^T + b_ih itself (W_ih fragments of the
    work-group's gate columns in registers, the products of step t+1 behind step t's hand-off).  Every output equals the
    gx form's, fed with the float64 projection rounded to f32, to a few f32 ulps of the pre-activation."""
    G, Kx = (4 if kind == "lstm" else 3), 64
    assert lib.seq_fwd_x_supported(Cn, H, Kx) and not lib.seq_fwd_x_supported(Cn, H, 48) and not lib.seq_fwd_x_supported(2048, H, Kx)
    g = torch.Generator().manual_seed(Cn + R + G)
    x = torch.randn((R, Cn, Kx), generator=g)
    wih = torch.randn((G * H, Kx), generator=g) / np.sqrt(Kx)
    bih = torch.randn((G * H,), generator=g) * 0.1
    whh = torch.randn((H, G * H), generator=g) / np.sqrt(H)
    bhh = torch.randn((G * H,), generator=g) * 0.1
    keep = (torch.rand((R, Cn), generator=g) > 0.15).float()
    h0, c0 = torch.randn((Cn, H), generator=g) * 0.5, torch.randn((Cn, H), generator=g) * 0.5
    gx = (x.double() @ wih.double().t() + bih.double()).float()
    d = lambda t: t.cuda().contiguous()
    sync = torch.zeros(192, dtype=torch.int32, device="cuda")

    def run(fused):
        gates, hout, cout = (torch.full(s_, 7.0, device="cuda") for s_ in [(R, Cn, 4 * H), (R, Cn, H), (R, Cn, H)])
        hprev, cprev = torch.full((R + 1, Cn, H), 7.0, device="cuda"), torch.full((R + 1, Cn, H), 7.0, device="cuda")
        hprev[0], cprev[0] = d(h0), d(c0)
        if kind == "lstm":
            if fused:
                lib.lstm_seq_fwd_x(d(x), d(wih), d(bih), d(whh), d(bhh), d(keep), gates, hprev, hout, cprev, cout, sync, R, Cn, H)
            else:
                lib.lstm_seq_fwd(d(gx), d(whh), d(bhh), d(keep), gates, hprev, hout, cprev, cout, sync, R, Cn, H)
            outs = dict(gates=gates, hout=hout, cout=cout, hprev=hprev, cprev=cprev)
        else:
            if fused:
                lib.gru_seq_fwd_x(d(x), d(wih), d(bih), d(whh), d(bhh), d(keep), gates, hprev, hout, sync, R, Cn, H)
            else:
                lib.gru_seq_fwd(d(gx), d(whh), d(bhh), d(keep), gates, hprev, hout, sync, R, Cn, H)
            outs = dict(gates=gates, hout=hout, hprev=hprev)
        torch.cuda.synchronize()
        assert int(sync[128]) == 0, "pass aborted"
        return outs

    a, b = run(True), run(False)
    for k in b:
        np.testing.assert_allclose(a[k].cpu().numpy(), b[k].cpu().numpy(), atol=4e-6, rtol=2e-5, err_msg=k)

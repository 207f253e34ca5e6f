"""The CPU oracle (oracle/sf_oracle.c) against golden vectors produced by the reference itself
(oracle/gen_golden.py ran Sample Factory's own functions).  This is what pins the oracle."""
import os

import numpy as np
import pytest

import oracle


def test_philox_known_answers():
    # Random123 philox4x32-10 known-answer vectors
    assert [hex(x) for x in oracle.philox((0, 0, 0, 0), (0, 0))] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(x) for x in oracle.philox((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2)] == [
        "0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert [hex(x) for x in oracle.philox((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0))] == [
        "0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def test_gae_matches_reference(golden):
    g = golden("gae")
    for i in range(int(g["num_cases"])):
        adv = oracle.gae(g[f"c{i}_rewards"], g[f"c{i}_dones"], g[f"c{i}_values"], g[f"c{i}_valids"],
                         float(g[f"c{i}_gamma"]), float(g[f"c{i}_lambda"]))
        np.testing.assert_allclose(adv, g[f"c{i}_adv"], rtol=0, atol=1e-5)  # north-star bound is 1e-4


def test_rms_matches_reference(golden):
    g = golden("rms")
    stats = np.array([0.0, 1.0, 1.0])
    for i in range(int(g["num_steps"])):
        stats = oracle.rms_update(stats, g[f"s{i}_x"])
        np.testing.assert_allclose(stats, g[f"s{i}_stats"], rtol=2e-6)
        np.testing.assert_allclose(oracle.rms_apply(stats, g[f"s{i}_x"]), g[f"s{i}_normalized"], atol=2e-6, rtol=1e-6)
        np.testing.assert_allclose(oracle.rms_apply(stats, g[f"s{i}_z"], denormalize=True), g[f"s{i}_denormalized"],
                                   atol=1e-5, rtol=1e-6)
    np.testing.assert_allclose(oracle.rms_apply(g["eval_stats"], g["eval_x"]), g["eval_normalized"], atol=2e-6)


def test_categorical_matches_reference(golden):
    g = golden("action_dist")
    lp, _, ent = oracle.categorical(g["ka_logits"], np.zeros(1))
    np.testing.assert_allclose(np.exp(lp), [[0.09003057, 0.24472847, 0.66524096]], atol=1e-7)  # reference test KA
    np.testing.assert_allclose(ent, g["ka_entropy"], atol=1e-6)
    for i in range(int(g["num_cat"])):
        lp, la, ent = oracle.categorical(g[f"cat{i}_logits"], g[f"cat{i}_actions"].reshape(-1))
        np.testing.assert_allclose(lp, g[f"cat{i}_log_probs"], atol=2e-6, rtol=1e-6)
        np.testing.assert_allclose(la, g[f"cat{i}_log_prob_actions"], atol=2e-6, rtol=1e-6)
        np.testing.assert_allclose(ent, g[f"cat{i}_entropy"], atol=2e-6, rtol=1e-5)


LEARNER_CASES = ["ff_default", "ff_invalids", "ff_bootstrap_nonorm", "ff_continuous"]


def _cfg_from_argv(argv):
    kv = {}
    for tok in str(argv).split():
        if tok.startswith("--") and "=" in tok:
            k, v = tok[2:].split("=", 1)
            kv[k] = v
    return kv


@pytest.mark.parametrize("case", LEARNER_CASES)
def test_prepare_batch_matches_reference(golden, case):
    g = golden("learner_" + case)
    kv = _cfg_from_argv(g["argv"])
    values = g["in_values"].copy()
    values[:, -1] = g["bootstrap_values"]
    norm = kv.get("normalize_returns", "True") == "True"
    out = oracle.prepare_batch(
        g["in_rewards"], g["in_dones"], g["in_time_outs"], values, g["in_policy_id"], g["in_policy_version"],
        g["in_actions"], g["in_log_prob_actions"], my_policy_id=0, train_step=int(g["train_step"]),
        max_policy_lag=int(kv.get("max_policy_lag", 1000)), normalize_returns=norm,
        value_bootstrap=kv.get("value_bootstrap", "False") == "True", gamma=0.99, lam=0.95,
        rms=g["in_rms"] if norm else (0, 1, 1))
    assert out["num_invalids"] == int(g["pb_num_invalids"])                       # integer: exact
    np.testing.assert_array_equal(out["valids"], g["out_valids_full"])            # mask: exact
    np.testing.assert_array_equal(out["valids"][:, :-1].reshape(-1), g["pb_valids"])
    np.testing.assert_allclose(out["rewards"], g["out_rewards"], atol=1e-6)
    np.testing.assert_allclose(out["advantages"].reshape(-1), g["pb_advantages"], atol=1e-5)
    np.testing.assert_allclose(out["returns"].reshape(-1), g["pb_returns"], atol=1e-5)
    np.testing.assert_array_equal(out["actions"].reshape(g["pb_actions"].shape), g["pb_actions"])
    np.testing.assert_array_equal(out["log_prob_actions"].reshape(-1), g["pb_log_prob_actions"])
    if norm:
        np.testing.assert_allclose(out["rms"], g["out_rms"], rtol=1e-6)


def _loss_kwargs(kv, continuous):
    expl = kv.get("exploration_loss", "entropy")
    coeff = float(kv.get("exploration_loss_coeff", 0.003))
    kind = 0 if coeff == 0 else (1 if expl == "entropy" else 2)
    return dict(action_kind=1 if continuous else 0, clip_ratio=0.1, clip_value=1.0, value_loss_coeff=0.5,
                exploration_coeff=coeff, exploration_kind=kind, kl_coeff=float(kv.get("kl_loss_coeff", 0.0)))


@pytest.mark.parametrize("case", LEARNER_CASES + ["ff_vtrace", "ff_tuple", "ff_tuple_symkl", "ff_tuple_mixed"])
def test_ppo_loss_matches_reference(golden, case):
    g = golden("learner_" + case)
    kv = _cfg_from_argv(g["argv"])
    n = int(g["mb_size"])
    continuous = case == "ff_continuous"
    if case == "ff_vtrace":
        rec = int(kv["recurrence"])
        vs, adv = oracle.vtrace(g["l_ratio"], g["l_values"], g["pb_rewards"][:n], g["pb_dones"][:n].astype(np.float32),
                                rec, 0.99, float(kv["vtrace_rho"]), float(kv["vtrace_c"]))
        np.testing.assert_allclose(vs, g["l_targets"], atol=1e-5)
        targets = vs
    else:
        adv, targets = g["pb_advantages"][:n], g["pb_returns"][:n]
        np.testing.assert_array_equal(targets, g["l_targets"])
    out = oracle.ppo_loss(g["l_params"], g["l_values"], g["pb_actions"][:n], g["pb_log_prob_actions"][:n],
                          g["pb_action_logits"][:n], g["pb_values"][:n], adv, targets, g["pb_valids"][:n],
                          head_sizes=[int(x) for x in g["head_sizes"]] if "head_sizes" in g else None,
                          **_loss_kwargs(kv, continuous))
    assert abs(out["adv_mean"] - float(g["l_adv_mean"])) < 1e-6
    assert abs(out["adv_std"] - float(g["l_adv_std"])) < 2e-6
    for k in ["policy_loss", "exploration_loss", "kl_loss", "value_loss"]:
        assert abs(out[k] - float(g["l_" + k])) < 2e-6 + 1e-5 * abs(float(g["l_" + k])), (k, out[k], float(g["l_" + k]))
    np.testing.assert_allclose(out["grad_params"], g["l_grad_params"], atol=2e-7, rtol=2e-4)
    np.testing.assert_allclose(out["grad_values"], g["l_grad_values"], atol=2e-7, rtol=2e-4)


def test_minibatch_index_expansion(golden):
    g = golden("minibatch_indices")
    rec, bs = int(g["recurrence"]), int(g["batch_size"])
    starts = g["chunk_start_permutation"]
    full = (starts[:, None] + np.arange(rec)[None, :]).reshape(-1)
    np.testing.assert_array_equal(full.reshape(-1, bs), g["minibatches"])


def test_masked_categorical_matches_reference(golden):
    """obs["action_mask"]: the oracle's masked sampler draws only allowed actions, its log-prob is the reference's
    masked_log_softmax at the drawn action, and the empirical frequencies follow the reference's masked_softmax."""
    g = golden("action_dist")
    z, mask, probs, logp = g["mask_logits"], g["mask_mask"], g["mask_probs"], g["mask_log_probs"]
    N, A = z.shape
    counts = np.zeros((N, A))
    for step in range(400):
        a, lp = oracle.sample_masked(z, mask, seed=3, step=step)
        ai = a.astype(int)
        rows = np.arange(N)
        ok = mask.sum(1) > 0
        assert mask[rows[ok], ai[ok]].all()                                  # never a masked-out action
        np.testing.assert_allclose(lp[ok], logp[rows[ok], ai[ok]], atol=3e-6)
        counts[rows, ai] += 1
    freq = counts / 400.0
    ok = mask.sum(1) > 0
    assert np.abs(freq[ok] - probs[ok]).max() < 0.12                         # 400 draws: 4 sigma of p(1-p)/400
    assert np.abs(freq[~ok] - 1.0 / A).max() < 0.12                           # all-masked rows: uniform fallback


@pytest.mark.skipif(not os.path.isdir("/root/reference/sample_factory"), reason="the reference only exists in the build container")
def test_reference_loads_a_checkpoint_written_here():
    """tests/golden/ours_checkpoint_mlp.pth was written by THIS engine's Learner.save() on the GPU
    (tests/test_gpu_runner.py::test_write_a_checkpoint_for_the_reference).  The reference's Learner.init() resumes from
    it through its own load_from_checkpoint (strict nn.Module.load_state_dict + torch.optim.Adam.load_state_dict) and
    its model reproduces the outputs this engine recorded for the probe batch.  Subprocess: this process may already
    hold this repo's `sample_factory` alias package."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.exists(os.path.join(root, "tests", "golden", "ours_checkpoint_mlp.pth")):
        pytest.skip("fixture not generated yet")
    r = subprocess.run([sys.executable, "-m", "oracle.check_our_checkpoint"], cwd=root, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0 and "INTEROP OK" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


def test_action_distribution_objects_match_reference_known_answers(golden):
    """sample_factory_amd.algo.utils.action_distributions (the object API of action_distributions.py:99-323) against the
    vectors the reference's own classes produced: probabilities, log-probs, entropy, KL(new || old), symmetric KL with the
    uniform prior, arg-max, masked soft-max — categorical, diagonal normal and tuple-of-heads."""
    import torch
    from sample_factory_amd.algo.utils.action_distributions import (CategoricalActionDistribution, ContinuousActionDistribution,
                                                                     TupleActionDistribution, argmax_actions,
                                                                     get_action_distribution, sample_actions_log_probs)
    from sample_factory_amd.envs import spaces
    g = golden("action_dist")
    t = lambda k: torch.from_numpy(np.asarray(g[k]))
    close = lambda a, k, tol=2e-6: np.testing.assert_allclose(a.numpy(), g[k], atol=tol, rtol=2e-6, err_msg=k)
    d = CategoricalActionDistribution(t("ka_logits"))
    close(d.probs, "ka_probs"); close(d.entropy(), "ka_entropy")
    for i in range(int(g["num_cat"])):
        d, do = CategoricalActionDistribution(t(f"cat{i}_logits")), CategoricalActionDistribution(t(f"cat{i}_old_logits"))
        close(d.probs, f"cat{i}_probs"); close(d.log_probs, f"cat{i}_log_probs", 1e-5); close(d.entropy(), f"cat{i}_entropy")
        close(d.log_prob(t(f"cat{i}_actions")), f"cat{i}_log_prob_actions", 1e-5)
        close(d.kl_divergence(do), f"cat{i}_kl", 1e-5); close(d.symmetric_kl_with_uniform_prior(), f"cat{i}_symkl_uniform", 1e-5)
        assert np.array_equal(argmax_actions(d).reshape(-1).numpy(), g[f"cat{i}_argmax"])
        a, lp = sample_actions_log_probs(d)
        assert a.shape == (d.probs.shape[0], 1) and torch.equal(lp, d.log_prob(a))
    for i in range(int(g["num_con"])):
        d, do = ContinuousActionDistribution(t(f"con{i}_params")), ContinuousActionDistribution(t(f"con{i}_old_params"))
        close(d.log_prob(t(f"con{i}_actions")), f"con{i}_log_prob_actions", 1e-5)
        close(d.entropy(), f"con{i}_entropy", 1e-5); close(d.kl_divergence(do), f"con{i}_kl", 2e-5)
        assert torch.equal(argmax_actions(d), d.means)
    for i in range(int(g["num_tup"])):
        space = spaces.Tuple([spaces.Discrete(int(n)) for n in g[f"tup{i}_heads"]])
        d, do = get_action_distribution(space, t(f"tup{i}_logits")), TupleActionDistribution(space, t(f"tup{i}_old_logits"))
        assert isinstance(d, TupleActionDistribution)
        close(d.log_prob(t(f"tup{i}_actions")), f"tup{i}_log_prob_actions", 1e-5); close(d.entropy(), f"tup{i}_entropy", 1e-5)
        close(d.kl_divergence(do), f"tup{i}_kl", 1e-5); close(d.symmetric_kl_with_uniform_prior(), f"tup{i}_symkl_uniform", 1e-5)
        a, lp = sample_actions_log_probs(d)
        assert a.shape == (45, len(g[f"tup{i}_heads"])) and torch.allclose(lp, d.log_prob(a))
        assert d.argmax().shape == a.shape
    d = CategoricalActionDistribution(t("mask_logits"), t("mask_mask").float())
    close(d.probs, "mask_probs"); close(d.log_probs, "mask_log_probs", 1e-5)
    assert d.sample().shape == (300, 1)            # rows with every action masked out still draw
    assert isinstance(get_action_distribution(spaces.Box(-1, 1, (3,), np.float32), torch.zeros(5, 6)), ContinuousActionDistribution)


@pytest.mark.parametrize("name,ref_bound", [("train_cnn84", 1e-4), ("train_c5", 1e-4), ("train_c5_gru", 1e-4)])
def test_float64_anchor_of_the_train_replays(golden, name, ref_bound):
    """The replays of the two BASELINE configurations carry the reference's own training loop run in float64 (delta64_*;
    oracle/gen_golden.py full_train_fp64).  The reference's fp32 run sits within 1e-4 of it on every tensor (max|d32 - d64| /
    max|d64|): that is the yardstick the GPU test holds the HIP path to (tests/test_gpu_parity_c2_c5.py); for the ReLU
    network the golden also counts the positive outputs of every ReLU layer per SGD step (relu_pos64)."""
    g = golden(name)
    worst = 0.0
    for k in g["param_names"]:
        k = str(k)
        d32, d64 = g["delta_" + k].astype(np.float64), g["delta64_" + k]
        assert d64.dtype == np.float64 and d64.shape == d32.shape
        worst = max(worst, float(np.abs(d32 - d64).max() / np.abs(d64).max()))
    assert worst <= ref_bound, worst
    steps = int(g["num_batches"]) * int(g["num_epochs"])
    if name == "train_cnn84":
        rp = g["relu_pos64"]
        assert rp.shape == (steps, 4) and (rp > 0).all()
        # conv1 32 x 20 x 20, conv2 64 x 9 x 9, conv3 64 x 7 x 7, fc 512 outputs per sample, 1024 samples per minibatch
        assert (rp <= np.array([12800, 5184, 3136, 512]) * 1024).all()
    else:
        assert "relu_pos64" not in g.files  # tanh network: no kinks to count


@pytest.mark.parametrize("case", ["ff_sync", "gru_scale_clip", "lstm_async", "u8_image", "multikey_policy1", "tuple_heads",
                                  "box_actions", "tuple_mixed"])
def test_rollout_restatement_equals_the_reference_runner(golden, case):
    """oracle.rollout_replay (numpy) against slab rows written by the reference's BatchedVectorEnvRunner
    (tests/golden/rollout_*.npz, oracle/gen_golden.py gen_rollout_case): bit-equal"""
    import oracle
    g = golden("rollout_" + case)
    T = int(g["T"])
    key = [str(k) for k in g["obs_keys"]][0]
    outs, st = oracle.rollout_replay(g["in_rew"], g["in_term"], g["in_trunc"], g["in_new_rnn"], g[f"in_obs_{key}"], T=T,
                                     reward_scale=float(g["reward_scale"]), reward_clip=float(g["reward_clip"]))
    assert len(outs) == int(g["n_rollouts"])
    for r, cur in enumerate(outs):
        for name in ("rewards", "dones", "time_outs", "rnn_states"):
            np.testing.assert_array_equal(cur[name], g[f"out{r}_{name}"], err_msg=f"rollout {r} {name}")
        np.testing.assert_array_equal(cur["obs"], g[f"out{r}_obs_{key}"])
        assert (g[f"out{r}_policy_id"] == int(g["policy_id"])).all()
        kind = str(g["action_kind"]) if "action_kind" in g.files else "discrete"
        ra = g["ref_actions"][r * T:(r + 1) * T]                           # [T, B] or [T, B, num_actions]
        if kind == "discrete":
            np.testing.assert_array_equal(g[f"out{r}_actions"][..., 0], ra.T.astype(np.float32))
        else:  # every policy output is stored as f32 [B, T, num_actions] (shared_buffers.py:100-103)
            np.testing.assert_array_equal(g[f"out{r}_actions"], ra.transpose(1, 0, 2).astype(np.float32))
        np.testing.assert_array_equal(g[f"out{r}_action_logits"], g["in_logits"][r * T:(r + 1) * T].transpose(1, 0, 2))
        np.testing.assert_array_equal(g[f"out{r}_policy_version"], np.broadcast_to(g["in_versions"][r * T:(r + 1) * T], g[f"out{r}_policy_version"].shape))
    for k in ("ep_reward", "ep_len", "final_ep_reward", "final_ep_len", "final_last_rnn"):
        np.testing.assert_array_equal(st[k], g[k], err_msg=k)
    if kind == "discrete":
        _, la, _ = oracle.categorical(g["in_logits"].reshape(-1, int(g["A"])), g["ref_actions"].reshape(-1))
        np.testing.assert_allclose(la.reshape(g["ref_logp"].shape), g["ref_logp"], atol=1e-6)
    elif kind == "tuple":  # independent heads: the log-probabilities add up (action_distributions.py:231-243)
        la, o = 0.0, 0
        for h, n in enumerate(int(v) for v in g["head_sizes"]):
            la = la + oracle.categorical(g["in_logits"][..., o:o + n].reshape(-1, n), g["ref_actions"][..., h].reshape(-1))[1]
            o += n
        np.testing.assert_allclose(la.reshape(g["ref_logp"].shape), g["ref_logp"], atol=2e-6)
        # what the env was handed: the [B, heads] int32 array itself (preprocess_actions' all_discrete branch)
        assert not bool(g["env_seen_is_list"]) and g["env_seen_actions"].shape[1:] == g["ref_actions"].shape[1:]
    elif kind == "tuple_mixed":  # Tuple(Discrete, Box, Discrete): members add up, the env gets ONE ARRAY PER MEMBER
        heads = [int(v) for v in g["head_sizes"]]
        la, o, c = 0.0, 0, 0
        assert bool(g["env_seen_is_list"]) and g["ref_actions"].shape[-1] == sum(1 if n > 0 else -n for n in heads)
        for h, n in enumerate(heads):
            seen = g[f"env_seen_member{h}"]
            if n > 0:
                col = g["ref_actions"][..., c]
                la = la + oracle.categorical(g["in_logits"][..., o:o + n].reshape(-1, n), col.reshape(-1))[1]
                assert seen.dtype == np.int32 and seen.shape == col.shape   # batched_sampling.py:56-57
                np.testing.assert_array_equal(seen, col.astype(np.int32))
                o, c = o + n, c + 1
            else:
                D = -n
                mu, ls = g["in_logits"][..., o:o + D], g["in_logits"][..., o + D:o + 2 * D]
                np.testing.assert_array_equal(g["ref_actions"][..., c:c + D], mu)   # deterministic action = the mean
                sd = np.clip(np.exp(ls.astype(np.float64)), 1e-4, 1e4)
                la = la + (-np.log(sd) - 0.5 * np.log(2 * np.pi)).sum(-1).reshape(-1)
                assert seen.dtype == np.float32                                # Box member: f32 [B, D], not clipped here
                np.testing.assert_array_equal(seen, mu)
                o, c = o + 2 * D, c + D
        np.testing.assert_allclose(np.asarray(la).reshape(g["ref_logp"].shape), g["ref_logp"], atol=1e-5)
        # oracle.sample_tuple's column layout is the slab's: [discrete | D box columns | discrete]
        acts, _ = oracle.sample_tuple(g["in_logits"].reshape(-1, int(g["A"])), heads, seed=1, step=0)[:2]
        assert acts.shape[1] == g["ref_actions"].shape[-1]
    else:  # Box: deterministic action = the mean, log-density of a diagonal normal at its mean
        D = int(g["A"]) // 2
        np.testing.assert_array_equal(g["ref_actions"], g["in_logits"][..., :D])
        sd = np.clip(np.exp(g["in_logits"][..., D:].astype(np.float64)), 1e-4, 1e4)
        np.testing.assert_allclose((-np.log(sd) - 0.5 * np.log(2 * np.pi)).sum(-1), g["ref_logp"], atol=1e-5)

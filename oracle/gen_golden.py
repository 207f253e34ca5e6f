"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (Sample Factory at /root/reference) on seeded inputs.

TEST INFRASTRUCTURE.  Runs only in the build container (the reference is not present on the GPU box); the
produced fixtures are committed.  Usage:  python -m oracle.gen_golden   (from the repo root)

Every fixture stores the exact inputs handed to the reference function and the outputs it returned; the
reference entry point exercised is named in the `ref` field of each file.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from oracle import ref_import  # noqa: F401  (installs stubs, adds /root/reference to sys.path)
from oracle.weights import seeded_state  # our deterministic weight generator (shared with tests)

import gymnasium as gym  # the stub
from sample_factory.algo.learning.learner import Learner
from sample_factory.algo.utils.action_distributions import (
    CategoricalActionDistribution,
    ContinuousActionDistribution,
    TupleActionDistribution,
    get_action_distribution,
)
from sample_factory.algo.utils.env_info import EnvInfo
from sample_factory.algo.utils.model_sharing import ParameterServer
from sample_factory.algo.utils.rl_utils import gae_advantages
from sample_factory.algo.utils.running_mean_std import RunningMeanStdInPlace
from sample_factory.algo.utils.shared_buffers import alloc_trajectory_tensors
from sample_factory.algo.utils.tensor_dict import clone_tensordict
from sample_factory.cfg.arguments import parse_full_cfg, parse_sf_args
from sample_factory.model.model_utils import get_rnn_size
from sample_factory.utils.attr_dict import AttrDict

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
torch.set_num_threads(1)


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB)")


def make_cfg(extra, use_rnn=False):
    base = [] if use_rnn else ["--use_rnn=False", "--recurrence=1"]
    argv = ["--algo=APPO", "--env=synthetic", "--experiment=golden", "--train_dir=/tmp/sf_golden", "--device=cpu",
            "--serial_mode=True", "--seed=0"] + base + list(extra)
    parser, _ = parse_sf_args(argv)
    return parse_full_cfg(parser, argv)


def make_learner(cfg, obs_space, action_space, num_agents):
    env_info = EnvInfo(obs_space, action_space, num_agents, True, True, None, None, 1)
    pv = torch.zeros([1], dtype=torch.int32)
    ps = ParameterServer(0, pv, True)
    learner = Learner(cfg, env_info, pv, 0, ps)
    learner.init()
    return learner, env_info


def load_seeded(actor_critic, seed):
    """Overwrite trainable parameters with our seeded generator so that tests can regenerate them."""
    shapes = [(k, tuple(v.shape)) for k, v in actor_critic.named_parameters()]
    st = seeded_state(shapes, seed)
    with torch.no_grad():
        for k, p in actor_critic.named_parameters():
            p.copy_(torch.from_numpy(st[k]))
    return shapes


def fill_batch(b, g, A, continuous=False, p_done=0.1, p_timeout=0.0, p_other_policy=0.0, versions=None):
    for k, v in b["obs"].items():
        if v.dtype == torch.uint8:
            v.copy_(torch.randint(0, 256, v.shape, generator=g, dtype=torch.uint8))
        else:
            v.copy_(torch.randn(v.shape, generator=g))
    b["rnn_states"].zero_()
    if b["rnn_states"].shape[-1] > 1:  # recurrent models: non-trivial stored states, zero after a done step
        b["rnn_states"].copy_(torch.randn(b["rnn_states"].shape, generator=g) * 0.5)
    if continuous:
        b["actions"].copy_(torch.randn(b["actions"].shape, generator=g))
    else:
        b["actions"].copy_(torch.randint(0, A, b["actions"].shape, generator=g).float())
    b["action_logits"].copy_(torch.randn(b["action_logits"].shape, generator=g) * 0.5)
    b["log_prob_actions"].copy_(-torch.rand(b["log_prob_actions"].shape, generator=g) * 2.0 - 0.5)
    b["values"].copy_(torch.randn(b["values"].shape, generator=g))
    b["rewards"].copy_(torch.randn(b["rewards"].shape, generator=g))
    b["dones"].copy_(torch.rand(b["dones"].shape, generator=g) < p_done)
    b["time_outs"].copy_((torch.rand(b["dones"].shape, generator=g) < p_timeout) & b["dones"])
    b["policy_id"].zero_()
    if p_other_policy > 0:
        other = torch.rand(b["policy_id"].shape, generator=g) < p_other_policy
        b["policy_id"][other] = -1
    if versions is None:
        b["policy_version"].zero_()
    else:
        b["policy_version"].copy_(torch.randint(versions[0], versions[1], b["policy_version"].shape, generator=g).float())
    b["valids"].fill_(False)


def batch_arrays(b, prefix="in_"):
    out = {}
    for k, v in b.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                out[f"{prefix}obs_{kk}"] = vv.numpy().copy()
        else:
            out[prefix + k] = v.numpy().copy()
    return out


# ------------------------------------------------------------------------------------------------- GAE
def gen_gae():
    g = torch.Generator().manual_seed(100)
    arrays = {"ref": "sample_factory/algo/utils/rl_utils.py:78-94 gae_advantages"}
    cases = [(5, 7, 0.99, 0.95, 0.15, 0.0), (64, 32, 0.99, 0.95, 0.05, 0.1), (33, 128, 0.997, 0.9, 0.02, 0.3),
             (1, 1, 0.9, 1.0, 0.5, 0.0), (128, 32, 0.99, 0.95, 0.0, 0.0)]
    arrays["num_cases"] = len(cases)
    for i, (E, T, gamma, lam, p_done, p_inv) in enumerate(cases):
        rewards = torch.randn(E, T, generator=g)
        dones = torch.rand(E, T, generator=g) < p_done
        values = torch.randn(E, T + 1, generator=g)
        valids = torch.rand(E, T + 1, generator=g) >= p_inv
        valids[:, -1] = valids[:, -2]
        adv = gae_advantages(rewards.clone(), dones.clone(), values.clone(), valids.clone(), gamma, lam)
        arrays.update({f"c{i}_rewards": rewards.numpy(), f"c{i}_dones": dones.numpy(), f"c{i}_values": values.numpy(),
                       f"c{i}_valids": valids.numpy(), f"c{i}_gamma": gamma, f"c{i}_lambda": lam,
                       f"c{i}_adv": adv.contiguous().numpy()})
    save("gae", **arrays)


# ------------------------------------------------------------------------------------------------- RMS
def gen_rms():
    g = torch.Generator().manual_seed(200)
    rms = RunningMeanStdInPlace((1,))
    rms.train()
    arrays = {"ref": "sample_factory/algo/utils/running_mean_std.py:22-110 RunningMeanStdInPlace((1,))"}
    sizes = [64, 1000, 3, 4096]
    arrays["num_steps"] = len(sizes)
    for i, n in enumerate(sizes):
        x = torch.randn(n, generator=g) * (2.0 + i) + (i - 1.5)
        if i == 2:
            x = x * 100.0  # exercise the +-5 clip
        arrays[f"s{i}_x"] = x.numpy().copy()
        y = x.clone()
        rms(y)
        arrays[f"s{i}_normalized"] = y.numpy().copy()
        arrays[f"s{i}_stats"] = np.array([rms.running_mean.item(), rms.running_var.item(), rms.count.item()])
        z = (torch.randn(n, generator=g) * 3.0)
        arrays[f"s{i}_z"] = z.numpy().copy()
        zz = z.clone()
        rms(zz, denormalize=True)
        arrays[f"s{i}_denormalized"] = zz.numpy().copy()
    rms.eval()
    x = torch.randn(50, generator=g)
    y = x.clone()
    rms(y)
    arrays["eval_x"], arrays["eval_normalized"] = x.numpy(), y.numpy()
    arrays["eval_stats"] = np.array([rms.running_mean.item(), rms.running_var.item(), rms.count.item()])
    save("rms", **arrays)


# ------------------------------------------------------------------------------------------------- distributions
def gen_action_dist():
    g = torch.Generator().manual_seed(300)
    arrays = {"ref": "sample_factory/algo/utils/action_distributions.py:99-194,290-323; "
                     "known answers tests/algo/test_action_distributions.py:142-162"}
    # the reference test's known answer: softmax([0,1,2])
    logits = torch.tensor([[0.0, 1.0, 2.0]])
    d = CategoricalActionDistribution(logits)
    arrays["ka_logits"], arrays["ka_probs"] = logits.numpy(), d.probs.numpy()
    arrays["ka_entropy"] = d.entropy().numpy()
    for i, (N, A, scale) in enumerate([(31, 6, 1.0), (257, 18, 4.0), (64, 2, 10.0)]):
        z = torch.randn(N, A, generator=g) * scale
        zo = torch.randn(N, A, generator=g) * scale
        act = torch.randint(0, A, (N, 1), generator=g).float()
        d, do = CategoricalActionDistribution(z), CategoricalActionDistribution(zo)
        arrays.update({f"cat{i}_logits": z.numpy(), f"cat{i}_old_logits": zo.numpy(), f"cat{i}_actions": act.numpy(),
                       f"cat{i}_log_probs": d.log_probs.numpy(), f"cat{i}_probs": d.probs.numpy(),
                       f"cat{i}_entropy": d.entropy().numpy(), f"cat{i}_log_prob_actions": d.log_prob(act).numpy(),
                       f"cat{i}_kl": d.kl_divergence(do).numpy(),
                       f"cat{i}_symkl_uniform": d.symmetric_kl_with_uniform_prior().numpy(),
                       f"cat{i}_argmax": torch.argmax(d.probs, dim=-1).numpy()})
    arrays["num_cat"] = 3
    for i, (N, D) in enumerate([(40, 8), (7, 1)]):
        p = torch.randn(N, 2 * D, generator=g)
        po = torch.randn(N, 2 * D, generator=g)
        act = torch.randn(N, D, generator=g)
        d, do = ContinuousActionDistribution(p), ContinuousActionDistribution(po)
        arrays.update({f"con{i}_params": p.numpy(), f"con{i}_old_params": po.numpy(), f"con{i}_actions": act.numpy(),
                       f"con{i}_log_prob_actions": d.log_prob(act).numpy(), f"con{i}_entropy": d.entropy().numpy(),
                       f"con{i}_kl": d.kl_divergence(do).numpy()})
    arrays["num_con"] = 2
    # Tuple of Discrete spaces (action_distributions.py:197-287)
    for i, heads in enumerate([(3, 5, 2), (6, 3)]):
        N, A = 45, sum(heads)
        space = gym.spaces.Tuple([gym.spaces.Discrete(n) for n in heads])
        z = torch.randn(N, A, generator=g) * 2.0
        zo = torch.randn(N, A, generator=g) * 2.0
        act = torch.cat([torch.randint(0, n, (N, 1), generator=g) for n in heads], dim=1).float()
        d, do = TupleActionDistribution(space, z), TupleActionDistribution(space, zo)
        arrays.update({f"tup{i}_heads": np.array(heads), f"tup{i}_logits": z.numpy(), f"tup{i}_old_logits": zo.numpy(),
                       f"tup{i}_actions": act.numpy(), f"tup{i}_log_prob_actions": d.log_prob(act).numpy(),
                       f"tup{i}_entropy": d.entropy().numpy(), f"tup{i}_kl": d.kl_divergence(do).numpy(),
                       f"tup{i}_symkl_uniform": d.symmetric_kl_with_uniform_prior().numpy()})
    arrays["num_tup"] = 2
    # masked categorical (obs["action_mask"]): masked_softmax / masked_log_softmax, action_distributions.py:84-96
    N, A = 300, 7
    z = torch.randn(N, A, generator=g) * 2.0
    mask = (torch.rand(N, A, generator=g) < 0.6).float()
    mask[:5] = 0.0          # rows with every action masked out
    mask[5:10] = 1.0
    d = CategoricalActionDistribution(z, mask)
    arrays.update(dict(mask_logits=z.numpy(), mask_mask=mask.numpy().astype(np.uint8), mask_probs=d.probs.numpy(),
                       mask_log_probs=d.log_probs.numpy()))
    save("action_dist", **arrays)


# ------------------------------------------------------------------------------------------------- learner
MLP_OBS = gym.spaces.Dict({"obs": gym.spaces.Box(-10, 10, (8,), np.float32)})
MLP_ARGS = ["--encoder_mlp_layers", "32", "32", "--nonlinearity=tanh", "--normalize_input=False"]


def capture_heads(learner):
    """Forward hooks that keep the loss-head inputs (action params, values) and let autograd retain their grads."""
    cap = {}

    def hook(name):
        def f(_m, _inp, out):
            out.retain_grad()
            cap[name] = out
        return f

    h1 = learner.actor_critic.action_parameterization.distribution_linear.register_forward_hook(hook("params"))
    h2 = learner.actor_critic.critic_linear.register_forward_hook(hook("values"))
    return cap, (h1, h2)


def gen_prepare_and_losses(only=None):
    variants = [
        dict(name="ff_default", args=[], E=16, T=8, A=6, fill={}),
        dict(name="ff_invalids", args=["--max_policy_lag=5", "--kl_loss_coeff=0.2"], E=16, T=8, A=6,
             fill=dict(p_other_policy=0.2, versions=(90, 100)), train_step=100),
        dict(name="ff_bootstrap_nonorm", args=["--normalize_returns=False", "--value_bootstrap=True",
                                                "--exploration_loss=symmetric_kl", "--exploration_loss_coeff=0.01"],
             E=12, T=16, A=4, fill=dict(p_timeout=0.5, p_done=0.2)),
        dict(name="ff_continuous", args=["--kl_loss_coeff=0.1", "--exploration_loss_coeff=0.002"], E=10, T=8, A=None,
             D=3, fill={}),
        dict(name="ff_vtrace", args=["--with_vtrace=True", "--normalize_returns=False", "--recurrence=8",
                                      "--vtrace_rho=0.9", "--vtrace_c=0.8"], E=12, T=8, A=5, fill={}),
        dict(name="ff_tuple", args=["--kl_loss_coeff=0.1", "--exploration_loss_coeff=0.01"], E=12, T=8, A=10,
             heads=(3, 5, 2), fill={}),
        dict(name="ff_tuple_symkl", args=["--exploration_loss=symmetric_kl", "--exploration_loss_coeff=0.02"], E=10,
             T=8, A=9, heads=(6, 3), fill=dict(p_other_policy=0.1)),
        # Tuple(Discrete(3), Box(2), Discrete(4)): TupleActionDistribution builds every member with get_action_distribution
        # (action_distributions.py:222-225); a negative entry -D of `heads` is a Box(D) member (2 D parameters, D action columns)
        dict(name="ff_tuple_mixed", args=["--kl_loss_coeff=0.1", "--exploration_loss_coeff=0.01"], E=12, T=8, A=11,
             heads=(3, -2, 4), fill=dict(p_other_policy=0.1)),
    ]
    for vi, var in enumerate(variants):
        if only and var["name"] not in only:
            continue
        E, T = var["E"], var["T"]
        continuous = var.get("A") is None
        action_space = gym.spaces.Box(-1, 1, (var["D"],), np.float32) if continuous else gym.spaces.Discrete(var["A"])
        if "heads" in var:
            action_space = gym.spaces.Tuple([gym.spaces.Discrete(n) if n > 0 else gym.spaces.Box(-1, 1, (-n,), np.float32)
                                             for n in var["heads"]])
        nb = 2
        cfg = make_cfg(MLP_ARGS + [f"--rollout={T}", f"--batch_size={E * T // nb}", f"--num_batches_per_epoch={nb}",
                                   "--num_epochs=1"] + var["args"])
        learner, env_info = make_learner(cfg, MLP_OBS, action_space, E)
        shapes = load_seeded(learner.actor_critic, seed=7)
        learner.train_step = var.get("train_step", 0)
        # give the returns normaliser a non-trivial state
        if cfg.normalize_returns:
            learner.actor_critic.returns_normalizer.running_mean[:] = 0.3
            learner.actor_critic.returns_normalizer.running_var[:] = 2.5
            learner.actor_critic.returns_normalizer.count[:] = 100.0
        g = torch.Generator().manual_seed(1000 + vi)
        b = alloc_trajectory_tensors(env_info, E, T, get_rnn_size(cfg), "cpu", False)
        fill_batch(b, g, var.get("A"), continuous=continuous, **var["fill"])
        if "heads" in var:  # one action per Discrete head, within its own range; D normal columns per Box(D) member
            b["actions"].copy_(torch.cat([torch.randint(0, n, (E, T, 1), generator=g).float() if n > 0 else
                                          torch.randn((E, T, -n), generator=g) for n in var["heads"]], dim=2))
        arrays = {"ref": "sample_factory/algo/learning/learner.py:943-1034 Learner._prepare_batch; "
                         ":537-669 Learner._calculate_losses", "argv": " ".join(var["args"]),
                  "param_seed": 7, "train_step": learner.train_step}
        arrays.update(batch_arrays(b))
        if "heads" in var:
            arrays["head_sizes"] = np.array(var["heads"])
        if cfg.normalize_returns:
            rn = learner.actor_critic.returns_normalizer
            arrays["in_rms"] = np.array([rn.running_mean.item(), rn.running_var.item(), rn.count.item()])
        bb = clone_tensordict(b)
        buff, size, num_invalids = learner._prepare_batch(bb)
        arrays["bootstrap_values"] = bb["values"][:, -1].numpy().copy()  # written by the reference at :967
        arrays["out_rewards"] = bb["rewards"].numpy().copy()  # mutated in place by value bootstrap (:990)
        arrays["out_valids_full"] = bb["valids"].numpy().copy()
        for k in ["valids", "actions", "log_prob_actions", "values", "rewards", "dones", "action_logits"]:
            arrays["pb_" + k] = buff[k].numpy().copy()
        if not cfg.with_vtrace:
            arrays["pb_advantages"] = buff["advantages"].numpy().copy()
            arrays["pb_returns"] = buff["returns"].numpy().copy()
        arrays["pb_num_invalids"] = num_invalids
        arrays["pb_size"] = size
        if cfg.normalize_returns:
            rn = learner.actor_critic.returns_normalizer
            arrays["out_rms"] = np.array([rn.running_mean.item(), rn.running_var.item(), rn.count.item()])

        # ---- _calculate_losses on the first minibatch (contiguous slice, learner.py:520-521)
        mb_size = cfg.batch_size
        mb = AttrDict(learner._get_minibatch(buff, slice(0, mb_size)))
        cap, hooks = capture_heads(learner)
        rec = {}
        orig_value_loss = learner._value_loss

        def spy_value_loss(new_values, old_values, target, clip_value, valids, num_inv):
            rec["targets"] = target.detach().clone()
            return orig_value_loss(new_values, old_values, target, clip_value, valids, num_inv)

        learner._value_loss = spy_value_loss
        (dist, policy_loss, exploration_loss, kl_old, kl_loss, value_loss, summ) = learner._calculate_losses(
            mb, num_invalids)
        loss = policy_loss + exploration_loss + kl_loss + value_loss
        for p in learner.actor_critic.parameters():
            p.grad = None
        loss.backward()
        for h in hooks:
            h.remove()
        learner._value_loss = orig_value_loss
        arrays["mb_size"] = mb_size
        arrays["l_params"] = cap["params"].detach().numpy().copy()
        arrays["l_values"] = cap["values"].detach().squeeze(-1).numpy().copy()
        if continuous:
            # the distribution sees params through torch.chunk; grads arrive on distribution_linear's output
            pass
        arrays["l_grad_params"] = cap["params"].grad.numpy().copy()
        arrays["l_grad_values"] = cap["values"].grad.squeeze(-1).numpy().copy()
        arrays["l_policy_loss"] = float(policy_loss)
        arrays["l_exploration_loss"] = float(exploration_loss)
        arrays["l_kl_loss"] = float(kl_loss)
        arrays["l_value_loss"] = float(value_loss)
        arrays["l_adv_mean"] = float(summ["adv_mean"])
        arrays["l_adv_std"] = float(summ["adv_std"])
        arrays["l_adv_normalized"] = summ["adv"].numpy().copy()
        arrays["l_ratio"] = summ["ratio"].detach().numpy().copy()
        arrays["l_targets"] = rec["targets"].numpy().copy()
        if kl_old is not None:
            arrays["l_kl_old"] = kl_old.detach().numpy().copy()
        arrays["num_params"] = sum(int(np.prod(s)) for _, s in shapes)
        save("learner_" + var["name"], **arrays)


def np_frames(seed, shape):
    """u8 frames too large to commit (C2 geometry): regenerated on both sides from numpy's PCG64 stream"""
    return np.random.default_rng(seed).integers(0, 256, size=shape, dtype=np.uint8)


def _to_double(x):
    if isinstance(x, dict):
        return type(x)({k: _to_double(v) for k, v in x.items()})
    return x.double() if torch.is_tensor(x) and x.is_floating_point() else x


def first_step_grads_fp64(learner, b, cfg, names, subsample):
    # NB: `learner` is consumed (its model is converted to double)
    """The gradient of the first SGD step once more, with the REFERENCE's own _calculate_losses + autograd executed in
    float64 (model.double(), minibatch cast to double after the fp32 _prepare_batch): the round-off-free value both
    fp32 implementations (torch-CPU in the reference, the HIP kernels here) approximate.  Lets the GPU test state how
    far each of them is from the truth instead of only how far they are from each other."""
    l64 = learner  # a fresh learner with the same seeded weights (the scripted conv head does not survive deepcopy)
    buff, _size, num_invalids = l64._prepare_batch(clone_tensordict(b))
    mb = AttrDict(l64._get_minibatch(buff, slice(0, cfg.batch_size)))
    l64.actor_critic.double()
    mb = AttrDict(_to_double(dict(mb)))
    (_d, policy_loss, exploration_loss, _kl_old, kl_loss, value_loss, _s) = l64._calculate_losses(mb, num_invalids)
    loss = policy_loss + exploration_loss + kl_loss + value_loss
    for p_ in l64.actor_critic.parameters():
        p_.grad = None
    loss.backward()
    out = {"g1_fp64_loss": float(loss)}
    g = dict(l64.actor_critic.named_parameters())
    print("  fp64 loss", float(loss), [k for k in names if g[k].grad is None])
    for k in names:
        assert g[k].grad.dtype == torch.float64
        out["g1_fp64_" + k] = g[k].grad.numpy().reshape(-1)[::subsample].copy()
    out["g1_fp64_norm"] = float(torch.sqrt(sum((g[k].grad ** 2).sum() for k in names)))
    return out


def full_train_fp64(l64, b, cfg, names, subsample):
    # NB: `l64` is consumed (its model is converted to double)
    """The WHOLE Learner.train replay once more with the reference's own `_train` loop (losses, autograd, clip_grad_norm_,
    torch.optim.Adam, LR schedule) executed in float64: `_prepare_batch` in fp32 as the replay does, then model.double()
    and every minibatch cast to double on its way out of `_get_minibatch`.  delta64_<param> = the weight change both fp32
    implementations approximate: the GPU test states how far EACH of them is from it (tests/test_gpu_parity_c2_c5.py)
    instead of bounding their mutual distance with a wide floor."""
    buff, experience_size, num_invalids = l64._prepare_batch(clone_tensordict(b))
    l64.actor_critic.double()
    sd0 = {k: v.clone() for k, v in l64.actor_critic.state_dict().items()}
    # how many outputs of every ReLU are positive, per SGD step: a ReLU network's gradient is discontinuous where a
    # pre-activation crosses zero, so an fp32 implementation that lands on the other side of ONE of these ~1e7 activations
    # than float64 does moves the weight gradients below it by ~1e-3 of their largest element — the replay test counts such
    # flips instead of hiding them in a wide tolerance.  (The conv encoder is a torch.jit.script module: no forward hooks;
    # the counts come from the same layers evaluated functionally on the minibatch with the current float64 weights.)
    relu_pos = []
    conv_arch = getattr(cfg, "encoder_conv_architecture", None)
    count_relu = cfg.nonlinearity == "relu" and conv_arch == "convnet_atari" and "obs" in b["obs"] and b["obs"]["obs"].dim() == 5
    orig = l64._get_minibatch

    def get_minibatch64(gpu_buffer, indices):
        mb = AttrDict(_to_double(dict(orig(gpu_buffer, indices))))
        if count_relu:
            import torch.nn.functional as F
            sd_ = l64.actor_critic.state_dict()
            pfx = "encoder.encoders.obs.enc."
            with torch.no_grad():
                x, row = mb["normalized_obs"]["obs"], []
                for i, stride in ((0, 4), (2, 2), (4, 1)):
                    x = F.relu(F.conv2d(x, sd_[pfx + f"conv_head.{i}.weight"], sd_[pfx + f"conv_head.{i}.bias"], stride=stride))
                    row.append(int((x > 0).sum()))
                x = F.relu(F.linear(x.reshape(x.shape[0], -1), sd_[pfx + "mlp_layers.0.weight"], sd_[pfx + "mlp_layers.0.bias"]))
                row.append(int((x > 0).sum()))
            relu_pos.append(row)
        return mb

    l64._get_minibatch = get_minibatch64
    l64._train(buff, cfg.batch_size, experience_size, num_invalids)
    sd = l64.actor_critic.state_dict()
    out = {}
    if relu_pos:
        out["relu_pos64"] = np.array(relu_pos, dtype=np.int64)
        print("  fp64 ReLU positives per SGD step:", out["relu_pos64"].tolist())
    for k in names:
        assert sd[k].dtype == torch.float64
        out["delta64_" + k] = (sd[k] - sd0[k]).numpy().reshape(-1)[::subsample].copy()
    return out


def gen_train(name, obs_space, model_args, E, T, A, nb, epochs, extra=(), param_seed=3, subsample=1, use_rnn=False,
              box_dims=0, obs_seed=None, p_other_policy=None, fill_extra=None, fp64_first_step=False):
    cfg = make_cfg(list(model_args) + [f"--rollout={T}", f"--batch_size={E * T // nb}",
                                       f"--num_batches_per_epoch={nb}", f"--num_epochs={epochs}"] + list(extra),
                   use_rnn=use_rnn)
    action_space = gym.spaces.Box(-1, 1, (box_dims,), np.float32) if box_dims else gym.spaces.Discrete(A)
    learner, env_info = make_learner(cfg, obs_space, action_space, E)
    shapes = load_seeded(learner.actor_critic, seed=param_seed)
    g = torch.Generator().manual_seed(4242)
    b = alloc_trajectory_tensors(env_info, E, T, get_rnn_size(cfg), "cpu", False)
    if p_other_policy is None:
        p_other_policy = 0.05 if "inv" in name else 0.0
    fill_batch(b, g, A, continuous=bool(box_dims), p_done=0.08, p_other_policy=p_other_policy, **(fill_extra or {}))
    if obs_seed is not None:
        b["obs"]["obs"].copy_(torch.from_numpy(np_frames(obs_seed, tuple(b["obs"]["obs"].shape))))
    arrays = {"ref": "sample_factory/algo/learning/learner.py:1036-1067 Learner.train (prepare_batch + _train: "
                     "losses, backward, clip_grad_norm_, torch.optim.Adam)", "argv": " ".join(list(model_args) + list(extra)),
              "param_seed": param_seed, "E": E, "T": T, "A": A, "num_batches": nb, "num_epochs": epochs,
              "subsample": subsample}
    arrays.update(batch_arrays(b))
    if obs_seed is not None:
        del arrays["in_obs_obs"]
        arrays["obs_seed"] = obs_seed
        arrays["obs_crc"] = int(b["obs"]["obs"].long().sum())
    sd0 = {k: v.clone() for k, v in learner.actor_critic.state_dict().items()}
    if fp64_first_step:
        l64, _ = make_learner(cfg, obs_space, action_space, E)
        load_seeded(l64.actor_critic, seed=param_seed)
        arrays.update(first_step_grads_fp64(l64, b, cfg, [k for k, _ in shapes], subsample))
        l64b, _ = make_learner(cfg, obs_space, action_space, E)
        load_seeded(l64b.actor_critic, seed=param_seed)
        arrays.update(full_train_fp64(l64b, b, cfg, [k for k, _ in shapes], subsample))
    # record per-SGD-step grad norms by wrapping clip_grad_norm_
    norms = []
    orig_clip = torch.nn.utils.clip_grad_norm_
    first_grads = {}

    relu_pos32 = []
    count32 = (fp64_first_step and nb == 1 and cfg.nonlinearity == "relu" and
               getattr(cfg, "encoder_conv_architecture", None) == "convnet_atari" and b["obs"]["obs"].dim() == 5)

    def spy_clip(params, max_norm, *a, **k):
        if not norms:  # gradient of the FIRST SGD step as the reference's fp32 autograd produced it (before clipping)
            for kname, p_ in learner.actor_critic.named_parameters():
                first_grads[kname] = p_.grad.detach().clone()
        if count32:  # positive ReLU outputs of the reference's OWN fp32 forward of this SGD step (one minibatch = the dataset):
            # how many activations the reference itself puts on the other side of zero than float64 (relu_pos64)
            import torch.nn.functional as F
            sd_ = learner.actor_critic.state_dict()
            pfx = "encoder.encoders.obs.enc."
            with torch.no_grad():
                row = []
                x = b["obs"]["obs"][:, :-1].reshape(-1, *b["obs"]["obs"].shape[2:]).float().mul_(1.0 / cfg.obs_scale)
                for i, stride in ((0, 4), (2, 2), (4, 1)):
                    x = F.relu(F.conv2d(x, sd_[pfx + f"conv_head.{i}.weight"], sd_[pfx + f"conv_head.{i}.bias"], stride=stride))
                    row.append(int((x > 0).sum()))
                x = F.relu(F.linear(x.reshape(x.shape[0], -1), sd_[pfx + "mlp_layers.0.weight"], sd_[pfx + "mlp_layers.0.bias"]))
                row.append(int((x > 0).sum()))
            relu_pos32.append(row)
        n = orig_clip(params, max_norm, *a, **k)
        norms.append(float(n))
        return n

    torch.nn.utils.clip_grad_norm_ = spy_clip
    stats = learner.train(clone_tensordict(b))
    torch.nn.utils.clip_grad_norm_ = orig_clip
    arrays["grad_norms"] = np.array(norms)
    if relu_pos32:
        arrays["relu_pos32"] = np.array(relu_pos32, dtype=np.int64)
        print("  fp32 (reference) ReLU positives per SGD step:", relu_pos32)
    for k, _ in shapes:
        arrays["g1_" + k] = first_grads[k].numpy().reshape(-1)[::subsample].copy()
    arrays["curr_lr"] = float(learner.curr_lr)
    arrays["train_step"] = learner.train_step
    arrays["env_steps"] = stats["learner_env_steps"]
    sd = learner.actor_critic.state_dict()
    opt = learner.optimizer.state_dict()["state"]
    for i, (k, shape) in enumerate(shapes):
        arrays["after_" + k] = sd[k].numpy().reshape(-1)[::subsample].copy()
        arrays["delta_" + k] = (sd[k].double() - sd0[k].double()).numpy().reshape(-1)[::subsample].copy()
        arrays["m_" + k] = opt[i]["exp_avg"].numpy().reshape(-1)[::subsample].copy()
        arrays["v_" + k] = opt[i]["exp_avg_sq"].numpy().reshape(-1)[::subsample].copy()
        arrays["sum_after_" + k] = float(sd[k].double().sum())
    arrays["param_names"] = np.array([k for k, _ in shapes])
    arrays["param_shapes"] = np.array([str(s) for _, s in shapes])
    rn = learner.actor_critic.returns_normalizer
    if rn is not None:
        arrays["out_rms"] = np.array([rn.running_mean.item(), rn.running_var.item(), rn.count.item()])
    if cfg.normalize_input:  # obs normaliser state + an eval-mode forward with the post-training weights and statistics
        on = learner.actor_critic.obs_normalizer.running_mean_std.running_mean_std["obs"]
        arrays["obsn_mean"] = on.running_mean.numpy().reshape(-1)[::subsample].copy()
        arrays["obsn_var"] = on.running_var.numpy().reshape(-1)[::subsample].copy()
        arrays["obsn_count"] = float(on.count.item())
        ac = learner.actor_critic
        ac.eval()
        with torch.no_grad():
            o = {k: v[:, 0].clone() for k, v in b["obs"].items()}
            nobs = ac.normalize_obs(o)
            if use_rnn:  # one inference step through the core as well (model/core.py:37-64)
                x, new_rnn = ac.forward_core(ac.forward_head(nobs), b["rnn_states"][:, 0].clone())
                res = ac.forward_tail(x, values_only=False, sample_actions=False)
                arrays["eval_new_rnn_states"] = new_rnn.numpy().copy()
            else:
                res = ac.forward_tail(ac.forward_head(nobs), values_only=False, sample_actions=False)
        arrays["eval_logits"] = res["action_logits"].numpy().copy()
        arrays["eval_values"] = res["values"].numpy().copy()
    save("train_" + name, **arrays)


C2_MODEL_ARGS = ["--encoder_conv_architecture=convnet_atari", "--nonlinearity=relu", "--obs_scale=255.0",
                 "--normalize_input=False", "--encoder_conv_mlp_layers", "512"]


def gen_train_cnn84():
    """The BASELINE configs[1] geometry at the Learner.train level: real Nature-CNN on 84x84x4 u8 frames, 512-wide fc,
    E=64 trajectories x T=32, 2 minibatches of 1024 samples (the launch sizes that take the LDS-DMA / LDS-image kernel
    families on the GPU), 5 % rows of another policy + stale versions (invalid rows)."""
    obs = gym.spaces.Dict({"obs": gym.spaces.Box(0, 255, (4, 84, 84), np.uint8)})
    gen_train("cnn84", obs, C2_MODEL_ARGS, E=64, T=32, A=6, nb=2, epochs=1, subsample=37, obs_seed=8484,
              p_other_policy=0.05, extra=["--exploration_loss_coeff=0.01"], fp64_first_step=True)


def gen_train_cnn84_norm():
    """train_cnn84 with cfg.normalize_input=True (the reference's default, cfg/cfg.py:337-341): the observation normaliser's
    per-pixel running statistics are updated on the dataset and applied to every frame in front of conv1
    (utils/normalize.py:40-70, running_mean_std.py:64-110) — the replay the loader-fused conv1 (sf_conv_fwd_norm) is held to."""
    obs = gym.spaces.Dict({"obs": gym.spaces.Box(0, 255, (4, 84, 84), np.uint8)})
    args = [a for a in C2_MODEL_ARGS if not a.startswith("--normalize_input")] + ["--normalize_input=True"]
    gen_train("cnn84_norm", obs, args, E=64, T=32, A=6, nb=2, epochs=1, subsample=37, obs_seed=8485,
              p_other_policy=0.05, extra=["--exploration_loss_coeff=0.01"], fp64_first_step=True)


def gen_train_cnn84_32k():
    """The reference's Learner.train at the LAUNCH SIZE of the headline number: Nature-CNN on 84x84x4 u8 frames, E = 1024
    trajectories x T = 32 = ONE minibatch of 32768 samples (BASELINE configs[1] / NS-2's batch_size), two epochs = two SGD
    steps on it (the second one sees the first one's Adam update), 2 % rows of another policy.  fp32 replay + the same
    loop in float64 + ReLU-positive counts, as train_cnn84 (which has 2 minibatches x 1024)."""
    torch.set_num_threads(max(1, os.cpu_count() or 1))
    obs = gym.spaces.Dict({"obs": gym.spaces.Box(0, 255, (4, 84, 84), np.uint8)})
    gen_train("cnn84_32k", obs, C2_MODEL_ARGS, E=1024, T=32, A=6, nb=1, epochs=2, subsample=37, obs_seed=32768,
              p_other_policy=0.02, extra=["--exploration_loss_coeff=0.01"], fp64_first_step=True)
    torch.set_num_threads(1)


C5_OBS = gym.spaces.Dict({"obs": gym.spaces.Box(-10, 10, (27,), np.float32)})
# sf_examples/mujoco/mujoco_params.py:1-38 + what BASELINE configs[4] adds (LSTM core, V-trace; SURVEY.md §8d "C5")
C5_MODEL_ARGS = ["--encoder_mlp_layers", "64", "64", "--nonlinearity=tanh", "--normalize_input=True",
                 "--adaptive_stddev=False", "--use_rnn=True", "--rnn_type=lstm", "--rnn_size=512"]
C5_ALGO_ARGS = ["--recurrence=32", "--with_vtrace=True", "--normalize_returns=False", "--kl_loss_coeff=0.1",
                "--value_bootstrap=True", "--max_grad_norm=3.5", "--ppo_clip_ratio=0.2", "--value_loss_coeff=1.3",
                "--exploration_loss_coeff=0.0", "--learning_rate=0.00295", "--gamma=0.99"]


def gen_train_c5(rnn_type="lstm"):
    """rnn_type="gru": the same replay with the reference's DEFAULT core type (cfg.py rnn_type=gru) at the same width.
    BASELINE configs[4] as ONE Learner.train replay: Ant-shaped obs f32[27], Box(8) with the learned stddev,
    MLP[64,64] tanh encoder, LSTM-512 core (packed-sequence BPTT in the reference, rnn_utils.py:114-158), V-trace
    (learner.py:601-640) + KL loss + value bootstrap on time-outs, input normalisation, invalid rows."""
    model_args = [a.replace("--rnn_type=lstm", f"--rnn_type={rnn_type}") for a in C5_MODEL_ARGS]
    gen_train("c5" if rnn_type == "lstm" else f"c5_{rnn_type}", C5_OBS, model_args, E=32, T=32, A=None, nb=2, epochs=1, subsample=37, use_rnn=True, box_dims=8,
              p_other_policy=0.04, extra=C5_ALGO_ARGS, fill_extra=dict(p_timeout=0.3), fp64_first_step=True)


def gen_ref_checkpoint():
    """A checkpoint WRITTEN BY THE REFERENCE (Learner.save after one Learner.train, learner.py:323-363) + the logits /
    values its model produces on a probe batch: this engine must resume from it (same progress counters, weights, Adam
    moments, normaliser statistics) and reproduce the outputs — SURVEY.md §8(f1)."""
    import glob
    import shutil
    E, T, A, nb = 16, 8, 6, 2
    args = ["--encoder_mlp_layers", "32", "32", "--nonlinearity=elu", "--normalize_input=True"]
    shutil.rmtree("/tmp/sf_golden/ckpt", ignore_errors=True)
    argv = ["--algo=APPO", "--env=synthetic", "--experiment=ckpt", "--train_dir=/tmp/sf_golden", "--device=cpu",
            "--serial_mode=True", "--seed=0", "--use_rnn=False", "--recurrence=1", f"--rollout={T}",
            f"--batch_size={E * T // nb}", f"--num_batches_per_epoch={nb}", "--num_epochs=2"] + args
    parser, _ = parse_sf_args(argv)
    cfg = parse_full_cfg(parser, argv)
    learner, env_info = make_learner(cfg, MLP_OBS, gym.spaces.Discrete(A), E)
    load_seeded(learner.actor_critic, seed=11)
    g = torch.Generator().manual_seed(777)
    b = alloc_trajectory_tensors(env_info, E, T, get_rnn_size(cfg), "cpu", False)
    fill_batch(b, g, A)
    learner.train(clone_tensordict(b))
    learner.best_performance = 12.5
    assert learner.save()
    path = sorted(glob.glob("/tmp/sf_golden/ckpt/checkpoint_p0/checkpoint_*.pth"))[-1]
    shutil.copy(path, os.path.join(OUT, "ref_checkpoint_mlp.pth"))
    ac = learner.actor_critic
    ac.eval()
    probe = torch.randn((32, 8), generator=g)
    with torch.no_grad():
        res = ac.forward_tail(ac.forward_head(ac.normalize_obs({"obs": probe.clone()})), values_only=False,
                              sample_actions=False)
    save("ref_checkpoint_mlp", ref="learner.py:323-363 Learner.save (checkpoint written by the reference)",
         file_name=os.path.basename(path), train_step=learner.train_step, env_steps=learner.env_steps, probe=probe.numpy(),
         logits=res["action_logits"].numpy(), values=res["values"].numpy(), argv=" ".join(args), E=E, T=T, A=A,
         num_batches=nb, **batch_arrays(b))


def gen_model_fwd():
    """Nature-CNN actor-critic forward on u8 frames — model/encoder.py:90-150, model/actor_critic.py:160-195,
    utils/normalize.py:51-70 (obs_scale=255)."""
    obs_space = gym.spaces.Dict({"obs": gym.spaces.Box(0, 255, (4, 84, 84), np.uint8)})
    cfg = make_cfg(["--encoder_conv_architecture=convnet_atari", "--nonlinearity=relu", "--obs_scale=255.0",
                    "--normalize_input=False", "--rollout=4", "--batch_size=8", "--num_batches_per_epoch=1"])
    learner, env_info = make_learner(cfg, obs_space, gym.spaces.Discrete(6), 2)
    shapes = load_seeded(learner.actor_critic, seed=5)
    g = torch.Generator().manual_seed(77)
    obs = torch.randint(0, 256, (6, 4, 84, 84), generator=g, dtype=torch.uint8)
    ac = learner.actor_critic
    ac.eval()
    with torch.no_grad():
        nobs = ac.normalize_obs({"obs": obs})
        head = ac.forward_head(nobs)
        res = ac.forward_tail(head, values_only=False, sample_actions=False)
        conv1 = ac.encoder.encoders["obs"].enc.conv_head[0](nobs["obs"])
    save("model_fwd_atari", ref="model/actor_critic.py:160-195 ActorCriticSharedWeights.forward_head/forward_tail",
         param_seed=5, obs=obs.numpy(), head_sum=float(head.double().sum()), head_sample=head[:, ::37].numpy(),
         action_logits=res["action_logits"].numpy(), values=res["values"].numpy(),
         conv1_preact_sample=conv1[:, ::5, ::3, ::3].numpy(),
         param_names=np.array([k for k, _ in shapes]), param_shapes=np.array([str(s) for _, s in shapes]))


def gen_model_fwd_multi():
    """MultiInputEncoder (model/encoder.py:33-69): image + vector observation keys, per-key normalisation
    (utils/normalize.py:51-70: obs_scale only on "obs"), encodings concatenated in sorted key order."""
    obs_space = gym.spaces.Dict({"obs": gym.spaces.Box(0, 255, (4, 36, 36), np.uint8),
                                 "measurements": gym.spaces.Box(-1, 1, (5,), np.float32)})
    cfg = make_cfg(["--encoder_conv_architecture=convnet_impala", "--nonlinearity=relu", "--obs_scale=255.0",
                    "--normalize_input=False", "--encoder_conv_mlp_layers", "32", "--encoder_mlp_layers", "16", "16",
                    "--rollout=4", "--batch_size=8", "--num_batches_per_epoch=1"])
    learner, env_info = make_learner(cfg, obs_space, gym.spaces.Discrete(6), 2)
    shapes = load_seeded(learner.actor_critic, seed=9)
    g = torch.Generator().manual_seed(78)
    obs = torch.randint(0, 256, (6, 4, 36, 36), generator=g, dtype=torch.uint8)
    meas = torch.rand((6, 5), generator=g) * 2 - 1
    ac = learner.actor_critic
    ac.eval()
    with torch.no_grad():
        nobs = ac.normalize_obs({"obs": obs, "measurements": meas})
        head = ac.forward_head(nobs)
        res = ac.forward_tail(head, values_only=False, sample_actions=False)
    save("model_fwd_multi", ref="model/encoder.py:33-69 MultiInputEncoder inside ActorCriticSharedWeights", param_seed=9,
         obs=obs.numpy(), measurements=meas.numpy(), head=head.numpy(), action_logits=res["action_logits"].numpy(),
         values=res["values"].numpy(), param_names=np.array([k for k, _ in shapes]),
         param_shapes=np.array([str(s) for _, s in shapes]))


def gen_model_fwd_separate():
    """ActorCriticSeparateWeights (model/actor_critic.py:198-334, --actor_critic_share_weights=False): an encoder, core
    and decoder each for actor and critic; recurrent state = [actor | critic].  Feed-forward MLP, GRU and LSTM cores."""
    for tag, extra, S in (("ff", ["--use_rnn=False"], 2),
                          ("gru", ["--use_rnn=True", "--rnn_type=gru", "--rnn_size=12", "--recurrence=4"], 24),
                          ("lstm", ["--use_rnn=True", "--rnn_type=lstm", "--rnn_size=10", "--recurrence=4",
                                    "--decoder_mlp_layers", "14"], 40)):
        argv = ["--algo=APPO", "--env=synthetic", "--experiment=golden", "--train_dir=/tmp/sf_golden", "--device=cpu",
                "--serial_mode=True", "--seed=0", "--actor_critic_share_weights=False", "--encoder_mlp_layers", "16", "12",
                "--nonlinearity=tanh", "--normalize_input=False", "--rollout=4", "--batch_size=8",
                "--num_batches_per_epoch=1"] + extra
        parser, _ = parse_sf_args(argv)
        cfg = parse_full_cfg(parser, argv)
        assert get_rnn_size(cfg) == S, (get_rnn_size(cfg), S)
        learner, env_info = make_learner(cfg, MLP_OBS, gym.spaces.Discrete(5), 2)
        ac = learner.actor_critic
        assert type(ac).__name__ == "ActorCriticSeparateWeights"
        shapes = load_seeded(ac, seed=21)
        g = torch.Generator().manual_seed(79)
        obs = torch.randn((6, 8), generator=g)
        rnn = torch.randn((6, S), generator=g) * 0.5
        ac.eval()
        with torch.no_grad():
            nobs = ac.normalize_obs({"obs": obs})
            head = ac.forward_head(nobs)
            core, new_rnn = ac.forward_core(head, rnn)
            res = ac.forward_tail(core, values_only=False, sample_actions=False)
        save("model_fwd_separate_" + tag, ref="model/actor_critic.py:198-334 ActorCriticSeparateWeights", param_seed=21,
             argv=" ".join(extra), obs=obs.numpy(), rnn_states=rnn.numpy(), head=head.numpy(), core=core.numpy(),
             new_rnn_states=new_rnn.numpy(), action_logits=res["action_logits"].numpy(),
             values=res["values"].reshape(-1).numpy(), param_names=np.array([k for k, _ in shapes]),
             param_shapes=np.array([str(s) for _, s in shapes]))


def gen_minibatch_indices():
    """Learner._get_minibatches — learner.py:498-526: contiguous slices by default; shuffled = permutation of
    recurrence-aligned chunk starts expanded to full index runs, np.split into minibatches."""
    cfg = make_cfg(MLP_ARGS + ["--rollout=8", "--recurrence=4", "--batch_size=32", "--num_batches_per_epoch=4",
                               "--shuffle_minibatches=True", "--use_rnn=False"])
    learner, _ = make_learner(cfg, MLP_OBS, gym.spaces.Discrete(3), 16)
    np.random.seed(123)
    mbs = learner._get_minibatches(32, 128)
    np.random.seed(123)
    perm = np.random.permutation(np.arange(0, 128, 4))
    save("minibatch_indices", ref="learner.py:498-526 Learner._get_minibatches", experience_size=128, batch_size=32,
         recurrence=4, chunk_start_permutation=perm, minibatches=np.stack(mbs))


def gen_host_logic():
    """SliceMerger (batcher.py:22-86) op traces and BufferMgr (shared_buffers.py:152-239) bookkeeping numbers"""
    import json
    import random
    from sample_factory.algo.learning.batcher import SliceMerger
    from sample_factory.algo.utils.shared_buffers import BufferMgr
    rng = random.Random(7)
    traces = []
    for case in range(6):
        unit, total = rng.choice([4, 8, 16]), 256
        pieces = [slice(i, i + unit) for i in range(0, total, unit)]
        rng.shuffle(pieces)
        sm, ops, pending = SliceMerger(), [], list(pieces)
        while pending or sm.total_num:
            r = rng.random()
            if pending and (r < 0.6 or not sm.total_num):
                s = pending.pop()
                sm.merge_slices(s)
                ops.append(["merge", s.start, s.stop, sm.total_num, sorted(sm.slice_starts)])
            elif r < 0.8:
                n = rng.choice([unit, 2 * unit, 3 * unit, 64])
                got = sm.get_exactly(n)
                ops.append(["exactly", n, None if got is None else [got.start, got.stop], sm.total_num])
            else:
                n = rng.choice([unit // 2, unit, 5 * unit])
                got = sm.get_at_most(n)
                ops.append(["at_most", n, None if got is None else [got.start, got.stop], sm.total_num])
        traces.append(ops)
    mgr = []
    obs = gym.spaces.Dict({"obs": gym.spaces.Box(-1, 1, (4,), np.float32)})
    for args, agents in [(["--num_workers=1", "--num_envs_per_worker=1", "--async_rl=False", "--batch_size=512",
                           "--num_batches_per_epoch=2", "--rollout=8", "--serial_mode=True"], 128),
                         (["--num_workers=2", "--num_envs_per_worker=2", "--worker_num_splits=2", "--async_rl=True",
                           "--batch_size=256", "--num_batches_per_epoch=1", "--rollout=16", "--serial_mode=True",
                           "--num_batches_to_accumulate=2"], 32),
                         (["--num_workers=1", "--num_envs_per_worker=1", "--async_rl=True", "--batch_size=4096",
                           "--num_batches_per_epoch=4", "--rollout=32", "--serial_mode=True",
                           "--num_batches_to_accumulate=2"], 64),
                         (["--num_workers=4", "--num_envs_per_worker=8", "--batched_sampling=False", "--async_rl=True",
                           "--batch_size=128", "--num_batches_per_epoch=1", "--rollout=8", "--serial_mode=True",
                           "--worker_num_splits=2"], 1)]:
        cfg = make_cfg(MLP_ARGS + args)
        env_info = EnvInfo(obs, gym.spaces.Discrete(3), agents, False, False, None, None, 1)
        bm = BufferMgr(cfg, env_info)
        (dev, nbuf), = bm.buffers_per_device.items()
        q = bm.traj_buffer_queues[dev]
        items = []
        while not q.empty():
            it = q.get()
            items.append([it.start, it.stop] if isinstance(it, slice) else int(it))
        mgr.append(dict(argv=args, num_agents=agents, buffers_for_device=int(nbuf),
                        allocated=int(bm.traj_tensors_torch[dev]["rewards"].shape[0]),
                        trajectories_per_training_iteration=int(bm.trajectories_per_training_iteration),
                        sampling_trajectories_per_iteration=int(bm.sampling_trajectories_per_iteration),
                        max_batches_to_accumulate=int(bm.max_batches_to_accumulate), queue=items))
    json.dump(dict(ref="sample_factory/algo/learning/batcher.py:22-86; algo/utils/shared_buffers.py:152-239",
                   slice_merger_traces=traces, buffer_mgr=mgr),
              open(os.path.join(OUT, "host_logic.json"), "w"))
    print("  wrote host_logic.json")


# ------------------------------------------------------------------------------------------------- rollout half
class ScriptedBatchedEnv(gym.Env):
    """A batched env (num_agents = B) that replays pre-generated per-step outputs: obs (dict of arrays), rewards,
    terminated, truncated — as torch tensors, the way a GPU/vector env hands them to BatchedVecEnv (make_env.py:147-237).
    It records the actions it was stepped with (what preprocess_actions produced, batched_sampling.py:30-82)."""

    def __init__(self, script, obs_space, action_space):
        self.s, self.k = script, 0
        self.num_agents = script["rew"].shape[1]
        self.is_multiagent = True
        self.observation_space, self.action_space = obs_space, action_space
        self.seen_actions = []

    def _obs(self):
        return {key: torch.from_numpy(v[self.k].copy()) for key, v in self.s["obs"].items()}

    def reset(self, **kwargs):
        self.k = 0
        return self._obs(), [dict() for _ in range(self.num_agents)]

    def step(self, actions):
        a = actions if isinstance(actions, (list, tuple)) else [actions]
        self.seen_actions.append([np.asarray(x).copy() for x in a])
        k = self.k
        self.k += 1
        return (self._obs(), torch.from_numpy(self.s["rew"][k].copy()), torch.from_numpy(self.s["term"][k].copy()),
                torch.from_numpy(self.s["trunc"][k].copy()), [dict() for _ in range(self.num_agents)])


def gen_rollout_case(name, B, T, n_rollouts, obs_spec, A, rnn, reward_scale, reward_clip, async_rl, seed,
                     num_policies=1, worker_idx=0, gpu_actions=False, action_kind="discrete"):
    # action_kind: "discrete" = Discrete(A); "tuple" = Tuple(Discrete(n) for n in A); "box" = Box(A) (2*A action parameters)
    """Drive the reference's BatchedVectorEnvRunner (batched_sampling.py:85-392: init, update_trajectory_buffers,
    generate_policy_request, advance_rollouts, _process_rewards, _process_env_step, _finalize_trajectories) for
    n_rollouts consecutive rollouts with a scripted env and scripted policy outputs; dump the slab rows it wrote."""
    from sample_factory.algo.sampling.batched_sampling import BatchedVectorEnvRunner
    from sample_factory.algo.utils.env_info import extract_env_info
    from sample_factory.algo.utils.make_env import BatchedVecEnv
    from sample_factory.algo.utils.shared_buffers import BufferMgr
    from sample_factory.envs.env_utils import register_env
    from sample_factory.utils.timing import Timing

    rng = np.random.default_rng(seed)
    steps = n_rollouts * T
    script = dict(obs={})
    spaces_ = {}
    for key, (shape, dtype) in obs_spec.items():
        if dtype == np.uint8:
            script["obs"][key] = rng.integers(0, 256, (steps + 1, B) + shape, dtype=np.uint8)
            spaces_[key] = gym.spaces.Box(0, 255, shape, np.uint8)
        else:
            script["obs"][key] = rng.standard_normal((steps + 1, B) + shape).astype(np.float32)
            spaces_[key] = gym.spaces.Box(-10, 10, shape, np.float32)
    # rewards: heavy-tailed so that clipping is active on both sides; a few exact zeros
    rew = (rng.standard_normal((steps, B)) * 4.0).astype(np.float32)
    rew[rng.random((steps, B)) < 0.1] = 0.0
    term = rng.random((steps, B)) < 0.12
    trunc = rng.random((steps, B)) < 0.06           # some rows are terminated AND truncated
    term[T - 1, : max(1, B // 4)] = True            # done at t = T-1 of the first rollout: reset hits rnn_states[:, T]
    trunc[T - 1, B // 4: B // 2] = True
    if n_rollouts > 1:
        term[T, -1] = True                          # and at t = 0 of the second
    script.update(rew=rew, term=term, trunc=trunc)
    obs_space = gym.spaces.Dict(spaces_)
    if action_kind == "tuple":
        action_space = gym.spaces.Tuple([gym.spaces.Discrete(int(n_)) for n_ in A])
        heads, A = [int(n_) for n_ in A], int(sum(A))
    elif action_kind == "tuple_mixed":  # a negative entry -D is a Box(D) member (2 D action parameters, D action columns)
        action_space = gym.spaces.Tuple([gym.spaces.Discrete(int(n_)) if n_ > 0 else
                                         gym.spaces.Box(-1.0, 1.0, (-int(n_),), np.float32) for n_ in A])
        heads, A = [int(n_) for n_ in A], int(sum(n_ if n_ > 0 else -2 * n_ for n_ in A))
    elif action_kind == "box":
        action_space = gym.spaces.Box(-1.0, 1.0, (int(A),), np.float32)
        heads, A = [], 2 * int(A)
    else:
        action_space = gym.spaces.Discrete(A)
        heads = [int(A)]

    rnn_args = ["--use_rnn=False"] if rnn is None else ["--use_rnn=True", f"--rnn_type={rnn[0]}", f"--rnn_size={rnn[1]}"]
    argv = ["--algo=APPO", f"--env=scripted_{name}", "--experiment=golden", "--train_dir=/tmp/sf_golden", "--device=cpu",
            "--serial_mode=True", "--seed=0", f"--rollout={T}", f"--recurrence={T if rnn else 1}",
            f"--batch_size={B * T}", "--num_batches_per_epoch=1", f"--num_workers={max(1, worker_idx + 1)}",
            "--num_envs_per_worker=1", "--worker_num_splits=1", "--batched_sampling=True",
            f"--async_rl={async_rl}", f"--reward_scale={reward_scale}", f"--reward_clip={reward_clip}",
            f"--num_policies={num_policies}", "--encoder_mlp_layers", "16", "--env_gpu_observations=False",
            f"--env_gpu_actions={gpu_actions}"] + rnn_args
    parser, _ = parse_sf_args(argv)
    cfg = parse_full_cfg(parser, argv)
    env_box = {}

    def make_env(full_env_name, cfg=None, env_config=None, render_mode=None):
        env_box["env"] = ScriptedBatchedEnv(script, obs_space, action_space)
        env_box["env_config"] = dict(env_config)
        return env_box["env"]

    register_env(cfg.env, make_env)
    env_info = extract_env_info(BatchedVecEnv(ScriptedBatchedEnv(script, obs_space, action_space)), cfg)
    bm = BufferMgr(cfg, env_info)
    (dev,) = bm.traj_tensors_torch.keys()
    slab = bm.traj_tensors_torch[dev]
    R = slab["rnn_states"].shape[-1]
    runner = BatchedVectorEnvRunner(cfg, env_info, 1, worker_idx, 0, bm, dev, [None] * num_policies)
    timing = Timing()
    runner.init(timing)

    # scripted policy outputs (what InferenceWorker._prepare_policy_outputs_batched scatters into policy_output_tensors,
    # inference_worker.py:235-269): deterministic actions = argmax of the logits (action_distributions.py:73-81)
    logits = (rng.standard_normal((steps, B, A)) * 1.5).astype(np.float32)
    if action_kind == "box":
        logits[..., A // 2:] = logits[..., A // 2:] * 0.3 - 0.5   # log-stddevs
    tl = torch.from_numpy(logits)
    from sample_factory.algo.utils.action_distributions import argmax_actions
    dist = get_action_distribution(action_space, tl.reshape(-1, A))
    if action_kind == "tuple":  # per-head arg-max (TupleActionDistribution.argmax is written for ONE sample: enjoy.py)
        acts = torch.stack([argmax_actions(d_) for d_ in dist.distributions], dim=1)
    elif action_kind == "tuple_mixed":  # arg-max of a Discrete member, the means of a Box member, column-wise
        acts = torch.cat([argmax_actions(d_).reshape(steps * B, -1).float() for d_ in dist.distributions], dim=1)
    else:
        acts = argmax_actions(dist)                  # deterministic actions (action_distributions.py:73-81)
    acts = acts.reshape(steps * B, -1)               # [N, num_actions] as the actor-critic stores them
    logp = dist.log_prob(acts if action_kind != "discrete" else acts.reshape(-1, 1)).reshape(steps, B).numpy()
    acts = acts.reshape(steps, B, -1)
    values = rng.standard_normal((steps, B)).astype(np.float32)
    new_rnn = (rng.standard_normal((steps, B, R)) * 0.7).astype(np.float32)
    if rnn is None:  # ModelCoreIdentity hands the (zero) width-1 dummy state through (model/core.py:67-77)
        new_rnn[:] = 0.0
    versions = (3 + np.arange(steps) // 5).astype(np.float32)  # the policy version advances inside a rollout

    po = runner.policy_output_tensors
    out_rollouts, slices, reports_all, seen_obs, seen_rnn = [], [], [], [], []
    k = 0
    for r in range(n_rollouts):
        assert runner.update_trajectory_buffers(timing), "no free trajectory slice"
        sl = runner.curr_traj_slice
        complete = []
        for t in range(T):
            req = runner.generate_policy_request()
            assert req == {runner.policy_id: (sl, t)}
            # what the inference worker would read for this request (inference_worker.py:183-205)
            seen_obs.append({key: v[sl, t].numpy().copy() for key, v in slab["obs"].items()})
            seen_rnn.append(slab["rnn_states"][sl, t].numpy().copy())
            po["actions"].copy_(acts[k].float())
            po["action_logits"].copy_(tl[k])
            po["log_prob_actions"].copy_(torch.from_numpy(logp[k]))
            po["values"].copy_(torch.from_numpy(values[k]))
            po["policy_version"].fill_(float(versions[k]))
            po["new_rnn_states"].copy_(torch.from_numpy(new_rnn[k]))
            complete, reports = runner.advance_rollouts(runner.policy_id, timing)
            reports_all.extend(reports)
            k += 1
        assert complete == [dict(policy_id=runner.policy_id, traj_buffer_idx=sl)]
        slices.append([sl.start, sl.stop])
        out_rollouts.append({kk: (vv[sl].numpy().copy() if not isinstance(vv, dict) else
                                  {k2: v2[sl].numpy().copy() for k2, v2 in vv.items()}) for kk, vv in slab.items()})
        # the batcher releases the slice: after training in sync mode (batcher.py:224-226), right after the copy into a
        # training batch in async mode (batcher.py:214-218) -> the queue hands the slices out round-robin
        bm.traj_buffer_queues[dev].put(sl)

    arrays = dict(ref="sample_factory/algo/sampling/batched_sampling.py:85-392 BatchedVectorEnvRunner "
                      "(init / generate_policy_request / advance_rollouts / _process_rewards / _process_env_step / "
                      "_finalize_trajectories), preprocess_actions :30-82",
                  B=B, T=T, A=A, n_rollouts=n_rollouts, rnn_type="" if rnn is None else rnn[0],
                  rnn_size=0 if rnn is None else rnn[1], rnn_state_width=R, reward_scale=reward_scale,
                  reward_clip=reward_clip, async_rl=async_rl, policy_id=runner.policy_id, seed=seed,
                  gpu_actions=gpu_actions,
                  slab_rows=slab["rewards"].shape[0], slices=np.asarray(slices),
                  obs_keys=np.asarray(sorted(obs_spec.keys())),
                  env_config=np.asarray([env_box["env_config"][q] for q in ("worker_index", "vector_index", "env_id")]),
                  in_rew=rew, in_term=term, in_trunc=trunc, in_logits=logits, in_values=values, in_new_rnn=new_rnn,
                  in_versions=versions, ref_logp=logp,
                  ref_actions=acts.numpy() if action_kind != "discrete" else acts.numpy()[..., 0],
                  action_kind=action_kind, head_sizes=np.asarray(heads, np.int64),
                  # what env.step received (preprocess_actions): one array for Discrete / Box, a LIST of per-head arrays for
                  # a Tuple space -> stacked here as [steps, heads, B]
                  # (a Tuple with a Box member: the members differ in shape and dtype -> env_seen_member<i>, [steps, B(, D)])
                  env_seen_actions=(np.zeros(0) if action_kind == "tuple_mixed" else
                                    np.stack([np.stack(a) if len(a) > 1 else a[0] for a in env_box["env"].seen_actions])),
                  env_seen_is_list=bool(len(env_box["env"].seen_actions[0]) > 1),
                  env_seen_actions_dtype=str(env_box["env"].seen_actions[0][0].dtype),
                  **({f"env_seen_member{i}": np.stack([a[i] for a in env_box["env"].seen_actions])
                      for i in range(len(heads))} if action_kind == "tuple_mixed" else {}),
                  final_ep_reward=runner.curr_episode_reward.numpy(), final_ep_len=runner.curr_episode_len.numpy(),
                  final_last_rnn=runner.last_rnn_state.numpy())
    for key, v in script["obs"].items():
        arrays[f"in_obs_{key}"] = v
    for k_, (so, sr) in enumerate(zip(seen_obs, seen_rnn)):
        arrays[f"seen_rnn_{k_}"] = sr
        for key, v in so.items():
            arrays[f"seen_obs_{key}_{k_}"] = v
    for r, out in enumerate(out_rollouts):
        for kk, vv in out.items():
            if isinstance(vv, dict):
                for k2, v2 in vv.items():
                    arrays[f"out{r}_obs_{k2}"] = v2
            else:
                arrays[f"out{r}_{kk}"] = vv
    ep_rew = np.concatenate([rp["episodic"]["reward"] for rp in reports_all]) if reports_all else np.zeros(0, np.float32)
    ep_len = np.concatenate([rp["episodic"]["len"] for rp in reports_all]) if reports_all else np.zeros(0, np.int32)
    arrays.update(ep_reward=ep_rew, ep_len=ep_len,
                  ep_min_raw=np.concatenate([rp["episodic"]["min_raw_reward"] for rp in reports_all]),
                  ep_max_raw=np.concatenate([rp["episodic"]["max_raw_reward"] for rp in reports_all]),
                  report_policy_ids=np.asarray([rp["policy_id"] for rp in reports_all]))
    assert int((term | trunc).sum()) == len(ep_rew)
    save("rollout_" + name, **arrays)


def gen_rollout_tuple_mixed():
    """Tuple(Discrete(3), Box(2), Discrete(4)) through the reference's runner: slab layout of the action columns and the
    per-member list preprocess_actions hands the env (batched_sampling.py:51-59)"""
    gen_rollout_case("tuple_mixed", B=12, T=6, n_rollouts=2, obs_spec={"obs": ((11,), np.float32)}, A=(3, -2, 4), rnn=None,
                     reward_scale=1.0, reward_clip=1000.0, async_rl=False, seed=508, action_kind="tuple_mixed")


def gen_rollout():
    vec = {"obs": ((11,), np.float32)}
    gen_rollout_case("ff_sync", B=24, T=8, n_rollouts=3, obs_spec=vec, A=6, rnn=None, reward_scale=1.0, reward_clip=1000.0,
                     async_rl=False, seed=501)
    gen_rollout_case("gru_scale_clip", B=16, T=8, n_rollouts=2, obs_spec=vec, A=5, rnn=("gru", 24), reward_scale=0.3,
                     reward_clip=1.5, async_rl=False, seed=502)
    gen_rollout_case("lstm_async", B=16, T=6, n_rollouts=3, obs_spec=vec, A=4, rnn=("lstm", 16), reward_scale=2.5,
                     reward_clip=5.0, async_rl=True, seed=503, gpu_actions=True)
    gen_rollout_case("u8_image", B=8, T=4, n_rollouts=2, obs_spec={"obs": ((4, 12, 12), np.uint8)}, A=6, rnn=None,
                     reward_scale=0.01, reward_clip=0.02, async_rl=True, seed=504)
    gen_rollout_case("tuple_heads", B=12, T=6, n_rollouts=2, obs_spec=vec, A=(3, 4), rnn=None, reward_scale=1.0,
                     reward_clip=1000.0, async_rl=False, seed=506, action_kind="tuple")
    gen_rollout_case("box_actions", B=12, T=6, n_rollouts=2, obs_spec=vec, A=3, rnn=("gru", 8), reward_scale=1.0,
                     reward_clip=1000.0, async_rl=False, seed=507, action_kind="box", gpu_actions=True)
    gen_rollout_tuple_mixed()
    gen_rollout_case("multikey_policy1", B=8, T=5, n_rollouts=2,
                     obs_spec={"obs": ((7,), np.float32), "aux": ((2, 3, 3), np.uint8)}, A=3, rnn=("gru", 8),
                     reward_scale=1.0, reward_clip=1.0, async_rl=False, seed=505, num_policies=2, worker_idx=1)


def main():
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["gae", "rms", "dist", "learner", "train", "model", "mb", "cfg", "ckpt", "host", "rollout"]
    if "gae" in which:
        gen_gae()
    if "rms" in which:
        gen_rms()
    if "dist" in which:
        gen_action_dist()
    if "rollout_tuple_mixed" in which:
        gen_rollout_tuple_mixed()
    if "learner" in which:
        gen_prepare_and_losses()
    for w in which:  # learner:<case>[,<case>] = only those cases of the learner fixtures
        if w.startswith("learner:"):
            gen_prepare_and_losses(only=w.split(":", 1)[1].split(","))
    if "mb" in which:
        gen_minibatch_indices()
    if "ckpt" in which:
        gen_ref_checkpoint()
    if "cnn84" in which:
        gen_train_cnn84()
    if "cnn84_norm" in which:
        gen_train_cnn84_norm()
    if "cnn84_32k" in which:
        gen_train_cnn84_32k()
    if "c5" in which:
        gen_train_c5()
    if "c5_gru" in which:
        gen_train_c5("gru")
    if "rnn2" in which or "train" in which:  # two stacked recurrent layers (model/core.py:19-64 with rnn_num_layers = 2): state [B, 2, H(*2)]
        rnn2 = ["--encoder_mlp_layers", "32", "--nonlinearity=relu", "--normalize_input=False", "--use_rnn=True",
                "--rnn_size=32", "--rnn_num_layers=2", "--recurrence=8"]
        gen_train("gru2", MLP_OBS, rnn2 + ["--rnn_type=gru"], E=16, T=8, A=6, nb=2, epochs=1, use_rnn=True)
        gen_train("lstm2", MLP_OBS, rnn2 + ["--rnn_type=lstm", "--kl_loss_coeff=0.1"], E=16, T=16, A=6, nb=2, epochs=2,
                  use_rnn=True, extra=["--recurrence=8"])
    if "train" in which:
        gen_train("mlp", MLP_OBS, MLP_ARGS, E=16, T=8, A=6, nb=2, epochs=2)
        gen_train("mlp_inv", MLP_OBS, MLP_ARGS, E=16, T=8, A=6, nb=4, epochs=1, extra=["--kl_loss_coeff=0.1"])
        gen_train("mlp_lamb", MLP_OBS, MLP_ARGS, E=16, T=8, A=6, nb=2, epochs=2, extra=["--optimizer=lamb"])
        gen_train("mlp_nonadaptive", MLP_OBS, MLP_ARGS, E=16, T=8, A=6, nb=2, epochs=2, box_dims=3,
                  extra=["--adaptive_stddev=False", "--initial_stddev=0.7", "--kl_loss_coeff=0.1"])
        gen_train("mlp_nonadaptive_tanh", MLP_OBS, MLP_ARGS, E=16, T=8, A=6, nb=2, epochs=2, box_dims=3,
                  extra=["--adaptive_stddev=False", "--initial_stddev=0.7", "--kl_loss_coeff=0.1",
                         "--continuous_tanh_scale=2.0"])
        cnn_obs = gym.spaces.Dict({"obs": gym.spaces.Box(0, 255, (4, 36, 36), np.uint8)})
        gen_train("cnn36", cnn_obs, ["--encoder_conv_architecture=convnet_atari", "--nonlinearity=relu",
                                     "--obs_scale=255.0", "--normalize_input=False",
                                     "--encoder_conv_mlp_layers", "128"],
                  E=8, T=4, A=6, nb=2, epochs=1, subsample=7)
        gen_train("cnn36_norm", cnn_obs, ["--encoder_conv_architecture=convnet_atari", "--nonlinearity=relu",
                                          "--obs_scale=255.0", "--normalize_input=True",
                                          "--encoder_conv_mlp_layers", "128"],
                  E=8, T=4, A=6, nb=2, epochs=1, subsample=7)
        rnn_common = ["--encoder_mlp_layers", "32", "--nonlinearity=relu", "--normalize_input=False", "--use_rnn=True",
                      "--rnn_size=32", "--recurrence=8"]
        gen_train("gru", MLP_OBS, rnn_common + ["--rnn_type=gru"], E=16, T=8, A=6, nb=2, epochs=1, use_rnn=True)
        gen_train("lstm_inv", MLP_OBS, rnn_common + ["--rnn_type=lstm", "--kl_loss_coeff=0.1"], E=16, T=16, A=6, nb=2,
                  epochs=2, use_rnn=True, extra=["--recurrence=8"])
        gen_train_cnn84()
        gen_train_c5()
        gen_train("mlp_norm", MLP_OBS, ["--encoder_mlp_layers", "32", "32", "--nonlinearity=elu",
                                        "--normalize_input=True"], E=16, T=8, A=6, nb=2, epochs=1)
    if "model" in which:
        gen_model_fwd()
        gen_model_fwd_multi()
        gen_model_fwd_separate()
    if "klmb" in which or "train" in which:  # per-minibatch KL-adaptive learning rate (learner.py:46-85), both directions
        gen_train("mlp_klmb_down", MLP_OBS, MLP_ARGS, E=16, T=8, A=6, nb=2, epochs=2,
                  extra=["--lr_schedule=kl_adaptive_minibatch", "--lr_schedule_kl_threshold=1e-7", "--learning_rate=1e-3"])
        gen_train("mlp_klmb_up", MLP_OBS, MLP_ARGS, E=16, T=8, A=6, nb=2, epochs=2,
                  extra=["--lr_schedule=kl_adaptive_minibatch", "--lr_schedule_kl_threshold=10.0", "--learning_rate=1e-4"])
        gen_train("mlp_klep", MLP_OBS, MLP_ARGS, E=16, T=8, A=6, nb=2, epochs=3,   # per-EPOCH KL-adaptive schedule
                  extra=["--lr_schedule=kl_adaptive_epoch", "--lr_schedule_kl_threshold=1e-7", "--learning_rate=1e-3"])
    if "separate" in which:
        gen_model_fwd_separate()
    if "separate" in which or "train" in which:  # ActorCriticSeparateWeights through the reference's Learner.train
        sep = ["--encoder_mlp_layers", "32", "--nonlinearity=relu", "--normalize_input=False",
               "--actor_critic_share_weights=False"]
        gen_train("sep_gru", MLP_OBS, sep + ["--use_rnn=True", "--rnn_size=32", "--recurrence=8", "--rnn_type=gru"], E=16, T=8,
                  A=6, nb=2, epochs=1, use_rnn=True)
        gen_train("sep_mlp", MLP_OBS, sep, E=16, T=8, A=6, nb=2, epochs=2, extra=["--kl_loss_coeff=0.1"])
    if "host" in which:
        gen_host_logic()
    if "rollout" in which:
        gen_rollout()
    if "cfg" in which:
        import json
        p, a = parse_sf_args(["--algo=APPO", "--env=x", "--experiment=e"])
        json.dump(dict(vars(a)), open(os.path.join(OUT, "cfg_defaults.json"), "w"), indent=0, sort_keys=True)


if __name__ == "__main__":
    main()

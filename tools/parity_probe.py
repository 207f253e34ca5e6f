"""Layer-by-layer accuracy probe of the network kernels on the C2-geometry golden batch (GPU box).

Runs minibatch 0 of tests/golden/train_cnn84.npz through the native forward / loss / backward, then repeats the network
part in torch float64 ON THE SAME loss-head gradient and reports, per layer, max|ours - fp64| / max|fp64| for the forward
activation, the data gradient and the weight / bias gradients.  Localises round-off: which kernel is how far from the
exact value.   python tools/parity_probe.py [n_samples]
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.weights import seeded_state  # noqa: E402  (deterministic weights shared with the goldens; not the oracle's arithmetic)
from sample_factory_amd.algo.learning.learner import Learner, ParameterServer  # noqa: E402
from sample_factory_amd.algo.utils.env_info import EnvInfo  # noqa: E402
from sample_factory_amd.algo.utils.shared_buffers import alloc_trajectory_tensors  # noqa: E402
from sample_factory_amd.cfg.arguments import default_cfg  # noqa: E402
from sample_factory_amd.envs import spaces  # noqa: E402


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "train_cnn84.npz"))
    E, T, A, nb = int(g["E"]), int(g["T"]), int(g["A"]), int(g["num_batches"])
    cfg = default_cfg(use_rnn=False, recurrence=1, nonlinearity="relu", normalize_input=False, obs_scale=255.0,
                      encoder_conv_architecture="convnet_atari", encoder_conv_mlp_layers=[512], rollout=T,
                      batch_size=E * T // nb, num_batches_per_epoch=nb, num_epochs=1, seed=0, exploration_loss_coeff=0.01,
                      serial_mode=True, train_dir="/tmp/sf_probe", experiment="t")
    obs_space = spaces.Dict({"obs": spaces.Box(0, 255, (4, 84, 84), np.uint8)})
    env_info = EnvInfo(obs_space, spaces.Discrete(A), E)
    pv = torch.zeros(1, dtype=torch.int32)
    ln = Learner(cfg, env_info, pv, 0, ParameterServer(0, pv))
    ln.init()
    ac = ln.actor_critic
    st = seeded_state([(n, eval(s)) for n, s in zip(g["param_names"], g["param_shapes"])], int(g["param_seed"]))
    ac.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=False)
    batch = alloc_trajectory_tensors(env_info, E, T, 1, "cuda")
    for k in ["rnn_states", "actions", "action_logits", "log_prob_actions", "values", "policy_version", "rewards",
              "dones", "time_outs", "policy_id", "valids"]:
        batch[k].copy_(torch.from_numpy(g["in_" + k]))
    fr = np.random.default_rng(int(g["obs_seed"])).integers(0, 256, size=tuple(batch["obs"]["obs"].shape), dtype=np.uint8)
    batch["obs"]["obs"].copy_(torch.from_numpy(fr))
    buff, size, ninv = ln._prepare_batch(batch)
    mb = ln._get_minibatches(cfg.batch_size, size)[0]
    acts, g_heads, _ = ln._losses_native(buff, mb, ninv)
    index, offset, n = mb
    ac.backward(acts, g_heads, buff.obs, n, sample_stride=ac.obs_elems, index=index, offset=offset, traj_T=buff.T)
    torch.cuda.synchronize()
    ours_g = ac.flat_to_ref(ac.flat_grads)

    # ---- the same network in float64 (torch, on the GPU), driven by OUR loss-head gradient
    rows = torch.arange(offset, offset + n, device="cuda")
    x = batch["obs"]["obs"][rows // T, rows % T].double() / 255.0  # (x - 0) * (1/255): fp64 of the same normalisation
    sd = {k: v.double().cuda().requires_grad_(True) for k, v in ac.state_dict().items() if v.dtype == torch.float32}
    p = "encoder.encoders.obs.enc."
    z1 = F.conv2d(x, sd[p + "conv_head.0.weight"], sd[p + "conv_head.0.bias"], stride=4); z1.retain_grad()
    a1 = F.relu(z1)
    z2 = F.conv2d(a1, sd[p + "conv_head.2.weight"], sd[p + "conv_head.2.bias"], stride=2); z2.retain_grad()
    a2 = F.relu(z2)
    z3 = F.conv2d(a2, sd[p + "conv_head.4.weight"], sd[p + "conv_head.4.bias"], stride=1); z3.retain_grad()
    a3 = F.relu(z3)
    zf = F.linear(a3.flatten(1), sd[p + "mlp_layers.0.weight"], sd[p + "mlp_layers.0.bias"]); zf.retain_grad()
    f = F.relu(zf)
    v = F.linear(f, sd["critic_linear.weight"], sd["critic_linear.bias"])
    lg = F.linear(f, sd["action_parameterization.distribution_linear.weight"],
                  sd["action_parameterization.distribution_linear.bias"])
    heads = torch.cat([v, lg], dim=1)
    gh = g_heads[:, :1 + A].double()
    heads.backward(gh)

    def e(a, b):
        return float((a.double() - b).abs().max() / b.abs().max())

    rep = {"n": n}
    nhwc = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1])
    rep["fwd"] = dict(conv1=e(acts[0], nhwc(a1)), conv2=e(acts[1], nhwc(a2)), conv3=e(acts[2], nhwc(a3)),
                      fc=e(acts[3], f), heads=e(acts[4][:, :1 + A], heads))
    gb = dict(ac._bufs)
    if ac._ctx["train"].get("relu_mask0") is not None:  # conv1's ReLU mask is applied inside its weight-gradient kernel
        gb[("g", 0)] = gb[("g", 0)] * (acts[0] > 0)
    rep["dgrad (gradient wrt the layer's pre-activation)"] = dict(
        conv1=e(gb[("g", 0)], nhwc(z1.grad)), conv2=e(gb[("g", 1)], nhwc(z2.grad)),
        conv3=e(gb[("g", 2)].view(-1, 64), nhwc(z3.grad)), fc=e(gb[("g", 3)], zf.grad))
    rep["wgrad"] = {k: e(ours_g[k].cuda(), sd[k].grad) for k in ours_g}
    # ReLU masks: activations whose sign the fp32 forward decides differently from float64 (|z| below its round-off)
    flips = {}
    for name, ours_a, z in [("conv1", acts[0], z1), ("conv2", acts[1], z2), ("conv3", acts[2], z3), ("fc", acts[3], zf)]:
        zz = nhwc(z) if z.dim() == 4 else z
        diff = (ours_a > 0) != (zz > 0)
        flips[name] = dict(activations=int(zz.numel()), mask_flips=int(diff.sum()),
                           max_abs_preactivation_at_flips=float(zz[diff].abs().max()) if diff.any() else 0.0,
                           preactivation_scale=float(zz.abs().mean()))
    rep["relu_mask_flips_vs_fp64"] = flips
    # the same quantities from stock torch fp32 on the GPU (MIOpen / rocBLAS): what another fp32 implementation gets
    sd32 = {k: v.detach().float().requires_grad_(True) for k, v in sd.items()}
    y1 = F.relu(F.conv2d(x.float(), sd32[p + "conv_head.0.weight"], sd32[p + "conv_head.0.bias"], stride=4))
    y2 = F.relu(F.conv2d(y1, sd32[p + "conv_head.2.weight"], sd32[p + "conv_head.2.bias"], stride=2))
    y3 = F.relu(F.conv2d(y2, sd32[p + "conv_head.4.weight"], sd32[p + "conv_head.4.bias"], stride=1))
    ff = F.relu(F.linear(y3.flatten(1), sd32[p + "mlp_layers.0.weight"], sd32[p + "mlp_layers.0.bias"]))
    h32 = torch.cat([F.linear(ff, sd32["critic_linear.weight"], sd32["critic_linear.bias"]),
                     F.linear(ff, sd32["action_parameterization.distribution_linear.weight"],
                              sd32["action_parameterization.distribution_linear.bias"])], dim=1)
    h32.backward(g_heads[:, :1 + A].contiguous())
    rep["wgrad_torch_fp32_gpu"] = {k: e(sd32[k].grad, sd[k].grad) for k in ours_g}
    # isolate the weight-gradient kernels: feed them the EXACT (fp64 -> fp32) output gradient and input
    from sample_factory_amd import lib
    iso = {}
    for li, (zin, zg, name) in enumerate([(None, z1, "conv_head.0"), (a1, z2, "conv_head.2"), (a2, z3, "conv_head.4")]):
        L = ac.layers[li]
        dy = nhwc(zg.grad).float().contiguous()
        dw, db = torch.zeros_like(L.gw), torch.zeros_like(L.gb)
        d = L.desc
        if li == 0:
            dense = batch["obs"]["obs"][rows // T, rows % T].contiguous()
            ws = torch.empty(lib.conv_wgrad_workspace(n, d), dtype=torch.uint8, device="cuda")
            lib.conv_wgrad_raw(dense, ac.obs_elems, None, 0, dy, dw, db, n, d, ws)
        else:
            xin = nhwc(zin).float().contiguous()
            ws = torch.empty(lib.conv_wgrad_workspace(n, d), dtype=torch.uint8, device="cuda")
            lib.conv_wgrad_raw(xin, d.H * d.W * d.Cin, None, 0, dy, dw, db, n, d, ws)
        wref = sd[p + name + ".weight"].grad
        iso[name + ".weight"] = e(L.w_to_ref(dw).cuda(), wref)
        iso[name + ".bias"] = e(db, sd[p + name + ".bias"].grad)
        iso[name + ".bias_torch_sum_of_fp32_dy"] = e(dy.sum(0), sd[p + name + ".bias"].grad)
        iso[name + ".bias_fp64_sum_of_fp32_dy"] = e(dy.double().sum(0), sd[p + name + ".bias"].grad)
    rep["wgrad_kernels_on_exact_inputs"] = iso
    print(json.dumps(rep, indent=1))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "parity_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

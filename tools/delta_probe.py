"""Weight deltas of the C2-geometry Learner.train replay (tests/golden/train_cnn84.npz) three ways: the HIP path, the
reference's fp32 run (delta_*) and the reference's own code in float64 (delta64_*).  Per tensor max|a - b| / max|b| and,
for the elements where the HIP path is furthest from float64, the first-step gradients behind them.
   python tools/delta_probe.py"""
import os, sys, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.weights import seeded_state
from tests.test_gpu_parity_c2_c5 import _load_batch


def main():
    from sample_factory_amd.algo.learning.learner import Learner, ParameterServer
    from sample_factory_amd.algo.utils.env_info import EnvInfo
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    g = np.load(os.path.join(ROOT, "tests", "golden", "train_cnn84.npz"), allow_pickle=True)
    E, T, A, nb = int(g["E"]), int(g["T"]), int(g["A"]), int(g["num_batches"])
    cfg = default_cfg(use_rnn=False, recurrence=1, nonlinearity="relu", normalize_input=False, obs_scale=255.0,
                      encoder_conv_architecture="convnet_atari", encoder_conv_mlp_layers=[512], rollout=T,
                      batch_size=E * T // nb, num_batches_per_epoch=nb, num_epochs=int(g["num_epochs"]), seed=0,
                      exploration_loss_coeff=0.01, serial_mode=True, train_dir="/tmp/delta_probe", experiment="t",
                      record_grad_norm=True)
    obs_space = spaces.Dict({"obs": spaces.Box(0, 255, (4, 84, 84), np.uint8)})
    env_info = EnvInfo(obs_space, spaces.Discrete(A), E)
    st = seeded_state([(n, eval(s)) for n, s in zip(g["param_names"], g["param_shapes"])], int(g["param_seed"]))
    pv = torch.zeros(1, dtype=torch.int32)
    ln = Learner(cfg, env_info, pv, 0, ParameterServer(0, pv))
    ln.init()
    ac = ln.actor_critic
    ac.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=False)
    batch = _load_batch(g, env_info, E, T, 1)
    before = {k: v.clone() for k, v in ac.state_dict().items()}
    ln.train(batch)
    torch.cuda.synchronize()
    after = ac.state_dict()
    sub = int(g["subsample"])
    rep = {}
    for name in g["param_names"]:
        name = str(name)
        ours = (after[name].double() - before[name].double()).reshape(-1)[::sub].cpu().numpy()
        r32, r64 = g["delta_" + name], g["delta64_" + name]
        sc = float(np.abs(r64).max())
        e = np.abs(ours - r64)
        rep[name] = dict(ours_vs_fp64=float(e.max() / sc), ref32_vs_fp64=float(np.abs(r32 - r64).max() / sc),
                         ours_vs_ref32=float(np.abs(ours - r32).max() / sc),
                         ours_frac_within_2e3=float((e <= 2e-3 * np.abs(r64) + 1e-10).mean()),
                         ref32_frac_within_2e3=float((np.abs(r32 - r64) <= 2e-3 * np.abs(r64) + 1e-10).mean()))
        w = np.argsort(-e)[:4]
        rep[name]["worst"] = [dict(i=int(i), ours=float(ours[i]), ref32=float(r32[i]), fp64=float(r64[i]),
                                   g1_ref32=float(g["g1_" + name][i]), g1_fp64=float(g["g1_fp64_" + name][i])) for i in w]
        print(name, {k: (round(v, 6) if isinstance(v, float) else v) for k, v in rep[name].items() if k != "worst"})
        for x in rep[name]["worst"][:2]:
            print("    ", x)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rep, open(os.path.join(ROOT, "gpurun_out", "delta_probe.json"), "w"), indent=1)


if __name__ == "__main__":
    main()

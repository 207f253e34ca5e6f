"""Per-SGD-step gradients of the C2-geometry Learner.train replay under different kernel switches (one process each):
which step and which tensor differ between kernel families, and by how much.   python tools/grad_steps_probe.py"""
import os, sys, subprocess
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(out):
    from oracle.weights import seeded_state
    from tests.test_gpu_parity_c2_c5 import _load_batch
    from sample_factory_amd import lib
    from sample_factory_amd.algo.learning.learner import Learner, ParameterServer
    from sample_factory_amd.algo.utils.env_info import EnvInfo
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    g = np.load(os.path.join(ROOT, "tests", "golden", "train_cnn84.npz"), allow_pickle=True)
    E, T, A, nb = int(g["E"]), int(g["T"]), int(g["A"]), int(g["num_batches"])
    cfg = default_cfg(use_rnn=False, recurrence=1, nonlinearity="relu", normalize_input=False, obs_scale=255.0,
                      encoder_conv_architecture="convnet_atari", encoder_conv_mlp_layers=[512], rollout=T,
                      batch_size=E * T // nb, num_batches_per_epoch=nb, num_epochs=int(g["num_epochs"]), seed=0,
                      exploration_loss_coeff=0.01, serial_mode=True, train_dir="/tmp/gsp", experiment="t")
    env_info = EnvInfo(spaces.Dict({"obs": spaces.Box(0, 255, (4, 84, 84), np.uint8)}), spaces.Discrete(A), E)
    st = seeded_state([(n, eval(s)) for n, s in zip(g["param_names"], g["param_shapes"])], int(g["param_seed"]))
    pv = torch.zeros(1, dtype=torch.int32)
    ln = Learner(cfg, env_info, pv, 0, ParameterServer(0, pv))
    ln.init()
    ac = ln.actor_critic
    ac.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=False)
    batch = _load_batch(g, env_info, E, T, 1)
    snaps, orig = [], lib.adam_step

    def spy(*a, **k):
        torch.cuda.synchronize()
        gr = ac.flat_to_ref(ac.flat_grads)
        acts = ac._ctx["train"]["acts"]
        snaps.append(dict(grads={n: v.clone().cpu() for n, v in gr.items()},
                          act_sums=[float(x.double().sum()) if x is not None else 0.0 for x in acts],
                          act_pos=[int((x > 0).sum()) if x is not None else 0 for x in acts]))
        return orig(*a, **k)

    lib.adam_step = spy
    import sample_factory_amd.algo.learning.learner as L
    ln.train(batch)
    torch.cuda.synchronize()
    torch.save(snaps, out)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "one":
        one(sys.argv[2])
    else:
        modes = {"default": {}, "wgrad_img0": {"SF_WGRAD_IMG": "0"}, "fwd_img0": {"SF_FWD_IMG": "0"}}
        for m, env in modes.items():
            subprocess.run([sys.executable, os.path.abspath(__file__), "one", f"/tmp/gsp_{m}.pt"], env=dict(os.environ, **env), check=True)
        S = {m: torch.load(f"/tmp/gsp_{m}.pt") for m in modes}
        base = S["wgrad_img0"]
        for m in ("default", "fwd_img0"):
            for step in range(len(base)):
                print(f"== {m} vs wgrad_img0, SGD step {step + 1}: act_pos {S[m][step]['act_pos']} vs {base[step]['act_pos']}")
                for n in base[step]["grads"]:
                    if "weight" not in n:
                        continue
                    a, b = S[m][step]["grads"][n].double(), base[step]["grads"][n].double()
                    print(f"   {n[-40:]:40s} max|a-b|/max|b| = {float((a - b).abs().max() / b.abs().max()):.3e}")

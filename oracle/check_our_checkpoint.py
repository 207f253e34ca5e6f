"""Feed a checkpoint written by THIS engine (tests/golden/ours_checkpoint_mlp.pth) to the REFERENCE's Learner
(build container only; TEST INFRASTRUCTURE).   python -m oracle.check_our_checkpoint"""
import os
import shutil
import sys

import numpy as np
import torch

from oracle import ref_import  # noqa: F401
import gymnasium as gym  # the stub
from oracle.gen_golden import MLP_OBS, make_learner
from sample_factory.cfg.arguments import parse_full_cfg, parse_sf_args

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "ours_checkpoint_mlp.npz"))
    shutil.rmtree("/tmp/sf_golden/ours_ckpt", ignore_errors=True)
    d = "/tmp/sf_golden/ours_ckpt/checkpoint_p0"
    os.makedirs(d)
    shutil.copy(os.path.join(ROOT, "tests", "golden", "ours_checkpoint_mlp.pth"), os.path.join(d, str(g["file_name"])))
    argv = ["--algo=APPO", "--env=synthetic", "--experiment=ours_ckpt", "--train_dir=/tmp/sf_golden", "--device=cpu",
            "--serial_mode=True", "--seed=0", "--use_rnn=False", "--recurrence=1", "--rollout=8", "--batch_size=64",
            "--num_batches_per_epoch=2", "--num_epochs=2"] + str(g["argv"]).split()
    parser, _ = parse_sf_args(argv)
    cfg = parse_full_cfg(parser, argv)
    learner, _ = make_learner(cfg, MLP_OBS, gym.spaces.Discrete(6), 16)   # Learner.init() -> load_from_checkpoint
    assert learner.train_step == int(g["train_step"]) and learner.env_steps == int(g["env_steps"]), \
        (learner.train_step, learner.env_steps)
    st = learner.optimizer.state_dict()["state"]
    assert len(st) == len(list(learner.actor_critic.parameters())) and float(st[0]["step"]) == int(g["train_step"])
    ac = learner.actor_critic
    ac.eval()
    with torch.no_grad():
        res = ac.forward_tail(ac.forward_head(ac.normalize_obs({"obs": torch.from_numpy(g["probe"]).clone()})),
                              values_only=False, sample_actions=False)
    np.testing.assert_allclose(res["action_logits"].numpy(), g["logits"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(res["values"].numpy(), g["values"], atol=2e-5, rtol=1e-4)
    print("INTEROP OK", learner.train_step, learner.env_steps)


if __name__ == "__main__":
    main()

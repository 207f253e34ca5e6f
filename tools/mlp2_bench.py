"""timing of sf_mlp2_fwd at the config-5 rollout shape (2048 x 27 -> 64 -> 64, tanh, normalised input)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sample_factory_amd import lib
lib.load()
n, D, H = 2048, 27, 64
g = torch.Generator().manual_seed(0)
x = torch.randn((n, 33, D), generator=g).cuda()
w1, b1, w2, b2 = torch.randn((D, H), generator=g).cuda(), torch.zeros(H).cuda(), torch.randn((H, H), generator=g).cuda() / 8, torch.zeros(H).cuda()
mu, rstd, out = torch.zeros(D).cuda(), torch.ones(D).cuda(), torch.empty((n, H)).cuda()
f = lambda: lib.mlp2_fwd(x[:, 3], x.stride(0), n, D, 0.0, 1.0, mu, rstd, w1, b1, w2, b2, 2, out)
for _ in range(5): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): f()
e1.record(); torch.cuda.synchronize()
print(f"sf_mlp2_fwd {e0.elapsed_time(e1) / 200 * 1e3:.1f} us per launch")

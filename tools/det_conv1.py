import sys, os
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from sample_factory_amd import lib
d = lib.sf_conv_desc(Cin=4, H=84, W=84, Cout=32, KH=8, KW=8, stride=4, OH=20, OW=20, in_u8=1, relu=1, traj_T=0, sub_mean=0.0, inv_scale=1/255.0)
n = 512
g = torch.Generator().manual_seed(0)
x = torch.randint(0, 256, (n, 4, 84, 84), generator=g, dtype=torch.uint8).cuda()
w = (torch.randn((256, 32), generator=g) / 16).cuda(); b = torch.zeros(32).cuda()
outs = []
for rep in range(5):
    out = torch.empty((n * 400, 32), device="cuda")
    lib.conv_fwd(x, 4 * 84 * 84, None, 0, w, b, out, n, d)
    outs.append(out.clone())
print("deterministic:", all(torch.equal(outs[0], o) for o in outs))
perm = torch.randperm(n, generator=g).to(torch.int32).cuda()
outp = torch.empty((n * 400, 32), device="cuda")
lib.conv_fwd(x, 4 * 84 * 84, perm, 0, w, b, outp, n, d)
ref = outs[0].view(n, 400, 32)[perm.long()]
print("pairing-independent:", torch.equal(ref, outp.view(n, 400, 32)), (ref - outp.view(n, 400, 32)).abs().max().item())
# small-n (old kernel) vs img kernel
outs_small = torch.empty((n * 400, 32), device="cuda")
for i in range(0, n, 64):
    lib.conv_fwd(x, 4 * 84 * 84, None, i, w, b, outs_small[i * 400:], 64, d)
print("img vs im2col max abs diff:", (outs_small - outs[0]).abs().max().item(), "max val", outs[0].abs().max().item())

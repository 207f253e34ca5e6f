"""Per-kernel timing of the network kernels at the bench shapes (HIP events, many reps).  python tools/kbench.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sample_factory_amd import lib

def desc(Cin, H, W, Cout, K, S, u8=0):
    return lib.sf_conv_desc(Cin=Cin, H=H, W=W, Cout=Cout, KH=K, KW=K, stride=S, OH=(H-K)//S+1, OW=(W-K)//S+1, in_u8=u8,
                            relu=1, traj_T=0, sub_mean=0.0, inv_scale=1/255.0 if u8 else 1.0)

LAYERS = [("conv1", desc(4,84,84,32,8,4,1)), ("conv2", desc(32,20,20,64,4,2)), ("conv3", desc(64,9,9,64,3,1)),
          ("fc", desc(3136,1,1,512,1,1)), ("heads", desc(512,1,1,8,1,1))]

def timeit(fn, reps):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps

def digest(t):  # KBENCH_HASH=1: seeded inputs + a digest of every result (bit-identity of two builds of the library)
    import hashlib
    return hashlib.sha1(t.detach().cpu().numpy().tobytes()).hexdigest()[:12]

def main():
    which = sys.argv[1:] or ["fwd", "wgrad", "dgrad"]
    HASH = os.environ.get("KBENCH_HASH") == "1"
    if HASH: torch.manual_seed(0)
    ns = [int(v) for v in os.environ["KBENCH_NS"].split(",")] if os.environ.get("KBENCH_NS") else [4096, 32768]
    only = os.environ.get("KBENCH_LAYERS", "").split(",") if os.environ.get("KBENCH_LAYERS") else None
    for n in ns:
        for name, d in LAYERS:
            if only and name not in only: continue
            K = d.KH*d.KW*d.Cin
            M = n*d.OH*d.OW
            flops = 2.0*M*d.Cout*K
            if d.in_u8: x = torch.randint(0,256,(n,d.Cin,d.H,d.W),dtype=torch.uint8,device="cuda")
            else: x = torch.randn((n,d.H,d.W,d.Cin),device="cuda")
            w = torch.randn((K,d.Cout),device="cuda")/np.sqrt(K); b = torch.zeros(d.Cout,device="cuda")
            out = torch.empty((M,d.Cout),device="cuda"); dy = torch.randn((M,d.Cout),device="cuda")
            reps = 5 if n >= 32768 else (20 if n >= 4096 else 100)
            stride = d.Cin*d.H*d.W
            res = []
            if "fwd" in which:
                wsb = lib.conv_fwd_workspace(n, d); ws = torch.empty(max(wsb,16),dtype=torch.uint8,device="cuda") if wsb else None
                t = timeit(lambda: lib.conv_fwd(x, stride, None, 0, w, b, out, n, d, ws), reps); res.append(f"fwd {t*1e3:8.1f}us {flops/t/1e9:6.1f}TF" + (f" #{digest(out)}" if HASH else ""))
            if "fwd" in which and lib.conv_fwd_t_supported(n, d):
                wt = torch.empty((d.Cout, K), device="cuda"); lib.transpose(w, wt, K, d.Cout)
                assert torch.equal(wt, w.t().contiguous())
                ref = out.clone(); out.zero_()
                for cfg in os.environ.get("GLDS_CFGS", "0").split(","):
                    nb = lib.conv_fwd_t_workspace(n, d); wst = torch.empty(nb,dtype=torch.uint8,device="cuda") if nb else None
                    t = timeit(lambda: lib.conv_fwd_t(x, stride, wt, b, out, n, d, wst), reps)
                    err = (out - ref).abs().max().item() / ref.abs().max().item()
                    res.append(f"fwd_t {t*1e3:8.1f}us {flops/t/1e9:6.1f}TF relerr {err:.1e}" + (f" #{digest(out)}" if HASH else ""))
            if "wgrad" in which and n == 32768:
                dw = torch.empty_like(w); db = torch.empty_like(b)
                ws = torch.empty(lib.conv_wgrad_workspace(n, d),dtype=torch.uint8,device="cuda")
                t = timeit(lambda: lib.conv_wgrad(x, stride, None, 0, dy, dw, db, n, d, ws), reps); res.append(f"wgrad {t*1e3:8.1f}us {flops/t/1e9:6.1f}TF" + (f" #{digest(dw)}" if HASH else ""))
            if "dgrad" in which and n == 32768 and not d.in_u8:
                din = torch.empty((n,d.H,d.W,d.Cin),device="cuda")
                t = timeit(lambda: lib.conv_dgrad(dy, w, x, din, n, d), reps); res.append(f"dgrad {t*1e3:8.1f}us {flops/t/1e9:6.1f}TF" + (f" #{digest(din)}" if HASH else ""))
            if "dgrad_noact" in which and n == 32768 and not d.in_u8:  # upper bound of what a mask-free epilogue could win
                din = torch.empty((n,d.H,d.W,d.Cin),device="cuda")
                t = timeit(lambda: lib.conv_dgrad(dy, w, None, din, n, d), reps); res.append(f"dgrad(no act read) {t*1e3:8.1f}us {flops/t/1e9:6.1f}TF" + (f" #{digest(din)}" if HASH else ""))
            print(f"n={n:6d} {name:6s} " + " | ".join(res), flush=True)

if __name__ == "__main__":
    main()

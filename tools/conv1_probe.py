"""conv1 (Nature-CNN, raw u8 frames) forward / weight gradient: accuracy against a float64 convolution and launch time,
for the kernel selected by SF_CONV1_BF16 (1: exact products on the bf16 pipe, 0: f32 MFMA).
   python tools/conv1_probe.py            # runs both settings in sub-processes
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one():
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from sample_factory_amd import lib
    lib.load()
    d = lib.sf_conv_desc(Cin=4, H=84, W=84, Cout=32, KH=8, KW=8, stride=4, OH=20, OW=20, in_u8=1, relu=1, traj_T=0,
                         sub_mean=float(os.environ.get("SUB", "0")), inv_scale=1 / 255.0)
    g = torch.Generator().manual_seed(0)
    K = 256
    w = (torch.randn((K, 32), generator=g) / 16).cuda()
    b = (torch.randn(32, generator=g) * 0.1).cuda()
    res = [f"SF_CONV1_BF16={os.environ.get('SF_CONV1_BF16', '1')} sub={d.sub_mean}"]
    for n in (4096, 32768):
        x = torch.randint(0, 256, (n, 4, 84, 84), dtype=torch.uint8, generator=g).cuda()
        out = torch.empty((n * 400, 32), device="cuda")
        name = lib.conv_kernel_name(0, n, d)
        fn = lambda: lib.conv_fwd(x, 4 * 84 * 84, None, 0, w, b, out, n, d, None)
        fn()
        torch.cuda.synchronize()
        # float64 reference on the first and last 64 samples: k = (c*KH + kh)*KW + kw  <->  OIHW weights
        w64 = w.double().t().reshape(32, 4, 8, 8)
        errs = []
        for sl in (slice(0, 64), slice(n - 64, n)):
            ref = torch.nn.functional.conv2d((x[sl].double() - d.sub_mean) / 255.0, w64, b.double(), stride=4).relu()
            got = out.view(n, 400, 32)[sl].double().permute(0, 2, 1).reshape(-1, 32, 20, 20)
            errs.append(float((got - ref).abs().max() / ref.abs().max()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20 if n == 4096 else 5
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * n * 400 * 32 * K
        res.append(f"n={n}: {name} {ms * 1e3:8.1f} us = {fl / ms / 1e9:6.1f} TFLOP/s-equivalent, "
                   f"{(n * 28224 + n * 400 * 128) / ms / 1e6:6.0f} GB/s algorithmic, max err vs f64 {max(errs):.2e}")
        # ---- weight / bias gradient of the same layer
        nw = 512 if n == 4096 else n
        dyv = (torch.randn((nw * 400, 32), generator=torch.Generator().manual_seed(n)) * 0.3).cuda()
        dw, db = torch.empty_like(w), torch.empty_like(b)
        ws = torch.empty(lib.conv_wgrad_workspace(nw, d), dtype=torch.uint8, device="cuda")
        wname = lib.conv_kernel_name(1, nw, d)
        gfn = lambda: lib.conv_wgrad(x[:nw], 4 * 84 * 84, None, 0, dyv, dw, db, nw, d, ws)
        gfn()
        torch.cuda.synchronize()
        ns = min(nw, 512)   # float64 reference on a 512-sample launch of its own
        dw2, db2 = torch.empty_like(w), torch.empty_like(b)
        lib.conv_wgrad(x[:ns], 4 * 84 * 84, None, 0, dyv[:ns * 400], dw2, db2, ns, d, ws)
        xin = ((x[:ns].double() - d.sub_mean) / 255.0).requires_grad_(False)
        w64r = w64.clone().requires_grad_(True)
        y = torch.nn.functional.conv2d(xin, w64r, None, stride=4)
        gy = dyv[:ns * 400].double().view(ns, 400, 32).permute(0, 2, 1).reshape(ns, 32, 20, 20)
        y.backward(gy)
        want = w64r.grad.reshape(32, 256).t()
        ew = float((dw2.double() - want).abs().max() / want.abs().max())
        eb = float((db2.double() - gy.sum((0, 2, 3))).abs().max() / gy.sum((0, 2, 3)).abs().max())
        e0.record()
        for _ in range(reps):
            gfn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        res.append(f"   wgrad n={nw}: {wname} {ms * 1e3:8.1f} us = {2.0 * nw * 400 * 32 * K / ms / 1e9:6.1f} TFLOP/s-equivalent, "
                   f"{(nw * 28224 + nw * 400 * 128) / ms / 1e6:6.0f} GB/s algorithmic, err vs f64 dW {ew:.2e} db {eb:.2e}")
    print("\n".join(res), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        for bf in ("1", "0"):
            for sub in ("0", "128"):
                subprocess.run([sys.executable, os.path.abspath(__file__), "one"],
                               env=dict(os.environ, SF_CONV1_BF16=bf, SUB=sub))

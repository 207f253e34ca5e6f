"""Network kernels at n = 32768 with HOT operands (the same buffers every launch, what tools/kbench.py times) against COLD
ones (rotating through enough copies of the input that nothing of it is left in L2 / Infinity Cache) — how much of the
in-step slowdown of a kernel against its stand-alone time is the memory system.   python tools/cold_probe.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sample_factory_amd import lib
from tools.kbench import desc, LAYERS


def bench(fn, nbuf, reps=12):
    for i in range(nbuf):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(reps):
        fn(i % nbuf)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    n = 32768
    for name, d in LAYERS[1:4]:
        K, M = d.KH * d.KW * d.Cin, n * d.OH * d.OW
        NB = 6
        xs = [torch.randn((n, d.H, d.W, d.Cin), device="cuda") for _ in range(NB)]
        w = torch.randn((K, d.Cout), device="cuda") / np.sqrt(K)
        b = torch.zeros(d.Cout, device="cuda")
        wt = w.t().contiguous()
        outs = [torch.empty((M, d.Cout), device="cuda") for _ in range(2)]
        stride = d.Cin * d.H * d.W
        nb = lib.conv_fwd_t_workspace(n, d)
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda") if nb else None
        f = lambda i: lib.conv_fwd_t(xs[i], stride, wt, b, outs[i % 2], n, d, ws)
        hot, cold = bench(lambda i: f(0), 1), bench(f, NB)
        # ... and right behind a kernel that has just WRITTEN the input (dirty lines in L2 / Infinity Cache, write-backs in
        # flight), as in the step where the previous layer produced it: events around the consumer only
        evs = []
        for i in range(12):
            xs[i % NB].mul_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(i % NB); e1.record()
            evs.append((e0, e1))
        torch.cuda.synchronize()
        fresh = sum(a.elapsed_time(b) for a, b in evs[2:]) / len(evs[2:]) * 1e3
        print(f"{name:6s} behind a producer of its input: {fresh:8.1f} us", flush=True)
        print(f"{name:6s} fwd_t  n={n}: hot {hot:8.1f} us   cold (rotating {NB} x {xs[0].numel() * 4 / 1e6:.0f} MB inputs) {cold:8.1f} us   {lib.conv_kernel_name(3, n, d)}", flush=True)


if __name__ == "__main__":
    main()

"""Does splitting a rollout step's inference into S env groups on S HIP streams fill the kernel tails?
python tools/split_probe.py  -> ms per 33-step chain for S = 1, 2, 4 (4096 envs total)"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sample_factory_amd import lib
from kbench import LAYERS


def build(n):
    bufs = []
    x = torch.randint(0, 256, (n, 4, 84, 84), dtype=torch.uint8, device="cuda")
    for name, d in LAYERS:
        K = d.KH * d.KW * d.Cin
        M = n * d.OH * d.OW
        w = torch.randn((K, d.Cout), device="cuda") / np.sqrt(K)
        b = torch.zeros(d.Cout, device="cuda")
        out = torch.empty((M, d.Cout), device="cuda")
        wt = w.t().contiguous()
        use_t = (not d.in_u8) and lib.conv_fwd_t_supported(n, d)
        nb = lib.conv_fwd_t_workspace(n, d) if use_t else lib.conv_fwd_workspace(n, d)
        ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda") if nb else None
        bufs.append((d, w, wt, b, out, ws, use_t))
    return x, bufs


def chain(n, x, bufs):
    inp, stride = x, 4 * 84 * 84
    for d, w, wt, b, out, ws, use_t in bufs:
        if use_t: lib.conv_fwd_t(inp, stride, wt, b, out, n, d, ws)
        else: lib.conv_fwd(inp, stride, None, 0, w, b, out, n, d, ws)
        inp, stride = out, d.OH * d.OW * d.Cout


def main():
    total, steps = 4096, 33
    for S in (1, 2, 4, 1, 2):
        n = total // S
        groups = [build(n) for _ in range(S)]
        streams = [torch.cuda.Stream() for _ in range(S)]
        def run():
            for _ in range(steps):
                for g in range(S):
                    with torch.cuda.stream(streams[g]):
                        chain(n, *groups[g])
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): run()
        torch.cuda.synchronize()
        print(f"S={S} n={n}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms per {steps}-step chain", flush=True)


if __name__ == "__main__":
    main()

"""Standalone timing of the fused LSTM sequence kernels at the config-5 minibatch shape (Cn = 512 chunks, R = 32,
H = 512), with the ablation bits of csrc/sf_rnn.hip (SF_LSTM_ABLATE) to see where a backward step's time goes.
   python tools/lstm_bench.py            # one process per ablation setting
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one():
    sys.path.insert(0, ROOT)
    import torch
    from sample_factory_amd import lib
    lib.load()
    R, Cn, H = 32, int(os.environ.get("LSTM_CN", "512")), 512
    g = torch.Generator().manual_seed(0)
    dev = lambda *s: torch.randn(s, generator=g).cuda() * 0.3
    gx, whh, bhh = dev(R, Cn, 4 * H), dev(H, 4 * H) / 20, dev(4 * H)
    keep = (torch.rand((R, Cn), generator=g) > 0.05).float().cuda()
    gates, hout, cout = (torch.empty(s, device="cuda") for s in [(R, Cn, 4 * H), (R, Cn, H), (R, Cn, H)])
    hprev, cprev = torch.zeros((R + 1, Cn, H), device="cuda"), torch.zeros((R + 1, Cn, H), device="cuda")
    dout, dgx = dev(R, Cn, H), torch.empty((R, Cn, 4 * H), device="cuda")
    sync = torch.zeros(192, dtype=torch.int32, device="cuda")
    fwd = lambda: lib.lstm_seq_fwd(gx, whh, bhh, keep, gates, hprev, hout, cprev, cout, sync, R, Cn, H)
    bwd = lambda: lib.lstm_seq_bwd(dout, gates, cprev, cout, keep, whh, dgx, sync, R, Cn, H)
    if os.environ.get("RNN_KIND", "lstm") == "gru":
        gx3, whh3, dgx3, dgh3 = gx[..., :3 * H].contiguous(), whh[:, :3 * H].contiguous(), torch.empty((R, Cn, 3 * H), device="cuda"), torch.empty((R, Cn, 3 * H), device="cuda")
        fwd = lambda: lib.gru_seq_fwd(gx3, whh3, bhh, keep, gates, hprev, hout, sync, R, Cn, H)
        bwd = lambda: lib.gru_seq_bwd(dout, gates, hprev, keep, whh3, dgx3, dgh3, sync, R, Cn, H)
    out = []
    for name, fn in [("fwd", fwd), ("bwd", bwd)]:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        out.append(f"{name} {ms * 1e3:7.1f} us = {ms * 1e3 / R:5.1f} us/step")
    print(f"{os.environ.get('RNN_KIND', 'lstm')} ablate={os.environ.get('SF_LSTM_ABLATE', '0'):>2}  " + "   ".join(out) + f"   aborted={int(sync[128])}")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        for ab in ["0", "1", "2", "4", "6", "8", "9", "15"]:
            subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, SF_LSTM_ABLATE=ab))

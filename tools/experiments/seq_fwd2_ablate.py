"""forward LSTM sequence pass at the configs[4] minibatch shape: k_lstm_seq_fwd (SF_SEQ_FWD2=0) against the PARKED
k_lstm_seq_fwd2 (tools/experiments/sf_rnn_fwd2.h — paste it into csrc/sf_rnn.hip and dispatch it as its header says; the
product library has neither the kernel nor the SF_SEQ_FWD2 switch) with its ablation bits (SF_LSTM_ABLATE: 1 no wait, 2 no h
loads, 4 no MFMAs, 8 no saves, 16 no arrivals, 32 no matrix-phase token).  Results: profiles/r04_d_seq_fwd2_ablate.log"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def one():
    sys.path.insert(0, ROOT)
    import torch
    from sample_factory_amd import lib
    lib.load()
    R, Cn, H, Kx = 32, 512, 512, 64
    g = torch.Generator().manual_seed(0)
    dev = lambda *s: torch.randn(s, generator=g).cuda() * 0.3
    x, wih, bih, whh, bhh = dev(R, Cn, Kx), dev(4 * H, Kx), dev(4 * H), dev(H, 4 * H) / 20, dev(4 * H)
    keep = (torch.rand((R, Cn), generator=g) > 0.05).float().cuda()
    gates, hout, cout = (torch.empty(s, device="cuda") for s in [(R, Cn, 4 * H), (R, Cn, H), (R, Cn, H)])
    hprev, cprev = torch.zeros((R + 1, Cn, H), device="cuda"), torch.zeros((R + 1, Cn, H), device="cuda")
    sync = torch.zeros(192, dtype=torch.int32, device="cuda")
    fn = lambda: lib.lstm_seq_fwd_x(x, wih, bih, whh, bhh, keep, gates, hprev, hout, cprev, cout, sync, R, Cn, H)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"fwd2={os.environ.get('SF_SEQ_FWD2','1')} ablate={os.environ.get('SF_LSTM_ABLATE','0'):>2}  {ms*1e3:7.1f} us = {ms*1e3/R:5.1f} us/step  aborted={int(sync[128])}", flush=True)
if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one()
    else:
        runs = [("0", "0")] + [("1", a) for a in sys.argv[1:] or ["0", "1", "2", "4", "8", "16", "17", "6", "7", "15", "31"]]
        for f2, ab in runs:
            subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=dict(os.environ, SF_SEQ_FWD2=f2, SF_LSTM_ABLATE=ab))

"""Debug helper: register-weight backward pass against the LDS-weight one on the same inputs (two processes)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def one(out):
    sys.path.insert(0, ROOT)
    import torch
    from sample_factory_amd import lib
    R, Cn, H = int(os.environ.get("R", 2)), int(os.environ.get("CN", 32)), int(os.environ.get("H", 512))
    g = torch.Generator().manual_seed(0)
    dev = lambda *s: (torch.randn(s, generator=g) * 0.3).cuda()
    gx, whh, bhh = dev(R, Cn, 4 * H), dev(H, 4 * H) / 20, dev(4 * H)
    keep = torch.ones((R, Cn)).cuda()
    gates, hout, cout = (torch.empty(s, device="cuda") for s in [(R, Cn, 4 * H), (R, Cn, H), (R, Cn, H)])
    hprev, cprev = torch.zeros((R + 1, Cn, H), device="cuda"), torch.zeros((R + 1, Cn, H), device="cuda")
    dout, dgx = dev(R, Cn, H), torch.zeros((R, Cn, 4 * H), device="cuda")
    if os.environ.get("ONLY_LAST"):
        dout[:-1] = 0
    sync = torch.zeros(192, dtype=torch.int32, device="cuda")
    lib.lstm_seq_fwd(gx, whh, bhh, keep, gates, hprev, hout, cprev, cout, sync, R, Cn, H)
    lib.lstm_seq_bwd(dout, gates, cprev, cout, keep, whh, dgx, sync, R, Cn, H)
    torch.cuda.synchronize()
    torch.save(dgx.cpu(), out)

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "one":
        one(sys.argv[2])
    else:
        import torch
        for tag, v in (("new", "1"), ("old", "0")):
            subprocess.run([sys.executable, os.path.abspath(__file__), "one", f"/tmp/dgx_{tag}.pt"], env=dict(os.environ, SF_SEQ_BWD_REGW=v), check=True)
        a, b = torch.load("/tmp/dgx_new.pt"), torch.load("/tmp/dgx_old.pt")
        R, Cn, G4 = a.shape
        H = G4 // 4
        for t in range(R):
            e = (a[t] - b[t]).abs()
            print(f"t={t} max err {e.max():.3e} of {b[t].abs().max():.3e}")
        e = (a[0] - b[0]).abs().view(Cn, 4, H)
        bad_rows = (e.amax(dim=(1, 2)) > 1e-5).nonzero().flatten().tolist()
        bad_units = (e.amax(dim=(0, 1)) > 1e-5).nonzero().flatten().tolist()
        print("bad rows", bad_rows[:70], len(bad_rows))
        print("bad units", bad_units[:140], len(bad_units))

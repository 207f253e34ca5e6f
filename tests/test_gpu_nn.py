"""GPU parity tests for the fp32-MFMA implicit-GEMM network kernels (sf_conv_fwd / wgrad / dgrad and the linear
wrappers) and for the whole model/learner against golden vectors produced by the reference's own ActorCritic /
Learner.train.  Floating-point kernels: the checker is plain torch fp32/fp64 math on the CPU (same op, different
implementation); tolerances are fp32-accumulation class and written next to each assert."""
import os
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle.weights import seeded_state
from tests.parity_util import compare_post_train

# the Learner.train replays against the reference: Adam first / second moments, weight DELTAS and gradient norms
# (tests/parity_util.py).  On these small batches both fp32 implementations sit within 4e-6 (m), 1.5e-5 (v) and 7e-4
# (delta) of each other relative to the tensor's largest element: relative tolerances 1e-4 / 2e-4 / 1e-3 with an absolute
# floor of 1e-4 (m, v) resp. 2.5e-3 (delta) of that largest element — 20x tighter than the benchmark-geometry replays of
# tests/test_gpu_parity_c2_c5.py, whose 32768-sample sums and ReLU mask flips are analysed in DESIGN.md 8.1
TIGHT = dict(m_rtol=1e-4, v_rtol=2e-4, d_rtol=1e-3, gn_rtol=5e-4, floor=1e-4)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from sample_factory_amd import lib as L
    L.load()
    return L


def desc(lib, Cin, H, W, Cout, K, S, in_u8=0, relu=1, sub_mean=0.0, inv_scale=1.0, traj_T=0):
    return lib.sf_conv_desc(Cin=Cin, H=H, W=W, Cout=Cout, KH=K, KW=K, stride=S, OH=(H - K) // S + 1, OW=(W - K) // S + 1,
                            in_u8=in_u8, relu=relu, traj_T=traj_T, sub_mean=sub_mean, inv_scale=inv_scale)


def to_kmajor(w_ref, u8):
    O, C, KH, KW = w_ref.shape
    if u8:
        return w_ref.reshape(O, -1).t().contiguous()
    return w_ref.permute(2, 3, 1, 0).reshape(KH * KW * C, O).contiguous()


def from_kmajor(w, O, C, KH, KW, u8):
    if u8:
        return w.t().reshape(O, C, KH, KW)
    return w.reshape(KH, KW, C, O).permute(3, 2, 0, 1)


GEOMS = [  # (Cin, H, W, Cout, K, S, u8)
    (4, 84, 84, 32, 8, 4, 1),    # Nature conv1 on raw u8 frames
    (32, 20, 20, 64, 4, 2, 0),   # conv2
    (64, 9, 9, 64, 3, 1, 0),     # conv3
    (4, 36, 36, 32, 8, 4, 1),    # the golden cnn36 model's conv1
    (16, 11, 13, 48, 3, 2, 0),   # odd sizes, Cout not a multiple of 32/64, H % stride != 0
    (8, 6, 6, 128, 3, 1, 0),     # convnet_simple-like wide layer
]


@pytest.mark.parametrize("geom", GEOMS)
@pytest.mark.parametrize("n", [3, 37])
def test_conv_fwd_wgrad_dgrad_vs_torch(lib, geom, n):
    Cin, H, W, Cout, K, S, u8 = geom
    g = torch.Generator().manual_seed(Cin * 131 + n)
    if u8:
        x = torch.randint(0, 256, (n, Cin, H, W), generator=g, dtype=torch.uint8)
        xf = (x.double() - 3.0) * float(np.float32(1 / 255.0))
        d = desc(lib, Cin, H, W, Cout, K, S, in_u8=1, sub_mean=3.0, inv_scale=float(np.float32(1 / 255.0)))
        x_dev = x.cuda()
        stride = Cin * H * W
    else:
        x = torch.randn((n, Cin, H, W), generator=g)
        xf = x.double()
        d = desc(lib, Cin, H, W, Cout, K, S)
        x_dev = x.permute(0, 2, 3, 1).contiguous().cuda()  # NHWC
        stride = Cin * H * W
    w_ref = torch.randn((Cout, Cin, K, K), generator=g) / np.sqrt(Cin * K * K)
    b = torch.randn(Cout, generator=g) * 0.1
    wk = to_kmajor(w_ref, u8).cuda()
    OH, OW = d.OH, d.OW
    out = torch.empty((n * OH * OW, Cout), device="cuda")
    lib.conv_fwd(x_dev, stride, None, 0, wk, b.cuda(), out, n, d)
    ref = F.relu(F.conv2d(xf, w_ref.double(), b.double(), stride=S))           # [n, Cout, OH, OW] fp64
    got = out.view(n, OH, OW, Cout).permute(0, 3, 1, 2).cpu().double()
    scale = ref.abs().max().item() + 1e-6
    assert (got - ref).abs().max().item() < 2e-5 * max(1.0, scale), "forward"   # fp32 accumulation over K<=576

    # weight / bias gradient for an upstream gradient dY (already ReLU-masked by the caller)
    dy = torch.randn((n, Cout, OH, OW), generator=g)
    dy_dev = dy.permute(0, 2, 3, 1).contiguous().cuda().view(n * OH * OW, Cout)
    dw = torch.zeros_like(wk)
    db = torch.zeros(Cout, device="cuda")
    ws = torch.empty(lib.conv_wgrad_workspace(n, d), dtype=torch.uint8, device="cuda")
    lib.conv_wgrad(x_dev, stride, None, 0, dy_dev, dw, db, n, d, ws)
    xr = xf.clone().requires_grad_(True)
    wr = w_ref.double().clone().requires_grad_(True)
    br = b.double().clone().requires_grad_(True)
    F.conv2d(xr, wr, br, stride=S).backward(dy.double())
    dw_got = from_kmajor(dw.cpu().double(), Cout, Cin, K, K, u8)
    s = wr.grad.abs().max().item() + 1e-6
    assert (dw_got - wr.grad).abs().max().item() < 3e-5 * max(1.0, s), "wgrad"
    assert (db.cpu().double() - br.grad).abs().max().item() < 3e-5 * max(1.0, br.grad.abs().max().item()), "bgrad"

    if not u8:  # data gradient (+ fused ReLU mask of the producer layer's activation)
        act = x_dev  # pretend x is the producer's post-ReLU output: mask = x > 0
        din = torch.full((n, H, W, Cin), 7.0, device="cuda")
        lib.conv_dgrad(dy_dev, wk, act, din, n, d)
        dref = (xr.grad * (xf > 0)).permute(0, 2, 3, 1)
        s = dref.abs().max().item() + 1e-6
        assert (din.cpu().double() - dref).abs().max().item() < 3e-5 * max(1.0, s), "dgrad"
        din2 = torch.empty((n, H, W, Cin), device="cuda")
        lib.conv_dgrad(dy_dev, wk, None, din2, n, d)
        assert (din2.cpu().double() - xr.grad.permute(0, 2, 3, 1)).abs().max().item() < 3e-5 * max(1.0, s), "dgrad nomask"


def fwd_t_ws(lib, n, d):
    nb = lib.conv_fwd_t_workspace(n, d)
    return torch.empty(nb, dtype=torch.uint8, device="cuda") if nb else None


@pytest.mark.parametrize("geom,n,act", [((3136, 1, 1, 512, 1, 1), 2048, 1), ((1024, 1, 1, 200, 1, 1), 4001, 2),
                                        ((64, 5, 5, 136, 5, 1), 4500, 1), ((1056, 1, 1, 128, 1, 1), 8200, 0)])
def test_glds_fwd_splitk_small_grids_vs_torch(lib, geom, n, act):
    """wide layer, long reduction, too few rows to fill the chip: sf_conv_fwd_t splits along K into workspace slices +
    k_splitk_finish.  Ragged row tile, Cout not a multiple of 128, K slices of unequal length, ReLU / tanh / no activation
    in the finisher; missing workspace must fail loudly.  (The fc layer of a 4096-env rollout step has >= 512 tiles of
    64x64 and runs unsplit: next test.)"""
    Cin, H, W, Cout, K, S = geom
    g = torch.Generator().manual_seed(Cin + n)
    x = torch.randn((n, Cin, H, W), generator=g)
    d = desc(lib, Cin, H, W, Cout, K, S)
    d.relu = act
    x_dev = x.permute(0, 2, 3, 1).contiguous().cuda()
    w_ref = torch.randn((Cout, Cin, K, K), generator=g) / np.sqrt(Cin * K * K)
    b = torch.randn(Cout, generator=g) * 0.1
    wk = to_kmajor(w_ref, 0).cuda()
    assert lib.conv_fwd_t_supported(n, d)
    nb = lib.conv_fwd_t_workspace(n, d)
    assert nb > 0, "this geometry is meant to take the split-K plan"
    wt = wk.t().contiguous()
    out = torch.full((n * d.OH * d.OW, Cout), 7.0, device="cuda")
    with pytest.raises(lib.SfHipError):
        lib.conv_fwd_t(x_dev, Cin * H * W, wt, b.cuda(), out, n, d, None)
    lib.conv_fwd_t(x_dev, Cin * H * W, wt, b.cuda(), out, n, d, fwd_t_ws(lib, n, d))
    pre = F.conv2d(x, w_ref, b, stride=S)
    ref = F.relu(pre) if act == 1 else torch.tanh(pre) if act == 2 else pre
    got = out.view(n, d.OH, d.OW, Cout).permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
    out2 = torch.empty_like(out)
    lib.conv_fwd_t(x_dev, Cin * H * W, wt, b.cuda(), out2, n, d, fwd_t_ws(lib, n, d))
    assert torch.equal(out, out2), "slices are added in a fixed order: bit-reproducible"


@pytest.mark.parametrize("n,cout,act", [(4096, 512, 1), (4100, 520, 2), (5000, 448, 0)])
def test_glds_fwd_64x64_unsplit_fc_of_a_rollout_step_vs_torch(lib, n, cout, act):
    """the fc layer of a rollout step (4096 x 3136 -> 512): 512 tiles of 64x64 = two work-groups per CU, whole reduction
    in one pass on k_fwd_glds<64, 64> — no workspace, no partial sums; ragged row / column tiles"""
    Cin = 3136
    g = torch.Generator().manual_seed(n + cout)
    x = torch.randn((n, Cin), generator=g)
    d = desc(lib, Cin, 1, 1, cout, 1, 1)
    d.relu = act
    w_ref = torch.randn((cout, Cin), generator=g) / np.sqrt(Cin)
    b = torch.randn(cout, generator=g) * 0.1
    assert lib.conv_fwd_t_supported(n, d)
    assert lib.conv_kernel_name(3, n, d) in ("k_fwd_glds<64, 64, 2, 2, 2>", "k_fwd_glds_z<64, 64, 2, 2>"), lib.conv_kernel_name(3, n, d)
    assert lib.conv_fwd_t_workspace(n, d) == 0
    out = torch.full((n, cout), 7.0, device="cuda")
    lib.conv_fwd_t(x.cuda(), Cin, w_ref.cuda().contiguous(), b.cuda(), out, n, d, None)
    pre = x.double() @ w_ref.double().t() + b.double()
    ref = torch.relu(pre) if act == 1 else torch.tanh(pre) if act == 2 else pre
    assert (out.cpu().double() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
    out2 = torch.empty_like(out)
    lib.conv_fwd_t(x.cuda(), Cin, w_ref.cuda().contiguous(), b.cuda(), out2, n, d, None)
    assert torch.equal(out, out2)


@pytest.mark.parametrize("n", [512, 513, 1024, 2048])
def test_glds_fwd_small_inference_launches_vs_torch(lib, n):
    """the per-split inference launches of a host-env run (n = 512 .. 2048 samples; BASELINE configs[2]: 1024 envs in two
    splits): conv2 on 64 x 64 LDS-DMA tiles, the fc layer on 64 x 64 tiles split along K with deterministic slice order —
    both used to fall to the register-staged kernels (profiles/r05_b_kbench_small_n.log: fc at n = 512 170 -> 26 us)"""
    import torch.nn.functional as F
    default = not any(os.environ.get(k) for k in ("SF_GLDS_SMALL64", "SF_GLDS_SPLIT64", "SF_GLDS_MIN_TILES", "SF_GLDS_CFG",
                                                  "SF_GLDS_SPLITK", "SF_GLDS_ZL"))
    g = torch.Generator().manual_seed(n)
    # conv2: 32 x 20 x 20 -> 64, 4 x 4 stride 2
    d = desc(lib, 32, 20, 20, 64, 4, 2)
    x = torch.randn((n, 32, 20, 20), generator=g)
    w_ref = torch.randn((64, 32, 4, 4), generator=g) / np.sqrt(512)
    b = torch.randn(64, generator=g) * 0.1
    if not default and not lib.conv_fwd_t_supported(n, d):
        pytest.skip("the switch under test sends this launch back to the register-staged kernels")
    assert lib.conv_fwd_t_supported(n, d)
    if default and n <= 1024:
        assert lib.conv_kernel_name(3, n, d) in ("k_fwd_glds<64, 64, 2, 2, 2>", "k_fwd_glds_z<64, 64, 2, 2>"), lib.conv_kernel_name(3, n, d)
    wk = to_kmajor(w_ref, 0).cuda()
    wt = torch.empty((64, 512), device="cuda")
    lib.transpose(wk, wt, 512, 64)
    out = torch.full((n * 81, 64), 7.0, device="cuda")
    nb = lib.conv_fwd_t_workspace(n, d)
    ws = torch.empty(nb, dtype=torch.uint8, device="cuda") if nb else None
    lib.conv_fwd_t(x.permute(0, 2, 3, 1).contiguous().cuda(), 32 * 400, wt, b.cuda(), out, n, d, ws)
    ref = F.relu(F.conv2d(x.double(), w_ref.double(), b.double(), stride=2)).permute(0, 2, 3, 1).reshape(n * 81, 64)
    assert (out.cpu().double() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
    # fc: 3136 -> 512
    Cin, cout = 3136, 512
    d = desc(lib, Cin, 1, 1, cout, 1, 1)
    x = torch.randn((n, Cin), generator=g)
    w_ref = torch.randn((cout, Cin), generator=g) / np.sqrt(Cin)
    b = torch.randn(cout, generator=g) * 0.1
    if not default and not lib.conv_fwd_t_supported(n, d):
        return
    assert lib.conv_fwd_t_supported(n, d)
    nb = lib.conv_fwd_t_workspace(n, d)
    if default:
        assert lib.conv_kernel_name(3, n, d) in ("k_fwd_glds<64, 64, 2, 2, 2>", "k_fwd_glds_z<64, 64, 2, 2>") and nb > 0, (lib.conv_kernel_name(3, n, d), nb)
    ws = torch.empty(max(nb, 16), dtype=torch.uint8, device="cuda")
    out = torch.full((n, cout), 7.0, device="cuda")
    lib.conv_fwd_t(x.cuda(), Cin, w_ref.cuda().contiguous(), b.cuda(), out, n, d, ws if nb else None)
    ref = torch.relu(x.double() @ w_ref.double().t() + b.double())
    assert (out.cpu().double() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
    out2 = torch.empty_like(out)
    lib.conv_fwd_t(x.cuda(), Cin, w_ref.cuda().contiguous(), b.cuda(), out2, n, d, ws if nb else None)
    assert torch.equal(out, out2)       # split-K slices are added in a fixed order


@pytest.mark.parametrize("geom,n", [((32, 20, 20, 64, 4, 2), 2500), ((64, 9, 9, 64, 3, 1), 2500),
                                    ((64, 9, 9, 64, 3, 1), 2049),    # odd n: the image kernels' last pair / block has one sample
                                    ((32, 20, 20, 64, 4, 2), 1537),
                                    ((32, 20, 20, 64, 4, 2), 4096),  # conv2 of a rollout step: tail-split grid (k_fwd_glds)
                                    ((32, 11, 13, 96, 3, 2), 4100), ((96, 1, 1, 160, 1, 1), 70001),
                                    ((32, 12, 14, 64, 3, 2), 4300),  # input row/col no filter tap reaches
                                    ((384, 1, 1, 160, 1, 1), 33001)])  # linear layer: dgrad = masked forward GEMM
def test_glds_fwd_dgrad_large_grids_vs_torch(lib, geom, n):
    """the gfx950 LDS-DMA kernels (sf_nn_glds.h) only take launches that fill the chip: forward through
    sf_conv_fwd_t (transposed weights), data gradient through sf_conv_dgrad's internal dispatch.  Odd sizes, boundary
    taps (zero page), Cout not a multiple of the column tile, ragged last row tile."""
    Cin, H, W, Cout, K, S = geom
    g = torch.Generator().manual_seed(Cin * 7 + n)
    x = torch.randn((n, Cin, H, W), generator=g)
    d = desc(lib, Cin, H, W, Cout, K, S)
    x_dev = x.permute(0, 2, 3, 1).contiguous().cuda()
    w_ref = torch.randn((Cout, Cin, K, K), generator=g) / np.sqrt(Cin * K * K)
    b = torch.randn(Cout, generator=g) * 0.1
    wk = to_kmajor(w_ref, 0).cuda()
    OH, OW = d.OH, d.OW
    assert lib.conv_fwd_t_supported(n, d)
    wt = torch.empty((Cout, K * K * Cin), device="cuda")
    lib.transpose(wk, wt, K * K * Cin, Cout)
    assert torch.equal(wt, wk.t().contiguous())
    out = torch.empty((n * OH * OW, Cout), device="cuda")
    lib.conv_fwd_t(x_dev, Cin * H * W, wt, b.cuda(), out, n, d, fwd_t_ws(lib, n, d))
    xr = x.clone().requires_grad_(True)
    wr = w_ref.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    pre = F.conv2d(xr, wr, br, stride=S)
    ref = F.relu(pre).detach()
    got = out.view(n, OH, OW, Cout).permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item()), "forward"
    out_old = torch.empty_like(out)
    lib.conv_fwd(x_dev, Cin * H * W, None, 0, wk, b.cuda(), out_old, n, d)
    assert (out - out_old).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())

    dy = torch.randn((n, Cout, OH, OW), generator=g)
    dy_dev = dy.permute(0, 2, 3, 1).contiguous().cuda().view(n * OH * OW, Cout)
    pre.backward(dy)
    din = torch.full((n, H, W, Cin), 7.0, device="cuda")
    lib.conv_dgrad(dy_dev, wk, x_dev, din, n, d)
    dref = (xr.grad * (x > 0)).permute(0, 2, 3, 1)
    s_ = dref.abs().max().item() + 1e-6
    assert (din.cpu() - dref).abs().max().item() < 3e-5 * max(1.0, s_), "dgrad"
    lib.conv_dgrad(dy_dev, wk, None, din, n, d)
    assert (din.cpu() - xr.grad.permute(0, 2, 3, 1)).abs().max().item() < 3e-5 * max(1.0, s_), "dgrad nomask"
    # weight / bias gradient: reduction over n*OH*OW >= 65536 rows -> LDS-DMA kernel; fp32 sums of ~1e5 terms
    dw = torch.zeros_like(wk)
    db = torch.zeros(Cout, device="cuda")
    ws = torch.empty(lib.conv_wgrad_workspace(n, d), dtype=torch.uint8, device="cuda")
    lib.conv_wgrad(x_dev, Cin * H * W, None, 0, dy_dev, dw, db, n, d, ws)
    dw_got = from_kmajor(dw.cpu(), Cout, Cin, K, K, 0)
    sw = wr.grad.abs().max().item() + 1e-6
    assert (dw_got - wr.grad).abs().max().item() < 2e-4 * max(1.0, sw), "wgrad"
    assert (db.cpu() - br.grad).abs().max().item() < 2e-4 * max(1.0, br.grad.abs().max().item()), "bgrad"


def test_glds_kernels_geometry_fuzz(lib):
    """random conv geometries through the large-grid dispatch of forward (sf_conv_fwd_t), weight gradient and data
    gradient (pixel-major for stride 1 / odd kernels, stride-group row-walking when KH, KW, W are multiples of S) vs
    torch; sizes chosen so that every LDS-DMA kernel is actually selected."""
    rng = np.random.default_rng(int(__import__('os').environ.get('SF_FUZZ_SEED', '2024')))
    cases = 0
    while cases < int(__import__('os').environ.get('SF_FUZZ_CASES', '14')):
        Cin, Cout = int(rng.choice([32, 64, 96])), int(rng.choice([32, 64, 96, 160]))
        K, S = int(rng.integers(1, 6)), int(rng.integers(1, 4))
        H, W = int(rng.integers(K, K + 9)), int(rng.integers(K, K + 9))
        if cases % 3 == 0 and S > 1:  # make some cases hit the stride-group kernel
            K, W = S * int(rng.integers(1, 3)), S * int(rng.integers(2, 6))
            H = max(H, K)
            W = max(W, K)
        OH, OW = (H - K) // S + 1, (W - K) // S + 1
        if OH < 1 or OW < 1:
            continue
        n = int(max(1100, -(-140000 // (OH * OW))))
        if n * H * W * Cin > 6e7:
            continue
        cases += 1
        g = torch.Generator().manual_seed(cases)
        x = torch.randn((n, Cin, H, W), generator=g)
        d = desc(lib, Cin, H, W, Cout, K, S)
        x_dev = x.permute(0, 2, 3, 1).contiguous().cuda()
        w_ref = torch.randn((Cout, Cin, K, K), generator=g) / np.sqrt(Cin * K * K)
        b = torch.randn(Cout, generator=g) * 0.1
        wk = to_kmajor(w_ref, 0).cuda()
        tag = f"Cin={Cin} H={H} W={W} Cout={Cout} K={K} S={S} n={n}"
        print("fuzz case", cases, tag, flush=True)
        xr, wr, br = x.clone().requires_grad_(True), w_ref.clone().requires_grad_(True), b.clone().requires_grad_(True)
        pre = F.conv2d(xr, wr, br, stride=S)
        out = torch.empty((n * OH * OW, Cout), device="cuda")
        if lib.conv_fwd_t_supported(n, d):
            wt = torch.empty((Cout, K * K * Cin), device="cuda")
            lib.transpose(wk, wt, K * K * Cin, Cout)
            lib.conv_fwd_t(x_dev, Cin * H * W, wt, b.cuda(), out, n, d, fwd_t_ws(lib, n, d))
        else:
            lib.conv_fwd(x_dev, Cin * H * W, None, 0, wk, b.cuda(), out, n, d)
        ref = F.relu(pre).detach()
        got = out.view(n, OH, OW, Cout).permute(0, 3, 1, 2).cpu()
        assert (got - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item()), "fwd " + tag
        dy = torch.randn((n, Cout, OH, OW), generator=g)
        dy_dev = dy.permute(0, 2, 3, 1).contiguous().cuda().view(n * OH * OW, Cout)
        pre.backward(dy)
        din = torch.full((n, H, W, Cin), 7.0, device="cuda")
        lib.conv_dgrad(dy_dev, wk, x_dev, din, n, d)
        dref = (xr.grad * (x > 0)).permute(0, 2, 3, 1)
        s_ = dref.abs().max().item() + 1e-6
        assert (din.cpu() - dref).abs().max().item() < 3e-5 * max(1.0, s_), "dgrad " + tag
        dw, db = torch.zeros_like(wk), torch.zeros(Cout, device="cuda")
        ws = torch.empty(lib.conv_wgrad_workspace(n, d), dtype=torch.uint8, device="cuda")
        lib.conv_wgrad(x_dev, Cin * H * W, None, 0, dy_dev, dw, db, n, d, ws)
        dw_got = from_kmajor(dw.cpu(), Cout, Cin, K, K, 0)
        sw = wr.grad.abs().max().item() + 1e-6
        assert (dw_got - wr.grad).abs().max().item() < 3e-4 * max(1.0, sw), "wgrad " + tag
        assert (db.cpu() - br.grad).abs().max().item() < 3e-4 * max(1.0, br.grad.abs().max().item()), "bgrad " + tag


@pytest.mark.parametrize("geom,n,mean", [((4, 84, 84, 32, 8, 4), 515, 3.0), ((4, 84, 84, 32, 8, 4), 300, 0.0),
                                         ((4, 36, 36, 32, 8, 4), 1001, 0.0), ((4, 84, 84, 24, 8, 4), 258, 1.5)])
def test_conv1_lds_image_kernel_vs_torch(lib, geom, n, mean):
    """raw-u8 first layer through the LDS-image kernel (n >= 256): odd sample counts (ragged last block), index
    gather, Cout < 32, with and without mean subtraction; must also agree with the im2col kernel."""
    Cin, H, W, Cout, K, S = geom
    g = torch.Generator().manual_seed(n)
    x = torch.randint(0, 256, (n + 7, Cin, H, W), generator=g, dtype=torch.uint8)
    inv = float(np.float32(1 / 255.0))
    d = desc(lib, Cin, H, W, Cout, K, S, in_u8=1, sub_mean=mean, inv_scale=inv)
    w_ref = torch.randn((Cout, Cin, K, K), generator=g) / np.sqrt(Cin * K * K)
    b = torch.randn(Cout, generator=g) * 0.1
    wk = to_kmajor(w_ref, 1).cuda()
    idx = torch.randperm(n + 7, generator=g)[:n].to(torch.int32)
    x_dev = x.cuda()
    out = torch.empty((n * d.OH * d.OW, Cout), device="cuda")
    lib.conv_fwd(x_dev, Cin * H * W, idx.cuda(), 0, wk, b.cuda(), out, n, d)
    ref = F.relu(F.conv2d((x[idx.long()].float() - mean) * inv, w_ref, b, stride=S))
    got = out.view(n, d.OH, d.OW, Cout).permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    out2 = torch.empty_like(out)   # dense (offset) addressing of the same samples, small batches -> im2col kernel
    xs = x_dev[idx.long().cuda()].contiguous()
    for i in range(0, n, 100):
        m = min(100, n - i)
        lib.conv_fwd(xs, Cin * H * W, None, i, wk, b.cuda(), out2[i * d.OH * d.OW:], m, d)
    assert (out - out2).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item())
    # weight / bias gradient of the same layer (strip-image kernel for the 84x84 geometry with 32 filters)
    dy = torch.randn((n, Cout, d.OH, d.OW), generator=g)
    dy_dev = dy.permute(0, 2, 3, 1).contiguous().cuda().view(n * d.OH * d.OW, Cout)
    dw = torch.zeros_like(wk)
    db = torch.zeros(Cout, device="cuda")
    ws = torch.empty(lib.conv_wgrad_workspace(n, d), dtype=torch.uint8, device="cuda")
    lib.conv_wgrad(x_dev, Cin * H * W, idx.cuda(), 0, dy_dev, dw, db, n, d, ws)
    wr = w_ref.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    F.conv2d((x[idx.long()].double() - mean) * inv, wr.double(), br.double(), stride=S).backward(dy.double())
    dw_got = from_kmajor(dw.cpu().double(), Cout, Cin, K, K, 1)
    sw = wr.grad.abs().max().item()
    assert (dw_got - wr.grad.double()).abs().max().item() < 3e-5 * max(1.0, sw), "wgrad"
    assert (db.cpu().double() - br.grad.double()).abs().max().item() < 3e-5 * max(1.0, br.grad.abs().max().item())


def test_conv1_exact_product_kernels_on_integers(lib):
    """csrc/sf_nn_u8.h: u8 pixels are bf16 numbers and f32 weights / output gradients are split EXACTLY into three bf16
    terms, so every product is exact and only the f32 accumulation rounds.  With integer-valued weights (|w| <= 64, 1/scale
    = 1, integer mean) every partial sum is an integer below 2^24: forward output, weight gradient and bias gradient must
    then equal the integer convolution EXACTLY; with full-mantissa weights the error against float64 stays at the
    accumulation level (a few 1e-7 of the largest value)."""
    n, Cin, H, W, Cout, K, S = 260, 4, 84, 84, 32, 8, 4
    g = torch.Generator().manual_seed(5)
    x = torch.randint(0, 256, (n, Cin, H, W), generator=g, dtype=torch.uint8)
    d = desc(lib, Cin, H, W, Cout, K, S, in_u8=1, sub_mean=128.0, inv_scale=1.0, relu=0)
    assert lib.conv_kernel_name(0, n, d).startswith("k_conv1_u8_bf16") and lib.conv_kernel_name(1, n, d).startswith("k_conv1_wgrad_bf16")
    w_int = torch.randint(-64, 65, (Cout, Cin, K, K), generator=g).float()
    b_int = torch.randint(-100, 101, (Cout,), generator=g).float()
    out = torch.empty((n * 400, Cout), device="cuda")
    lib.conv_fwd(x.cuda(), Cin * H * W, None, 0, to_kmajor(w_int, 1).cuda(), b_int.cuda(), out, n, d)
    ref = F.conv2d(x.double() - 128.0, w_int.double(), b_int.double(), stride=S)     # |sum| <= 256*128*64 < 2^24: exact in f32
    got = out.view(n, 20, 20, Cout).permute(0, 3, 1, 2).cpu().double()
    assert torch.equal(got, ref)
    dy_int = torch.randint(-3, 4, (n, Cout, 20, 20), generator=g).float()          # |dW| <= 260*400*128*3 < 2^27: sums of
    dyd = dy_int.permute(0, 2, 3, 1).contiguous().cuda().view(n * 400, Cout)       # integers, exact while below 2^24 -> use 64 samples
    m = 64
    dw, db = torch.zeros((Cin * K * K, Cout), device="cuda"), torch.zeros(Cout, device="cuda")
    ws = torch.empty(lib.conv_wgrad_workspace(n, d), dtype=torch.uint8, device="cuda")
    xs = torch.cat([x[:m], torch.zeros((n - m, Cin, H, W), dtype=torch.uint8) + 128]).cuda()   # x - mean = 0 beyond the first m samples
    lib.conv_wgrad(xs, Cin * H * W, None, 0, dyd, dw, db, n, d, ws)
    wr = torch.zeros((Cout, Cin, K, K), dtype=torch.float64, requires_grad=True)
    F.conv2d(x[:m].double() - 128.0, wr, None, stride=S).backward(dy_int[:m].double())
    assert torch.equal(from_kmajor(dw.cpu().double(), Cout, Cin, K, K, 1), wr.grad)
    assert torch.equal(db.cpu().double(), dy_int.double().sum((0, 2, 3)))
    # full-mantissa weights, 1/255 scale: only the f32 accumulation (and the one multiplication by 1/scale) rounds
    d2 = desc(lib, Cin, H, W, Cout, K, S, in_u8=1, sub_mean=0.0, inv_scale=float(np.float32(1 / 255.0)), relu=0)
    w_f = torch.randn((Cout, Cin, K, K), generator=g) / 16
    lib.conv_fwd(x.cuda(), Cin * H * W, None, 0, to_kmajor(w_f, 1).cuda(), b_int.cuda() * 0, out, n, d2)
    inv = float(np.float32(1 / 255.0))
    ref = F.conv2d(x.double() * inv, w_f.double(), None, stride=S)
    got = out.view(n, 20, 20, Cout).permute(0, 3, 1, 2).cpu().double()
    assert (got - ref).abs().max().item() < 6e-7 * ref.abs().max().item()
    # ... and, because the products are exact, the error obeys the A-PRIORI bound of a pure f32 summation in ANY order:
    # |fl(sum) - sum| <= (m - 1) u / (1 - (m - 1) u) * sum |terms| with m = 3 * 256 exact terms and u = 2^-24, plus one
    # rounding each for the multiplication by 1/scale and for that constant itself — element by element, not just on the
    # largest value (a kernel that rounded an operand, e.g. x/255 in f32, would add K u sum|x w| on top of it)
    u = 2.0 ** -24
    mag = F.conv2d(x.double() * inv, w_f.double().abs(), None, stride=S)          # sum_k |x_k w_k| / 255
    bound = ((3 * 256 - 1) * u / (1 - 3 * 256 * u) + 2 * u) * mag + 1e-30
    assert ((got - ref).abs() <= bound).all(), float(((got - ref).abs() / bound).max())


@pytest.mark.parametrize("M,K,N", [(4096, 3136, 512), (257, 512, 7), (64, 8, 32), (33, 27, 5), (1000, 64, 64),
                                   (16384, 27, 64), (70001, 64, 17),   # (K, N <= 64: k_linear_wgrad_small)
                                   (16384, 64, 2048), (16399, 64, 1536)])  # (W_ih of a recurrent core: 64-row weight tiles / 64-row dgrad tiles)
def test_linear_fwd_bwd_vs_torch(lib, M, K, N):
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn((M, K), generator=g)
    w = torch.randn((K, N), generator=g) / np.sqrt(K)
    b = torch.randn(N, generator=g)
    dy = torch.randn((M, N), generator=g)
    xd, wd, bd, dyd = x.cuda(), w.cuda(), b.cuda(), dy.cuda()
    out = torch.empty((M, N), device="cuda")
    lib.linear_fwd(xd, wd, bd, out, M, K, N, True)
    ref = F.relu(x.double() @ w.double() + b.double())
    assert (out.cpu().double() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
    lib.linear_fwd(xd, wd, bd, out, M, K, N, False)
    ref = x.double() @ w.double() + b.double()
    assert (out.cpu().double() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
    dw, db = torch.zeros_like(wd), torch.zeros_like(bd)
    ws = torch.empty(lib.linear_wgrad_workspace(M, K, N), dtype=torch.uint8, device="cuda")
    lib.linear_wgrad(xd, dyd, dw, db, M, K, N, ws)
    rw = x.double().t() @ dy.double()
    assert (dw.cpu().double() - rw).abs().max().item() < 3e-5 * max(1.0, rw.abs().max().item())
    assert (db.cpu().double() - dy.double().sum(0)).abs().max().item() < 3e-5 * max(1.0, dy.double().sum(0).abs().max().item())
    din = torch.empty((M, K), device="cuda")
    lib.linear_dgrad(dyd, wd, xd, din, M, K, N)
    rd = (dy.double() @ w.double().t()) * (x.double() > 0)
    assert (din.cpu().double() - rd).abs().max().item() < 3e-5 * max(1.0, rd.abs().max().item())


def test_split_k_forward_small_grid(lib):
    """inference-batch FC (4096x3136x512: 256 tiles) takes the split-K path when a workspace is supplied"""
    M, K, N = 4096, 3136, 512
    g = torch.Generator().manual_seed(4)
    x, w, b = torch.randn((M, K), generator=g).cuda(), (torch.randn((K, N), generator=g) / 56).cuda(), torch.randn(N, generator=g).cuda()
    d = lib.sf_conv_desc(Cin=K, H=1, W=1, Cout=N, KH=1, KW=1, stride=1, OH=1, OW=1, in_u8=0, relu=1, traj_T=0, sub_mean=0.0, inv_scale=1.0)
    wsb = lib.conv_fwd_workspace(M, d)
    assert wsb > 0
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    o1, o2 = torch.empty((M, N), device="cuda"), torch.empty((M, N), device="cuda")
    lib.conv_fwd(x, K, None, 0, w, b, o1, M, d, ws)
    lib.conv_fwd(x, K, None, 0, w, b, o2, M, d, None)
    ref = F.relu(x.double() @ w.double() + b.double())
    assert (o1.double() - ref).abs().max().item() < 3e-5 * ref.abs().max().item()
    assert (o2.double() - ref).abs().max().item() < 3e-5 * ref.abs().max().item()
    o3 = torch.empty((M, N), device="cuda")
    lib.conv_fwd(x, K, None, 0, w, b, o3, M, d, ws)
    assert torch.equal(o1, o3)  # ordered partial sums: deterministic
    assert lib.conv_fwd_workspace(32768, d) == 0  # training batch fills the chip without splitting


def test_conv1_reads_slab_in_place(lib):
    """index gather + dataset->slab row mapping (flat e*T+t -> row e*(T+1)+t): integer indexing must be exact,
    so the result has to be BIT-identical to running the same kernel on a dense gathered copy."""
    E, T, n = 6, 4, 16
    g = torch.Generator().manual_seed(9)
    slab = torch.randint(0, 256, (E, T + 1, 4, 36, 36), generator=g, dtype=torch.uint8).cuda()
    idx = torch.randperm(E * T, generator=g)[:n].to(torch.int32).cuda()
    d = desc(lib, 4, 36, 36, 32, 8, 4, in_u8=1, inv_scale=float(np.float32(1 / 255.0)), traj_T=T)
    d0 = desc(lib, 4, 36, 36, 32, 8, 4, in_u8=1, inv_scale=float(np.float32(1 / 255.0)))
    w = torch.randn((4 * 64, 32), generator=g).cuda()
    b = torch.zeros(32).cuda()
    out1 = torch.empty((n * 64, 32), device="cuda")
    lib.conv_fwd(slab, 4 * 36 * 36, idx, 0, w, b, out1, n, d)
    e, t = idx.long() // T, idx.long() % T
    dense = slab[e, t].contiguous()
    out2 = torch.empty_like(out1)
    lib.conv_fwd(dense, 4 * 36 * 36, None, 0, w, b, out2, n, d0)
    assert torch.equal(out1, out2)
    out3 = torch.empty((8 * 64, 32), device="cuda")  # contiguous offset slice of the dataset
    lib.conv_fwd(slab, 4 * 36 * 36, None, 8, w, b, out3, 8, d)
    ii = torch.arange(8, 16)
    out4 = torch.empty_like(out3)
    lib.conv_fwd(slab[ii // T, ii % T].contiguous(), 4 * 36 * 36, None, 0, w, b, out4, 8, d0)
    assert torch.equal(out3, out4)
    # strided view slab[:, t] as used by the rollout / bootstrap forward
    out5 = torch.empty((E * 64, 32), device="cuda")
    lib.conv_fwd(slab[:, 2], slab.stride(0), None, 0, w, b, out5, E, d0)
    out6 = torch.empty_like(out5)
    lib.conv_fwd(slab[:, 2].contiguous(), 4 * 36 * 36, None, 0, w, b, out6, E, d0)
    assert torch.equal(out5, out6)


def test_conv1_strip_image_kernels_read_slab_in_place(lib):
    """the same exactness contract for the 84x84 strip-image kernels (forward AND weight gradient, n >= 256): index
    gather + dataset->slab row mapping, contiguous offset slices and the strided slab[:, t] view are bit-identical to
    a dense gathered copy."""
    E, T, n = 40, 8, 300
    g = torch.Generator().manual_seed(11)
    slab = torch.randint(0, 256, (E, T + 1, 4, 84, 84), generator=g, dtype=torch.uint8).cuda()
    idx = torch.randperm(E * T, generator=g)[:n].to(torch.int32).cuda()
    inv = float(np.float32(1 / 255.0))
    d = desc(lib, 4, 84, 84, 32, 8, 4, in_u8=1, inv_scale=inv, traj_T=T)
    d0 = desc(lib, 4, 84, 84, 32, 8, 4, in_u8=1, inv_scale=inv)
    w = (torch.randn((256, 32), generator=g) / 16).cuda()
    b = torch.randn(32, generator=g).cuda()
    S = 4 * 84 * 84
    out1 = torch.empty((n * 400, 32), device="cuda")
    lib.conv_fwd(slab, S, idx, 0, w, b, out1, n, d)
    e, t = idx.long() // T, idx.long() % T
    dense = slab[e, t].contiguous()
    out2 = torch.empty_like(out1)
    lib.conv_fwd(dense, S, None, 0, w, b, out2, n, d0)
    assert torch.equal(out1, out2)
    out3 = torch.empty((264 * 400, 32), device="cuda")
    lib.conv_fwd(slab, S, None, 8, w, b, out3, 264, d)
    ii = torch.arange(8, 272)
    out4 = torch.empty_like(out3)
    lib.conv_fwd(slab[ii // T, ii % T].contiguous(), S, None, 0, w, b, out4, 264, d0)
    assert torch.equal(out3, out4)
    big = torch.randint(0, 256, (256, 3, 4, 84, 84), generator=g, dtype=torch.uint8).cuda()
    out5 = torch.empty((256 * 400, 32), device="cuda")
    lib.conv_fwd(big[:, 1], big.stride(0), None, 0, w, b, out5, 256, d0)       # rollout-style strided view
    out6 = torch.empty_like(out5)
    lib.conv_fwd(big[:, 1].contiguous(), S, None, 0, w, b, out6, 256, d0)
    assert torch.equal(out5, out6)
    dy = torch.randn((n * 400, 32), generator=g).cuda()
    ws = torch.empty(lib.conv_wgrad_workspace(n, d), dtype=torch.uint8, device="cuda")
    dw1, db1, dw2, db2 = torch.zeros_like(w), torch.zeros(32, device="cuda"), torch.zeros_like(w), torch.zeros(32, device="cuda")
    lib.conv_wgrad(slab, S, idx, 0, dy, dw1, db1, n, d, ws)
    lib.conv_wgrad(dense, S, None, 0, dy, dw2, db2, n, d0, ws)
    assert torch.equal(dw1, dw2) and torch.equal(db1, db2)


def make_model(cfg_over, obs_shape, A):
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    from sample_factory_amd.model.actor_critic import ActorCritic
    cfg = default_cfg(use_rnn=False, nonlinearity="relu", normalize_input=False, encoder_conv_architecture="convnet_atari",
                      obs_scale=255.0, **cfg_over)
    obs_space = spaces.Dict({"obs": spaces.Box(0, 255, obs_shape, np.uint8)})
    return cfg, obs_space, ActorCritic(cfg, obs_space, spaces.Discrete(A), "cuda")


def load_seeded(ac, names, shapes, seed):
    st = seeded_state([(n, eval(s)) for n, s in zip(names, shapes)], seed)
    ac.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=False)
    return st


def test_model_forward_matches_reference_atari(lib, golden):
    """Nature-CNN actor-critic on u8 84x84x4 frames vs the REFERENCE model's outputs (model_fwd_atari.npz)."""
    g = golden("model_fwd_atari")
    cfg, _, ac = make_model({}, (4, 84, 84), 6)
    assert ac.num_params() == 1687719                                   # P of the reference model
    assert [n for n, _ in ac.ref_param_shapes()] == list(g["param_names"])
    load_seeded(ac, g["param_names"], g["param_shapes"], int(g["param_seed"]))
    obs = torch.from_numpy(g["obs"]).cuda()
    res = ac.forward({"obs": obs}, None)
    np.testing.assert_allclose(res["action_logits"].cpu().numpy(), g["action_logits"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(res["values"].cpu().numpy(), g["values"], atol=2e-5, rtol=1e-4)
    # integer-indexed consumers of the logits reproduce the reference's indices (argmax action, enjoy.py:177-182)
    np.testing.assert_array_equal(res["action_logits"].argmax(1).cpu().numpy(), g["action_logits"].argmax(1))
    # state_dict round trip in the reference's names / layouts
    sd = ac.state_dict()
    st = seeded_state([(n, eval(s)) for n, s in zip(g["param_names"], g["param_shapes"])], int(g["param_seed"]))
    for k, v in st.items():
        np.testing.assert_array_equal(sd[k].numpy(), v)


def test_learner_train_matches_reference_cnn36(lib, golden, tmp_path):
    """Full Learner.train on a trajectory batch (prepare_batch, 2 minibatches: fwd, loss, bwd, clip, Adam) vs the
    post-training parameters / Adam moments / grad norms / normaliser state of the REFERENCE Learner.train."""
    from sample_factory_amd.algo.learning.learner import Learner, ParameterServer
    from sample_factory_amd.algo.utils.env_info import EnvInfo
    from sample_factory_amd.algo.utils.shared_buffers import alloc_trajectory_tensors
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    g = golden("train_cnn36")
    E, T, A, nb = int(g["E"]), int(g["T"]), int(g["A"]), int(g["num_batches"])
    cfg = default_cfg(use_rnn=False, recurrence=1, nonlinearity="relu", normalize_input=False, obs_scale=255.0,
                      encoder_conv_architecture="convnet_atari", encoder_conv_mlp_layers=[128], rollout=T,
                      batch_size=E * T // nb, num_batches_per_epoch=nb, num_epochs=int(g["num_epochs"]), seed=0,
                      serial_mode=True, train_dir=str(tmp_path), experiment="t", record_grad_norm=True)
    obs_space = spaces.Dict({"obs": spaces.Box(0, 255, (4, 36, 36), np.uint8)})
    env_info = EnvInfo(obs_space, spaces.Discrete(A), E)
    pv = torch.zeros(1, dtype=torch.int32)
    learner = Learner(cfg, env_info, pv, 0, ParameterServer(0, pv))
    learner.init()
    ac = learner.actor_critic
    load_seeded(ac, g["param_names"], g["param_shapes"], int(g["param_seed"]))
    batch = alloc_trajectory_tensors(env_info, E, T, 1, "cuda")
    for k in ["rnn_states", "actions", "action_logits", "log_prob_actions", "values", "policy_version", "rewards",
              "dones", "time_outs", "policy_id", "valids"]:
        batch[k].copy_(torch.from_numpy(g["in_" + k]))
    batch["obs"]["obs"].copy_(torch.from_numpy(g["in_obs_obs"]))
    before = {k: v.clone() for k, v in ac.state_dict().items()}
    stats = learner.train(batch)
    assert stats["learner_env_steps"] == int(g["env_steps"]) and learner.train_step == int(g["train_step"])
    np.testing.assert_allclose(learner._grad_norms, g["grad_norms"], rtol=2e-4)
    np.testing.assert_allclose(ac.returns_normalizer.stats.cpu().numpy(), g["out_rms"], rtol=1e-5)
    compare_post_train(learner, g, before, "cnn36", **TIGHT)
    # checkpoint in the reference's format, reload, continue
    learner.save()
    l2 = Learner(cfg, env_info, pv, 0, ParameterServer(0, pv))
    l2.init()
    assert l2.train_step == learner.train_step and l2.env_steps == learner.env_steps
    assert torch.equal(l2.actor_critic.flat_params, ac.flat_params) and torch.equal(l2.exp_avg, learner.exp_avg)
    cp = torch.load(Learner.get_checkpoints(Learner.checkpoint_dir(cfg, 0))[-1], weights_only=False)
    assert set(cp) == {"train_step", "env_steps", "best_performance", "model", "optimizer", "curr_lr"}
    assert "encoder.encoders.obs.enc.conv_head.0.weight" in cp["model"] and cp["model"]["returns_normalizer.count"].dtype == torch.float64


@pytest.mark.parametrize("name", ["mlp", "mlp_inv", "mlp_lamb", "mlp_nonadaptive", "mlp_nonadaptive_tanh", "mlp_klmb_down",
                                  "mlp_klmb_up", "mlp_klep"])
def test_learner_train_matches_reference_mlp(lib, golden, tmp_path, name):
    """vector-observation MLP encoder with tanh (the reference's Mujoco-style model), 2 epochs / KL loss / invalid rows:
    full Learner.train vs the reference's post-training state (train_mlp*.npz)."""
    from sample_factory_amd.algo.learning.learner import Learner, ParameterServer
    from sample_factory_amd.algo.utils.env_info import EnvInfo
    from sample_factory_amd.algo.utils.shared_buffers import alloc_trajectory_tensors
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    g = golden("train_" + name)
    E, T, A, nb = int(g["E"]), int(g["T"]), int(g["A"]), int(g["num_batches"])
    kl = 0.1 if "kl_loss_coeff=0.1" in str(g["argv"]) else 0.0
    cfg = default_cfg(use_rnn=False, recurrence=1, nonlinearity="tanh", normalize_input=False, encoder_mlp_layers=[32, 32],
                      rollout=T, batch_size=E * T // nb, num_batches_per_epoch=nb, num_epochs=int(g["num_epochs"]),
                      kl_loss_coeff=kl, seed=0, serial_mode=True, train_dir=str(tmp_path), experiment="t",
                      record_grad_norm=True, optimizer="lamb" if "optimizer=lamb" in str(g["argv"]) else "adam")
    if "klmb" in name or "klep" in name:  # KL-adaptive schedules (learner.py:46-85): per minibatch the rate changes after EVERY
        # SGD step — here on the device (sf_lr_kl_adaptive + sf_adam_step_dlr), no read-back between the steps; per epoch it
        # changes with the epoch's read-back
        argv = dict(a.lstrip("-").split("=") for a in str(g["argv"]).split() if "=" in a)
        cfg.lr_schedule, cfg.lr_schedule_kl_threshold = argv["lr_schedule"], float(argv["lr_schedule_kl_threshold"])
        cfg.learning_rate = float(argv["learning_rate"])
    obs_space = spaces.Dict({"obs": spaces.Box(-10, 10, (8,), np.float32)})
    action_space = spaces.Discrete(A)
    if name.startswith("mlp_nonadaptive"):  # Box(3) with one learned log-stddev vector (action_parameterization.py:42-78)
        cfg.adaptive_stddev, cfg.initial_stddev = False, 0.7
        cfg.continuous_tanh_scale = 2.0 if name.endswith("tanh") else 0.0  # means squashed to [-2, 2] (:62-66)
        action_space = spaces.Box(-1, 1, (3,), np.float32)
    env_info = EnvInfo(obs_space, action_space, E)
    pv = torch.zeros(1, dtype=torch.int32)
    learner = Learner(cfg, env_info, pv, 0, ParameterServer(0, pv))
    learner.init()
    ac = learner.actor_critic
    assert [n for n, _ in ac.ref_param_shapes()] == list(g["param_names"])
    if name.startswith("mlp_nonadaptive"):
        assert abs(float(ac.state_dict()["action_parameterization.learned_stddev"][0]) - np.log(0.7)) < 1e-6
    load_seeded(ac, g["param_names"], g["param_shapes"], int(g["param_seed"]))
    batch = alloc_trajectory_tensors(env_info, E, T, 1, "cuda")
    for k in ["rnn_states", "actions", "action_logits", "log_prob_actions", "values", "policy_version", "rewards",
              "dones", "time_outs", "policy_id", "valids"]:
        batch[k].copy_(torch.from_numpy(g["in_" + k]))
    batch["obs"]["obs"].copy_(torch.from_numpy(g["in_obs_obs"]))
    before = {k: v.clone() for k, v in ac.state_dict().items()}
    stats = learner.train(batch)
    assert stats["learner_env_steps"] == int(g["env_steps"]) and learner.train_step == int(g["train_step"])
    np.testing.assert_allclose(learner._grad_norms, g["grad_norms"], rtol=3e-4)
    np.testing.assert_allclose(ac.returns_normalizer.stats.cpu().numpy(), g["out_rms"], rtol=1e-5)
    compare_post_train(learner, g, before, name, **TIGHT)
    if "klmb" in name or "klep" in name:
        assert hasattr(learner, "_lr_dev") == ("klmb" in name), "the per-minibatch schedule (and only it) runs on the device"
        np.testing.assert_allclose(learner.curr_lr, float(g["curr_lr"]), rtol=1e-6)
        assert abs(learner.curr_lr / cfg.learning_rate - 1.0) > 0.5   # the schedule moved the rate (x 1.5^4 or / 1.5^4)


@pytest.mark.parametrize("name", ["cnn36_norm", "mlp_norm"])
def test_learner_train_matches_reference_normalize_input(lib, golden, tmp_path, name):
    """normalize_input=True (the reference's default): per-element running mean/std of the observations, updated once
    per dataset in the learner and applied in both inference and training — vs the reference's Learner.train."""
    from sample_factory_amd.algo.learning.learner import Learner, ParameterServer
    from sample_factory_amd.algo.utils.env_info import EnvInfo
    from sample_factory_amd.algo.utils.shared_buffers import alloc_trajectory_tensors
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    g = golden("train_" + name)
    E, T, A, nb = int(g["E"]), int(g["T"]), int(g["A"]), int(g["num_batches"])
    if name.startswith("cnn"):
        over = dict(nonlinearity="relu", obs_scale=255.0, encoder_conv_architecture="convnet_atari", encoder_conv_mlp_layers=[128])
        obs_space = spaces.Dict({"obs": spaces.Box(0, 255, (4, 36, 36), np.uint8)})
    else:
        over = dict(nonlinearity="elu", encoder_mlp_layers=[32, 32])
        obs_space = spaces.Dict({"obs": spaces.Box(-10, 10, (8,), np.float32)})
    cfg = default_cfg(use_rnn=False, recurrence=1, normalize_input=True, rollout=T, batch_size=E * T // nb,
                      num_batches_per_epoch=nb, num_epochs=int(g["num_epochs"]), seed=0, serial_mode=True,
                      train_dir=str(tmp_path), experiment="t", record_grad_norm=True, **over)
    env_info = EnvInfo(obs_space, spaces.Discrete(A), E)
    pv = torch.zeros(1, dtype=torch.int32)
    learner = Learner(cfg, env_info, pv, 0, ParameterServer(0, pv))
    learner.init()
    ac = learner.actor_critic
    load_seeded(ac, g["param_names"], g["param_shapes"], int(g["param_seed"]))
    batch = alloc_trajectory_tensors(env_info, E, T, 1, "cuda")
    for k in ["rnn_states", "actions", "action_logits", "log_prob_actions", "values", "policy_version", "rewards",
              "dones", "time_outs", "policy_id", "valids"]:
        batch[k].copy_(torch.from_numpy(g["in_" + k]))
    batch["obs"]["obs"].copy_(torch.from_numpy(g["in_obs_obs"]))
    before = {k: v.clone() for k, v in ac.state_dict().items()}
    learner.train(batch)
    compare_post_train(learner, g, before, name, **TIGHT)
    sub = int(g["subsample"])
    sd = ac.state_dict()
    pfx = "obs_normalizer.running_mean_std.running_mean_std.obs."
    assert sd[pfx + "running_mean"].shape == tuple(obs_space["obs"].shape) and sd[pfx + "running_mean"].dtype == torch.float64
    np.testing.assert_allclose(sd[pfx + "running_mean"].reshape(-1)[::sub].numpy(), g["obsn_mean"], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(sd[pfx + "running_var"].reshape(-1)[::sub].numpy(), g["obsn_var"], rtol=1e-5, atol=1e-7)
    assert float(sd[pfx + "count"]) == float(g["obsn_count"])
    np.testing.assert_allclose(learner._grad_norms, g["grad_norms"], rtol=5e-4)
    ac.eval()
    res = ac.forward({"obs": batch["obs"]["obs"][:, 0].contiguous()}, None)
    np.testing.assert_allclose(res["action_logits"].cpu().numpy(), g["eval_logits"], atol=2e-4, rtol=2e-3)
    np.testing.assert_allclose(res["values"].cpu().numpy(), g["eval_values"], atol=2e-4, rtol=2e-3)
    # checkpoint round trip carries the statistics
    learner.save()
    l2 = Learner(cfg, env_info, pv, 0, ParameterServer(0, pv))
    l2.init()
    assert torch.equal(l2.actor_critic.obs_normalizer.mean, ac.obs_normalizer.mean)
    assert torch.equal(l2.actor_critic.obs_normalizer.mu_tab, ac.obs_normalizer.mu_tab)


@pytest.mark.parametrize("rnn_type", ["gru", "lstm"])
def test_rnn_cells_match_torch(lib, rnn_type):
    """GRU / LSTM cell forward + one-step backward vs torch.nn (CPU, fp64)"""
    Cn, F_, H = 37, 24, 16
    kind = 0 if rnn_type == "gru" else 1
    G = 3 if kind == 0 else 4
    g = torch.Generator().manual_seed(kind + 5)
    ref = (torch.nn.GRU if kind == 0 else torch.nn.LSTM)(F_, H).double()
    x, h0, c0 = torch.randn((Cn, F_), generator=g).double(), torch.randn((Cn, H), generator=g).double(), torch.randn((Cn, H), generator=g).double()
    h0r, c0r, xr = h0.clone().requires_grad_(True), c0.clone().requires_grad_(True), x.clone().requires_grad_(True)
    if kind == 0:
        out, hn = ref(xr[None], h0r[None]); cn = None
    else:
        out, (hn, cn) = ref(xr[None], (h0r[None], c0r[None]))
    dh = torch.randn((Cn, H), generator=g).double()
    dc = torch.randn((Cn, H), generator=g).double()
    ((out[0] * dh).sum() + ((cn[0] * dc).sum() if kind == 1 else 0)).backward()
    f32 = lambda t: t.detach().float().cuda().contiguous()
    w_ih, w_hh = f32(ref.weight_ih_l0.t()), f32(ref.weight_hh_l0.t())      # K-major [in, G*H]
    gx, gh = torch.empty((Cn, G * H), device="cuda"), torch.empty((Cn, G * H), device="cuda")
    lib.linear_fwd(f32(x), w_ih, f32(ref.bias_ih_l0), gx, Cn, F_, G * H, False)
    lib.linear_fwd(f32(h0), w_hh, f32(ref.bias_hh_l0), gh, Cn, H, G * H, False)
    gates = torch.empty((Cn, 4 * H), device="cuda")
    h_out, c_out = torch.empty((Cn, H), device="cuda"), torch.empty((Cn, H), device="cuda")
    h_next, c_next = torch.empty((Cn, H), device="cuda"), torch.empty((Cn, H), device="cuda")
    keep = (torch.rand(Cn, generator=g) > 0.3).float().cuda()
    lib.rnn_cell_fwd(kind, gx, gh, f32(h0), H, f32(c0), H, keep, Cn, H, gates, h_out, c_out, h_next, c_next)
    np.testing.assert_allclose(h_out.cpu().numpy(), hn[0].detach().numpy(), atol=2e-6)
    np.testing.assert_allclose(h_next.cpu().numpy(), (hn[0].detach() * keep.cpu()[:, None].double()).numpy(), atol=2e-6)
    if kind == 1:
        np.testing.assert_allclose(c_out.cpu().numpy(), cn[0].detach().numpy(), atol=2e-6)
    dgx, dgh = torch.empty_like(gx), torch.empty_like(gx)
    dh_direct, dc_prev = torch.zeros((Cn, H), device="cuda"), torch.zeros((Cn, H), device="cuda")
    lib.rnn_cell_bwd(kind, f32(dh), f32(dc) if kind == 1 else None, gates, f32(h0), H, f32(c0), H, c_out, Cn, H, dgx,
                     dgh if kind == 0 else None, dh_direct, dc_prev)
    dgh_eff = dgh if kind == 0 else dgx
    dx = dgx.cpu().double() @ ref.weight_ih_l0.detach()
    dhp = dgh_eff.cpu().double() @ ref.weight_hh_l0.detach() + (dh_direct.cpu().double() if kind == 0 else 0)
    np.testing.assert_allclose(dx.numpy(), xr.grad.numpy(), atol=3e-6)
    np.testing.assert_allclose(dhp.numpy(), h0r.grad.numpy(), atol=3e-6)
    if kind == 1:
        np.testing.assert_allclose(dc_prev.cpu().numpy(), c0r.grad.numpy(), atol=3e-6)


@pytest.mark.parametrize("name", ["gru", "lstm_inv", "gru2", "lstm2"])
def test_learner_train_matches_reference_rnn(lib, golden, tmp_path, name):
    """Recurrent policies: the reference's PackedSequence BPTT (rnn_utils.py) vs the native masked time loop —
    full Learner.train (stored chunk-start states, resets on dones / invalid rows, 2 minibatches [x 2 epochs]).
    gru2 / lstm2: cfg.rnn_num_layers = 2 (nn.GRU / nn.LSTM with two stacked layers, model/core.py:27-30,42-58) on the NATIVE
    model since round 6: one (input projection, recurrent projection) pair per layer on the same cell kernels, layer l's state
    in columns [l * SL, (l + 1) * SL) of the state rows."""
    from sample_factory_amd.algo.learning.learner import Learner, ParameterServer
    from sample_factory_amd.algo.utils.env_info import EnvInfo
    from sample_factory_amd.algo.utils.shared_buffers import alloc_trajectory_tensors
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    from sample_factory_amd.model.actor_critic import get_rnn_size
    g = golden("train_" + name)
    E, T, A, nb = int(g["E"]), int(g["T"]), int(g["A"]), int(g["num_batches"])
    rnn_type = "gru" if name.startswith("gru") else "lstm"
    layers = 2 if name.endswith("2") else 1
    cfg = default_cfg(use_rnn=True, rnn_type=rnn_type, rnn_size=32, rnn_num_layers=layers, recurrence=8, nonlinearity="relu",
                      normalize_input=False, encoder_mlp_layers=[32], rollout=T, batch_size=E * T // nb,
                      num_batches_per_epoch=nb, num_epochs=int(g["num_epochs"]),
                      kl_loss_coeff=0.1 if rnn_type == "lstm" else 0.0, seed=0, serial_mode=True, train_dir=str(tmp_path),
                      experiment="t", record_grad_norm=True)
    obs_space = spaces.Dict({"obs": spaces.Box(-10, 10, (8,), np.float32)})
    env_info = EnvInfo(obs_space, spaces.Discrete(A), E)
    pv = torch.zeros(1, dtype=torch.int32)
    learner = Learner(cfg, env_info, pv, 0, ParameterServer(0, pv))
    learner.init()
    ac = learner.actor_critic
    from sample_factory_amd.model.actor_critic import ActorCritic
    assert isinstance(ac, ActorCritic) and ac.rnn_L == layers
    assert [n for n, _ in ac.ref_param_shapes()] == list(g["param_names"])
    load_seeded(ac, g["param_names"], g["param_shapes"], int(g["param_seed"]))
    batch = alloc_trajectory_tensors(env_info, E, T, get_rnn_size(cfg), "cuda")
    for k in ["rnn_states", "actions", "action_logits", "log_prob_actions", "values", "policy_version", "rewards",
              "dones", "time_outs", "policy_id", "valids"]:
        batch[k].copy_(torch.from_numpy(g["in_" + k]))
    batch["obs"]["obs"].copy_(torch.from_numpy(g["in_obs_obs"]))
    before = {k: v.clone() for k, v in ac.state_dict().items()}
    stats = learner.train(batch)
    assert stats["learner_env_steps"] == int(g["env_steps"]) and learner.train_step == int(g["train_step"])
    np.testing.assert_allclose(learner._grad_norms, g["grad_norms"], rtol=5e-4)
    np.testing.assert_allclose(ac.returns_normalizer.stats.cpu().numpy(), g["out_rms"], rtol=1e-5)
    compare_post_train(learner, g, before, name, **TIGHT)
    after = ac.state_dict()
    assert "core.core.weight_hh_l0" in after and after["core.core.weight_ih_l0"].shape == ((3 if rnn_type == "gru" else 4) * 32, 32)
    assert (f"core.core.weight_hh_l{layers - 1}" in after) and (f"core.core.weight_ih_l{layers}" not in after)


def test_recurrent_policy_rollout_and_training(lib):
    """GRU policy on image observations end to end: state carried through the slab, reset on dones, BPTT"""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_synthetic_env
    from sample_factory_amd.train import make_runner
    register_env("synthetic_atari", make_synthetic_env)
    cfg = default_cfg(env="synthetic_atari", use_rnn=True, rnn_type="gru", rnn_size=64, nonlinearity="relu",
                      normalize_input=False, obs_scale=255.0, encoder_conv_architecture="convnet_atari", rollout=8,
                      batch_size=256, num_batches_per_epoch=2, num_epochs=1, num_workers=1, num_envs_per_worker=1,
                      async_rl=False, seed=1, serial_mode=True, synthetic_num_agents=64)
    cfg, runner = make_runner(cfg)
    runner.init()
    assert cfg.recurrence == 8
    for _ in range(2):
        stats = runner.iteration()
    torch.cuda.synchronize()
    tr = runner.traj
    assert tr["rnn_states"].shape == (64, 9, 64) and torch.isfinite(tr["rnn_states"]).all()
    done_prev = tr["dones"][:, :-1]
    assert (tr["rnn_states"][:, 1:-1][done_prev].abs().sum() == 0)      # state zeroed after a done step
    assert tr["rnn_states"][:, 1:].abs().sum() > 0 and np.isfinite(stats["train"]["loss"])


def _kv(argv):
    return {t[2:].split("=", 1)[0]: t[2:].split("=", 1)[1] for t in str(argv).split() if t.startswith("--") and "=" in t}


@pytest.mark.parametrize("case", ["ff_default", "ff_invalids", "ff_bootstrap_nonorm", "ff_continuous", "ff_vtrace",
                                  "ff_tuple", "ff_tuple_symkl", "ff_tuple_mixed"])
def test_learner_prepare_batch_and_losses_match_reference(lib, golden, tmp_path, case):
    """The native Learner's _prepare_batch + _calculate_losses (network forward included) replayed on the reference's
    golden batches: discrete / invalid+stale samples + KL loss / value bootstrap + symmetric-KL / Box actions / V-trace."""
    from sample_factory_amd.algo.learning.learner import Learner, ParameterServer
    from sample_factory_amd.algo.utils.env_info import EnvInfo
    from sample_factory_amd.algo.utils.shared_buffers import alloc_trajectory_tensors
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    g = golden("learner_" + case)
    kv = _kv(g["argv"])
    E, T = g["in_rewards"].shape
    continuous = case == "ff_continuous"
    A = g["in_action_logits"].shape[-1]
    action_space = spaces.Box(-1, 1, (A // 2,), np.float32) if continuous else spaces.Discrete(A)
    if "head_sizes" in g:
        action_space = spaces.Tuple([spaces.Discrete(int(nh)) if int(nh) > 0 else spaces.Box(-1, 1, (-int(nh),), np.float32)
                                     for nh in g["head_sizes"]])  # -D: a Box(D) member (mixed Tuple)
    over = dict(exploration_loss=kv.get("exploration_loss", "entropy"),
                exploration_loss_coeff=float(kv.get("exploration_loss_coeff", 0.003)),
                kl_loss_coeff=float(kv.get("kl_loss_coeff", 0.0)), max_policy_lag=int(kv.get("max_policy_lag", 1000)),
                normalize_returns=kv.get("normalize_returns", "True") == "True",
                value_bootstrap=kv.get("value_bootstrap", "False") == "True",
                with_vtrace=kv.get("with_vtrace", "False") == "True", recurrence=int(kv.get("recurrence", 1)),
                vtrace_rho=float(kv.get("vtrace_rho", 1.0)), vtrace_c=float(kv.get("vtrace_c", 1.0)))
    cfg = default_cfg(use_rnn=False, nonlinearity="tanh", normalize_input=False, encoder_mlp_layers=[32, 32], rollout=T,
                      batch_size=E * T // 2, num_batches_per_epoch=2, num_epochs=1, seed=0, serial_mode=True,
                      train_dir=str(tmp_path), experiment="t", **over)
    obs_space = spaces.Dict({"obs": spaces.Box(-10, 10, (8,), np.float32)})
    env_info = EnvInfo(obs_space, action_space, E)
    pv = torch.zeros(1, dtype=torch.int32)
    learner = Learner(cfg, env_info, pv, 0, ParameterServer(0, pv))
    learner.init()
    ac = learner.actor_critic
    names = [n for n, _ in ac.ref_param_shapes()]
    st = seeded_state(ac.ref_param_shapes(), int(g["param_seed"]))
    ac.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=False)
    assert sum(int(np.prod(s)) for _, s in ac.ref_param_shapes()) == int(g["num_params"])
    learner.train_step = int(g["train_step"])
    if cfg.normalize_returns:
        ac.returns_normalizer.stats.copy_(torch.as_tensor(g["in_rms"]))
    batch = alloc_trajectory_tensors(env_info, E, T, 1, "cuda")
    for k in ["rnn_states", "actions", "action_logits", "log_prob_actions", "values", "policy_version", "rewards",
              "dones", "time_outs", "policy_id", "valids"]:
        batch[k].copy_(torch.from_numpy(g["in_" + k]))
    batch["obs"]["obs"].copy_(torch.from_numpy(g["in_obs_obs"]))
    buff, size, num_invalids = learner._prepare_batch(batch)
    assert size == int(g["pb_size"]) and num_invalids == int(g["pb_num_invalids"])
    np.testing.assert_allclose(batch["values"][:, -1].cpu().numpy(), g["bootstrap_values"], atol=2e-6, rtol=1e-5)
    np.testing.assert_array_equal(buff.valids.cpu().numpy(), g["pb_valids"])
    np.testing.assert_allclose(batch["rewards"].cpu().numpy(), g["out_rewards"], atol=2e-6)
    if not cfg.with_vtrace:
        np.testing.assert_allclose(buff.advantages.cpu().numpy(), g["pb_advantages"], atol=1e-5)   # north star: 1e-4
        np.testing.assert_allclose(buff.returns.cpu().numpy(), g["pb_returns"], atol=1e-5)
    if cfg.normalize_returns:
        np.testing.assert_allclose(ac.returns_normalizer.stats.cpu().numpy(), g["out_rms"], rtol=1e-6)
    mb = int(g["mb_size"])
    acts, g_heads, sc = learner._losses_native(buff, (None, 0, mb), num_invalids)
    sc = sc.cpu().numpy()
    heads = acts[-1].cpu().numpy()
    np.testing.assert_allclose(heads[:, 1:1 + A], g["l_params"], atol=3e-6, rtol=1e-5)
    np.testing.assert_allclose(heads[:, 0], g["l_values"], atol=3e-6, rtol=1e-5)
    for i, k in enumerate(["policy_loss", "exploration_loss", "kl_loss", "value_loss"]):
        ref = float(g["l_" + k])
        assert abs(sc[i] - ref) < 3e-6 + 2e-5 * abs(ref), (k, sc[i], ref)
    assert abs(sc[6] - float(g["l_adv_mean"])) < 2e-6 and abs(sc[7] - float(g["l_adv_std"])) < 3e-6
    gh = g_heads.cpu().numpy()
    np.testing.assert_allclose(gh[:, 1:1 + A], g["l_grad_params"], atol=3e-7, rtol=3e-4)
    np.testing.assert_allclose(gh[:, 0], g["l_grad_values"], atol=3e-7, rtol=3e-4)
    assert np.all(gh[:, 1 + A:] == 0)


def test_action_mask_end_to_end(lib):
    """obs["action_mask"]: stored in the slab next to the observation, honoured by the sampler (a masked-out action is
    never taken), ignored by the encoder and — as in the reference — by the learner's distribution; the policy learns
    the bandit."""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_masked_bandit_env
    from sample_factory_amd.train import make_runner
    register_env("masked_bandit", make_masked_bandit_env)
    cfg = default_cfg(env="masked_bandit", use_rnn=False, nonlinearity="tanh", normalize_input=False,
                      encoder_mlp_layers=[32], rollout=8, batch_size=256, num_batches_per_epoch=2, num_epochs=2,
                      num_workers=1, num_envs_per_worker=1, async_rl=False, seed=4, serial_mode=True,
                      synthetic_num_agents=64, learning_rate=3e-3, gamma=0.0, normalize_returns=False)
    cfg, runner = make_runner(cfg)
    runner.init()
    assert runner.traj["obs"]["action_mask"].shape == (64, 9, 6)
    first = None
    for it in range(25):
        runner.iteration()
        tr = runner.traj
        # (column 0 of the slab already holds the NEXT rollout's first observation/mask: carry_over)
        a = tr["actions"][:, 1:, 0].long()
        mk = tr["obs"]["action_mask"][:, 1:-1]
        assert mk.gather(-1, a.unsqueeze(-1)).all()                     # the sampler never picks a masked-out action
        # recorded log-prob = masked log-softmax of the recorded raw logits at the action
        lg = tr["action_logits"][:, 1:] + (mk == 0) * -1e9
        lp = torch.log_softmax(lg, -1).gather(-1, a.unsqueeze(-1)).squeeze(-1)
        assert (lp - tr["log_prob_actions"][:, 1:]).abs().max() < 1e-5
        r = float(tr["rewards"].mean())
        first = r if first is None else first
    assert r > first + 0.25 and r > 0.7, (first, r)                      # random policy over ~3.5 allowed actions: ~0.3


def test_tuple_action_space_end_to_end(lib):
    """Tuple(Discrete(6), Discrete(3)) policy: two categorical heads sampled and trained through the native path"""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_synthetic_tuple_env
    from sample_factory_amd.train import make_runner
    register_env("synthetic_tuple", make_synthetic_tuple_env)
    cfg = default_cfg(env="synthetic_tuple", use_rnn=False, nonlinearity="relu", normalize_input=False, obs_scale=255.0,
                      encoder_conv_architecture="convnet_atari", rollout=8, batch_size=256, num_batches_per_epoch=2,
                      num_epochs=1, num_workers=1, num_envs_per_worker=1, async_rl=False, seed=2, serial_mode=True,
                      synthetic_num_agents=64, kl_loss_coeff=0.05)
    cfg, runner = make_runner(cfg)
    runner.init()
    ac = runner.learner.actor_critic
    assert ac.num_action_params == 9 and runner.traj["actions"].shape == (64, 8, 2)
    for _ in range(3):
        stats = runner.iteration()
    torch.cuda.synchronize()
    a = runner.traj["actions"]
    assert ((a[..., 0] >= 0) & (a[..., 0] < 6)).all() and ((a[..., 1] >= 0) & (a[..., 1] < 3)).all()
    assert np.isfinite(stats["train"]["loss"]) and torch.isfinite(ac.flat_params).all()
    # recorded log-prob = sum of the two heads' log-softmax at the recorded actions
    lg = runner.traj["action_logits"]
    lp = (torch.log_softmax(lg[..., :6], -1).gather(-1, a[..., :1].long()) +
          torch.log_softmax(lg[..., 6:], -1).gather(-1, a[..., 1:].long())).squeeze(-1)
    assert (lp - runner.traj["log_prob_actions"]).abs().max() < 1e-5


@pytest.mark.parametrize("vtrace", [False, True])
def test_mixed_tuple_action_space_end_to_end(lib, vtrace):
    """Tuple(Discrete(6), Box(2), Discrete(3)) — a Tuple with a Box member, which the reference's TupleActionDistribution
    composes like any other (action_distributions.py:197-287): sampled by sf_sample_write_step_tuple (categorical heads +
    a diagonal normal), trained through sf_ppo_loss (and sf_vtrace); the env receives the reference's per-member list
    (batched_sampling.py:51-59: int32 [agents] for a Discrete member, f32 [agents, D] for the Box member)"""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_synthetic_tuple_env
    from sample_factory_amd.train import make_runner
    register_env("synthetic_tuple", make_synthetic_tuple_env)
    cfg = default_cfg(env="synthetic_tuple", use_rnn=False, nonlinearity="relu", normalize_input=False, obs_scale=255.0,
                      encoder_conv_architecture="convnet_atari", rollout=8, recurrence=8 if vtrace else 1, batch_size=256,
                      num_batches_per_epoch=2, num_epochs=1, num_workers=1, num_envs_per_worker=1, async_rl=False, seed=2,
                      serial_mode=True, synthetic_num_agents=64, kl_loss_coeff=0.05, synthetic_head_sizes=(6, -2, 3),
                      with_vtrace=vtrace, normalize_returns=not vtrace, shuffle_minibatches=not vtrace)
    cfg, runner = make_runner(cfg)
    runner.init()
    ac = runner.learner.actor_critic
    assert ac.num_action_params == 6 + 4 + 3 and runner.traj["actions"].shape == (64, 8, 4)
    p0 = ac.flat_params.clone()
    for _ in range(3):
        stats = runner.iteration()
    torch.cuda.synchronize()
    a = runner.traj["actions"]
    assert ((a[..., 0] >= 0) & (a[..., 0] < 6) & (a[..., 0] == a[..., 0].round())).all()
    assert ((a[..., 3] >= 0) & (a[..., 3] < 3) & (a[..., 3] == a[..., 3].round())).all()
    assert a[..., 1:3].std() > 0.1 and torch.isfinite(a).all()                     # the Box member's two dims: real-valued
    assert np.isfinite(stats["train"]["loss"]) and torch.isfinite(ac.flat_params).all() and not torch.equal(p0, ac.flat_params)
    # recorded log-prob = categorical(6) + Normal over 2 dims + categorical(3) at the recorded actions
    lg = runner.traj["action_logits"]
    mu, sd = lg[..., 6:8], lg[..., 8:10].exp().clamp(1e-4, 1e4)
    lp = (torch.log_softmax(lg[..., :6], -1).gather(-1, a[..., :1].long()).squeeze(-1) +
          torch.distributions.Normal(mu, sd).log_prob(a[..., 1:3]).sum(-1) +
          torch.log_softmax(lg[..., 10:], -1).gather(-1, a[..., 3:].long()).squeeze(-1))
    assert (lp - runner.traj["log_prob_actions"]).abs().max() < 2e-5
    # what the env was handed at the last step: the reference's per-member list
    last = runner.samplers[0].env.last_actions
    assert isinstance(last, list) and len(last) == 3
    assert last[0].dtype == torch.int32 and tuple(last[0].shape) == (64,) and tuple(last[1].shape) == (64, 2) \
        and last[1].dtype == torch.float32 and last[2].dtype == torch.int32
    assert torch.equal(last[0].float(), a[:, -1, 0]) and torch.equal(last[1], a[:, -1, 1:3])


def test_continuous_env_rollout_and_vtrace_training(lib):
    """Box actions end to end: generic (non zero-copy) GPU env -> Normal sampler -> slab -> V-trace learner."""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_synthetic_continuous_env
    from sample_factory_amd.train import make_runner
    register_env("synthetic_ant", make_synthetic_continuous_env)
    # third variant = BASELINE.json configs[4] in miniature: LSTM core + V-trace + Box actions (mujoco_params.py preset)
    for vtrace, lstm in ((False, False), (True, False), (True, True)):
        cfg = default_cfg(env="synthetic_ant", use_rnn=lstm, rnn_type="lstm", rnn_size=64, nonlinearity="tanh",
                          normalize_input=lstm, encoder_mlp_layers=[64, 64], rollout=8,
                          recurrence=8 if (vtrace or lstm) else 1, batch_size=512,
                          num_batches_per_epoch=2, num_epochs=2, num_workers=1, num_envs_per_worker=1, async_rl=False,
                          seed=2, serial_mode=True, synthetic_num_agents=128, kl_loss_coeff=0.1, with_vtrace=vtrace,
                          normalize_returns=not vtrace, shuffle_minibatches=not vtrace)
        cfg, runner = make_runner(cfg)
        runner.init()
        p0 = runner.learner.actor_critic.flat_params.clone()
        for _ in range(2):
            stats = runner.iteration()
        torch.cuda.synchronize()
        tr = runner.traj
        assert stats["learner_env_steps"] == 2 * 128 * 8 and tr["actions"].shape == (128, 8, 8)
        assert torch.isfinite(tr["actions"]).all() and torch.isfinite(tr["log_prob_actions"]).all()
        assert torch.isfinite(runner.learner.actor_critic.flat_params).all()
        assert not torch.equal(p0, runner.learner.actor_critic.flat_params)
        assert np.isfinite(stats["train"]["loss"]) and stats["train"]["kl_divergence"] >= -1e-6


@pytest.mark.parametrize("act,kind", [("tanh", 2), ("elu", 3)])
def test_activation_kinds_fwd_bwd(lib, act, kind):
    M, K, N = 300, 64, 96
    g = torch.Generator().manual_seed(kind)
    x, w, b = torch.randn((M, K), generator=g), torch.randn((K, N), generator=g) / 8, torch.randn(N, generator=g)
    dy = torch.randn((M, 32), generator=g)
    w2 = torch.randn((N, 32), generator=g) / 8
    d = lib.sf_conv_desc(Cin=K, H=1, W=1, Cout=N, KH=1, KW=1, stride=1, OH=1, OW=1, in_u8=0, relu=kind, traj_T=0, sub_mean=0.0, inv_scale=1.0)
    out = torch.empty((M, N), device="cuda")
    lib.conv_fwd(x.cuda(), K, None, 0, w.cuda(), b.cuda(), out, M, d)
    xr = x.double().requires_grad_(True)
    pre = xr @ w.double() + b.double()
    y = torch.tanh(pre) if act == "tanh" else F.elu(pre)
    assert (out.cpu().double() - y.detach()).abs().max().item() < 2e-5
    # next layer's dgrad fuses this activation's derivative: d(pre) = (dy @ w2^T) * act'(y)
    d2 = lib.sf_conv_desc(Cin=N, H=1, W=1, Cout=32, KH=1, KW=1, stride=1, OH=1, OW=1, in_u8=0, relu=kind, traj_T=0, sub_mean=0.0, inv_scale=1.0)
    din = torch.empty((M, N), device="cuda")
    lib.conv_dgrad(dy.cuda(), w2.cuda(), out, din, M, d2)
    ref = (dy.double() @ w2.double().t()) * ((1 - y.detach() ** 2) if act == "tanh" else torch.where(y.detach() > 0, torch.ones_like(y), y.detach() + 1))
    assert (din.cpu().double() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())


def test_end_to_end_rollout_and_train_small(lib):
    """synthetic env -> rollout into the slab -> train, a few iterations; checks the slab protocol invariants"""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_synthetic_env
    from sample_factory_amd.train import make_runner
    register_env("synthetic_atari", make_synthetic_env)
    cfg = default_cfg(env="synthetic_atari", use_rnn=False, nonlinearity="relu", normalize_input=False, obs_scale=255.0,
                      encoder_conv_architecture="convnet_atari", rollout=8, batch_size=256, num_batches_per_epoch=2,
                      num_epochs=1, num_workers=1, num_envs_per_worker=1, async_rl=False, seed=1, serial_mode=True,
                      synthetic_num_agents=64, exploration_loss_coeff=0.01, shuffle_minibatches=True,
                      summaries_every_train=True)
    cfg, runner = make_runner(cfg)
    runner.init()
    p0 = runner.learner.actor_critic.flat_params.clone()
    for _ in range(3):
        stats = runner.iteration()
    torch.cuda.synchronize()
    tr = runner.traj
    assert stats["learner_env_steps"] == 3 * 64 * 8
    assert torch.all(tr["policy_id"] == 0) and tr["valids"][:, :-1].all()
    assert ((tr["actions"] >= 0) & (tr["actions"] < 6)).all()
    assert torch.isfinite(runner.learner.actor_critic.flat_params).all()
    assert not torch.equal(p0, runner.learner.actor_critic.flat_params)
    assert (tr["policy_version"] == 4.0).all()   # third rollout was collected by the policy after 2*2 SGD steps
    # the reference's train/* summary set (learner.py:843-923), device-reduced, one readback
    ts = stats["train"]
    for k in ["lr", "actual_lr", "valids_fraction", "same_policy_fraction", "grad_norm", "loss", "value", "entropy",
              "policy_loss", "kl_loss", "value_loss", "exploration_loss", "act_min", "act_max", "adv_min", "adv_max",
              "adv_std", "adv_mean", "max_abs_logprob", "kl_divergence", "kl_divergence_max", "value_delta",
              "value_delta_max", "fraction_clipped", "ratio_mean", "ratio_min", "ratio_max", "num_sgd_steps",
              "adam_max_second_moment", "version_diff_avg", "version_diff_min", "version_diff_max"]:
        assert k in ts and np.isfinite(ts[k]), k
    assert ts["valids_fraction"] == 1.0 and ts["same_policy_fraction"] == 1.0 and ts["num_sgd_steps"] == 2
    assert ts["ratio_min"] <= 1.0 <= ts["ratio_max"] and 0.0 <= ts["fraction_clipped"] <= 1.0 and ts["ratio_mean"] >= 0
    assert 0 <= ts["act_min"] <= ts["act_max"] <= 5 and ts["adv_min"] <= ts["adv_mean"] <= ts["adv_max"]
    assert ts["version_diff_min"] == ts["version_diff_max"] == 2.0     # data is two SGD steps old at the last minibatch
    assert ts["grad_norm"] > 0 and ts["adam_max_second_moment"] > 0 and ts["value_delta_max"] >= ts["value_delta"]
    # rewards follow the env rule given the recorded actions (identical-rollout bookkeeping)
    import oracle
    step_last = 3 * 8 - 1
    r_ref, _ = oracle.synth_step(tr["actions"][:, -1, 0].cpu().numpy().astype(np.int32), 0, 6, 1, step_last)
    np.testing.assert_array_equal(tr["rewards"][:, -1].cpu().numpy(), r_ref)
    # the frame after the last step sits in slot T; the next rollout's set_slab() carries it over to slot 0
    np.testing.assert_array_equal(tr["obs"]["obs"][:, 8].cpu().numpy().reshape(64, -1), oracle.synth_obs(64, 0, 28224, 1, 24))
    s = runner.sampler.episode_stats()
    assert s["episodes"] >= 0
    # PBT hooks (learner.py:388-428): new hyper-parameters are picked up at the next train() call
    runner.learner.set_new_cfg(dict(learning_rate=3e-4, exploration_loss_coeff=0.02, ppo_clip_ratio=0.2))
    runner.iteration()
    assert runner.learner.curr_lr == 3e-4 and abs(runner.learner.loss_cfg.exploration_coeff - 0.02) < 1e-9
    assert abs(runner.learner.loss_cfg.clip_ratio - 0.2) < 1e-7 and runner.learner.new_cfg is None


@pytest.mark.parametrize("variant", ["ff", "gru_norm"])
def test_async_rl_two_streams_equal_same_schedule_on_one_stream(lib, variant):
    """cfg.async_rl: rollout k+1 on its own stream overlaps Learner.train(k) (two slabs, published weight snapshots).
    The overlapped run must be bit-identical to the same schedule issued on ONE stream (any missing event / shared
    scratch buffer / torn snapshot would show up as a difference), and the recorded policy lag is one dataset."""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_synthetic_env
    from sample_factory_amd.train import make_runner
    register_env("synthetic_atari", make_synthetic_env)
    extra = dict(use_rnn=True, rnn_type="gru", rnn_size=64, normalize_input=True) if variant == "gru_norm" else \
        dict(use_rnn=False, normalize_input=False)

    def run(one_stream):
        cfg = default_cfg(env="synthetic_atari", nonlinearity="relu", obs_scale=255.0,
                          encoder_conv_architecture="convnet_atari", rollout=8, batch_size=1024, num_batches_per_epoch=2,
                          num_epochs=1, num_workers=1, num_envs_per_worker=1, async_rl=True, seed=3, serial_mode=False,
                          synthetic_num_agents=256, shuffle_minibatches=True, **extra)
        cfg, runner = make_runner(cfg)
        runner.init()
        if one_stream:
            runner.rollout_stream = torch.cuda.current_stream()
        stats = None
        for _ in range(6):
            stats = runner.iteration() or stats
        torch.cuda.synchronize()
        return runner, stats

    ra, sa = run(False)
    rb, sb = run(True)
    assert ra.rollout_stream != torch.cuda.current_stream()
    assert torch.equal(ra.learner.actor_critic.flat_params, rb.learner.actor_critic.flat_params)
    for k in ("actions", "rewards", "values", "policy_version", "rnn_states"):
        assert torch.equal(ra.slabs[1][k], rb.slabs[1][k]), k
    assert sa["train"]["loss"] == sb["train"]["loss"] and np.isfinite(sa["train"]["loss"])
    assert sa["learner_env_steps"] == 5 * 256 * 8            # 6 iterations = 6 rollouts, 5 trained datasets
    # rollout 5 (slab 1) ran on the weights published after train(3): 4 datasets * 2 SGD steps; the learner was one
    # dataset ahead when it trained on it -> lag of 2 SGD steps, as async APPO records (inference_worker.py:313-330)
    assert (ra.slabs[1]["policy_version"] == 8.0).all() and ra.learner.train_step == 10


def test_user_torch_model_inference_runs_in_eval_mode_and_frozen_parameters_are_published(lib, tmp_path):
    """A user encoder with Dropout and a frozen (requires_grad=False) parameter: rollout forwards ("inf") run in eval mode
    whether they use the learner's module (sync) or a published snapshot (async_rl, the reference's default) — the
    reference's inference worker holds an eval() copy — while the training forward keeps train mode; publish_weights also
    copies frozen parameters, which are not part of the flat trainable buffer"""
    from torch import nn
    from sample_factory_amd.algo.learning.learner import Learner, ParameterServer
    from sample_factory_amd.algo.utils.env_info import EnvInfo
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    from sample_factory_amd.model.model_factory import global_model_factory
    from sample_factory_amd.model.torch_policy import TorchPolicyAdapter

    class DropEncoder(nn.Module):
        def __init__(self, cfg, obs_space):
            super().__init__()
            self.net = nn.Sequential(nn.Linear(8, 32), nn.Tanh(), nn.Dropout(0.5))
            self.gain = nn.Parameter(torch.ones(32), requires_grad=False)

        def forward(self, obs_dict):
            return self.net(obs_dict["obs"]) * self.gain

        def get_out_size(self):
            return 32

    obs_space = spaces.Dict({"obs": spaces.Box(-10, 10, (8,), np.float32)})
    env_info = EnvInfo(obs_space, spaces.Discrete(4), 16)
    pv = torch.zeros(1, dtype=torch.int32)
    global_model_factory().register_encoder_factory(lambda cfg, obs_space: DropEncoder(cfg, obs_space))
    try:
        lr = Learner(default_cfg(train_dir=str(tmp_path), use_rnn=False, normalize_input=False, rollout=8, batch_size=64,
                                 num_batches_per_epoch=2, serial_mode=True, experiment="t"), env_info, pv, 0,
                     ParameterServer(0, pv))
        lr.init()
    finally:
        global_model_factory().reset()
    ac = lr.actor_critic
    assert isinstance(ac, TorchPolicyAdapter) and ac.module.training
    x = torch.randn(64, 8, device="cuda")
    a = ac.forward_heads(x, 64, sample_stride=8, tag="inf")[-1].clone()
    b = ac.forward_heads(x, 64, sample_stride=8, tag="inf")[-1].clone()
    assert torch.equal(a, b) and ac.module.training            # no dropout noise in rollouts; train mode restored
    t1 = ac.forward_heads(x, 64, sample_stride=8, tag="train")[-1].clone()
    t2 = ac.forward_heads(x, 64, sample_stride=8, tag="train")[-1].clone()
    assert not torch.equal(t1, t2)                              # the learner's forward does see Dropout
    ac.enable_weight_snapshots()
    ac.publish_weights(0)
    ac.snap_read = 0
    s0 = ac.forward_heads(x, 64, sample_stride=8, tag="inf")[-1].clone()
    assert torch.equal(s0, a)                                   # snapshot (eval copy) == sync-mode inference
    with torch.no_grad():
        ac.module.encoder.gain.mul_(2.0)                        # a frozen parameter changed by a user callback
    ac.publish_weights(0)
    s1 = ac.forward_heads(x, 64, sample_stride=8, tag="inf")[-1].clone()
    assert not torch.equal(s1, s0)                              # ... reaches the published copy


def test_user_registered_torch_model_trains_through_the_native_path(lib, tmp_path):
    """Model plugin surface (model/model_factory.py:16-60): a custom encoder registered with
    global_model_factory().register_encoder_factory keeps working — the network runs through torch autograd, everything
    around it (sampling, slab, GAE, PPO loss fwd/bwd, clip, Adam on the flat buffer, checkpoints) stays native.  The
    fallback must agree with the NATIVE model when the custom encoder is the default MLP with the same weights."""
    from torch import nn
    from sample_factory_amd.algo.learning.learner import Learner, ParameterServer
    from sample_factory_amd.algo.utils.env_info import EnvInfo
    from sample_factory_amd.algo.utils.shared_buffers import alloc_trajectory_tensors
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    from sample_factory_amd.model.model_factory import global_model_factory
    from sample_factory_amd.model.torch_policy import TorchPolicyAdapter

    class MyEncoder(nn.Module):
        def __init__(self, cfg, obs_space):
            super().__init__()
            self.net = nn.Sequential(nn.Linear(8, 32), nn.Tanh(), nn.Linear(32, 32), nn.Tanh())

        def forward(self, obs_dict):
            return self.net(obs_dict["obs"])

        def get_out_size(self):
            return 32

    E, T, A = 16, 8, 6
    cfgkw = dict(use_rnn=False, recurrence=1, nonlinearity="tanh", normalize_input=False, encoder_mlp_layers=[32, 32],
                 rollout=T, batch_size=E * T // 2, num_batches_per_epoch=2, num_epochs=2, seed=0, serial_mode=True,
                 experiment="t", kl_loss_coeff=0.1)
    obs_space = spaces.Dict({"obs": spaces.Box(-10, 10, (8,), np.float32)})
    env_info = EnvInfo(obs_space, spaces.Discrete(A), E)
    pv = torch.zeros(1, dtype=torch.int32)
    native = Learner(default_cfg(train_dir=str(tmp_path / "a"), **cfgkw), env_info, pv, 0, ParameterServer(0, pv))
    native.init()
    global_model_factory().register_encoder_factory(lambda cfg, obs_space: MyEncoder(cfg, obs_space))
    try:
        custom = Learner(default_cfg(train_dir=str(tmp_path / "b"), **cfgkw), env_info, pv, 0, ParameterServer(0, pv))
        custom.init()
    finally:
        global_model_factory().reset()
    ca = custom.actor_critic
    assert isinstance(ca, TorchPolicyAdapter) and ca.num_params() == native.actor_critic.num_params()
    # same weights in both: reference-layout state dict of the native model -> the torch module
    sd = native.actor_critic.state_dict()
    ca.load_state_dict({"encoder.net.0.weight": sd["encoder.encoders.obs.mlp_head.0.weight"],
                        "encoder.net.0.bias": sd["encoder.encoders.obs.mlp_head.0.bias"],
                        "encoder.net.2.weight": sd["encoder.encoders.obs.mlp_head.2.weight"],
                        "encoder.net.2.bias": sd["encoder.encoders.obs.mlp_head.2.bias"],
                        "critic_linear.weight": sd["critic_linear.weight"], "critic_linear.bias": sd["critic_linear.bias"],
                        "action_parameterization.distribution_linear.weight":
                            sd["action_parameterization.distribution_linear.weight"],
                        "action_parameterization.distribution_linear.bias":
                            sd["action_parameterization.distribution_linear.bias"]}, strict=False)
    g = torch.Generator().manual_seed(0)
    batch = alloc_trajectory_tensors(env_info, E, T, 1, "cuda")
    batch["obs"]["obs"].copy_(torch.randn((E, T + 1, 8), generator=g))
    batch["actions"].copy_(torch.randint(0, A, (E, T, 1), generator=g).float())
    batch["action_logits"].copy_(torch.randn((E, T, A), generator=g) * 0.5)
    batch["log_prob_actions"].copy_(-torch.rand((E, T), generator=g) - 0.5)
    batch["values"].copy_(torch.randn((E, T + 1), generator=g))
    batch["rewards"].copy_(torch.randn((E, T), generator=g))
    batch["dones"].copy_(torch.rand((E, T), generator=g) < 0.1)
    batch["time_outs"].zero_()
    batch["policy_id"].zero_()
    batch["policy_version"].zero_()
    from sample_factory_amd.algo.utils.tensor_dict import clone_tensordict
    s1 = native.train(clone_tensordict(batch))
    s2 = custom.train(clone_tensordict(batch))
    assert s1["learner_env_steps"] == s2["learner_env_steps"] and custom.train_step == native.train_step == 4
    assert abs(s1["train"]["loss"] - s2["train"]["loss"]) < 1e-5
    after_n, after_c = native.actor_critic.state_dict(), ca.state_dict()
    for kn, kc in [("encoder.encoders.obs.mlp_head.0.weight", "encoder.net.0.weight"),
                   ("encoder.encoders.obs.mlp_head.2.bias", "encoder.net.2.bias"),
                   ("critic_linear.weight", "critic_linear.weight"),
                   ("action_parameterization.distribution_linear.weight", "action_parameterization.distribution_linear.weight")]:
        assert (after_n[kn] - after_c[kc]).abs().max() < 2e-5, kn      # 4 Adam steps at lr 1e-4
        assert not torch.equal(after_c[kc], sd[kn])                     # ... and it did train
    custom.save()   # checkpoint round trip with the module's own parameter names
    assert "encoder.net.0.weight" in torch.load(Learner.get_checkpoints(Learner.checkpoint_dir(custom.cfg, 0))[-1],
                                                weights_only=False)["model"]


def test_learner_config_matrix_smoke(lib, tmp_path):
    """Learner.train over a matrix of model / action-space / loss / optimiser options on batches large enough
    (2048-4096 samples, 64-wide layers) that the MLP layers take the large-grid LDS-DMA kernels too: every combination
    must run, stay finite, move the weights and give gradients that agree with a finite-difference probe of the
    total loss along the gradient direction."""
    from sample_factory_amd.algo.learning.learner import Learner, ParameterServer
    from sample_factory_amd.algo.utils.env_info import EnvInfo
    from sample_factory_amd.algo.utils.shared_buffers import alloc_trajectory_tensors
    from sample_factory_amd.algo.utils.tensor_dict import clone_tensordict
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    from sample_factory_amd.model.actor_critic import get_rnn_size
    combos = [
        dict(space="disc"), dict(space="tuple", kl_loss_coeff=0.1), dict(space="box", with_vtrace=True, normalize_returns=False),
        dict(space="tuple", with_vtrace=True, normalize_returns=False, recurrence=8),
        dict(space="box", adaptive_stddev=False, optimizer="lamb"), dict(space="disc", use_rnn=True, rnn_type="gru"),
        dict(space="tuple", use_rnn=True, rnn_type="lstm", exploration_loss="symmetric_kl"),
        dict(space="disc", normalize_input=True, shuffle_minibatches=True, num_epochs=2),
        dict(space="disc", nonlinearity="elu", value_bootstrap=True, optimizer="lamb", max_grad_norm=0.0),
    ]
    obs_space = spaces.Dict({"obs": spaces.Box(-10, 10, (64,), np.float32)})
    for ci, c in enumerate(combos):
        c = dict(c)
        sp = c.pop("space")
        action_space = {"disc": spaces.Discrete(5), "tuple": spaces.Tuple([spaces.Discrete(4), spaces.Discrete(3)]),
                        "box": spaces.Box(-1, 1, (3,), np.float32)}[sp]
        use_rnn = c.get("use_rnn", False)
        E, T = 256, 16
        kw = dict(use_rnn=use_rnn, recurrence=T if use_rnn else 1, rnn_size=64, nonlinearity="tanh", normalize_input=False,
                  encoder_mlp_layers=[64, 64], rollout=T, batch_size=E * T // 2, num_batches_per_epoch=2, num_epochs=1,
                  seed=ci, serial_mode=True, train_dir=str(tmp_path / f"c{ci}"), experiment="t", learning_rate=1e-3)
        kw.update(c)
        cfg = default_cfg(**kw)
        env_info = EnvInfo(obs_space, action_space, E)
        pv = torch.zeros(1, dtype=torch.int32)
        learner = Learner(cfg, env_info, pv, 0, ParameterServer(0, pv))
        learner.init()
        ac = learner.actor_critic
        g = torch.Generator().manual_seed(100 + ci)
        b = alloc_trajectory_tensors(env_info, E, T, get_rnn_size(cfg), "cuda")
        b["obs"]["obs"].copy_(torch.randn((E, T + 1, 64), generator=g))
        na = b["actions"].shape[-1]
        if sp == "box":
            b["actions"].copy_(torch.randn((E, T, na), generator=g))
        elif sp == "tuple":
            b["actions"].copy_(torch.cat([torch.randint(0, 4, (E, T, 1), generator=g), torch.randint(0, 3, (E, T, 1), generator=g)], 2).float())
        else:
            b["actions"].copy_(torch.randint(0, 5, (E, T, 1), generator=g).float())
        b["action_logits"].copy_(torch.randn(b["action_logits"].shape, generator=g) * 0.3)
        b["log_prob_actions"].copy_(-torch.rand((E, T), generator=g) - 0.5)
        b["values"].copy_(torch.randn((E, T + 1), generator=g))
        b["rewards"].copy_(torch.randn((E, T), generator=g))
        b["dones"].copy_(torch.rand((E, T), generator=g) < 0.05)
        b["time_outs"].copy_(b["dones"].cpu() & (torch.rand((E, T), generator=g) < 0.5))
        b["rnn_states"].copy_(torch.randn(b["rnn_states"].shape, generator=g) * 0.3 if use_rnn else torch.zeros(b["rnn_states"].shape))
        b["policy_id"].zero_()
        b["policy_version"].zero_()
        p0 = ac.flat_params.clone()
        stats = learner.train(clone_tensordict(b))
        torch.cuda.synchronize()
        tag = f"combo {ci}: {sp} {c}"
        assert stats is not None and np.isfinite(stats["train"]["loss"]), tag
        assert torch.isfinite(ac.flat_params).all() and not torch.equal(p0, ac.flat_params), tag
        # directional finite-difference check of the analytic gradient of the first minibatch (fresh learner state)
        ac.flat_params.copy_(p0)
        ac.params_changed()
        buff, size, ninv = learner._prepare_batch(clone_tensordict(b))
        mb = learner._get_minibatches(cfg.batch_size, size)[0]
        if cfg.shuffle_minibatches or cfg.normalize_input:
            continue  # the index set / normaliser statistics change between calls: the probe needs a fixed function
        acts, g_heads, sc = learner._losses_native(buff, mb, ninv)
        ac.backward(acts, g_heads, buff.obs, mb[2], sample_stride=ac.obs_elems, index=mb[0], offset=mb[1], traj_T=buff.T)
        grad = ac.flat_grads.clone()
        direction = grad / (grad.norm() + 1e-12)
        lossf = lambda: float(learner._losses_native(buff, mb, ninv)[2][:4].sum().item())
        eps = 2e-3
        ac.flat_params.copy_(p0 + eps * direction); ac.params_changed(); lp = lossf()
        ac.flat_params.copy_(p0 - eps * direction); ac.params_changed(); lm = lossf()
        fd, an = (lp - lm) / (2 * eps), float(grad.norm())
        if cfg.with_vtrace:
            continue  # V-trace targets are recomputed from the perturbed policy (no gradient flows through them)
        assert abs(fd - an) < 0.05 * max(abs(an), 1e-3) + 2e-3, (tag, fd, an)


def test_cartpole_learns(lib):
    """BASELINE.json configs[0] as a learning test (the reference's own end-to-end check is a learning test too,
    tests/examples/test_example.py:159-174): host CartPole env, MLP policy, sync APPO on the GPU; the mean episode length
    must rise well above the ~22 steps of a random policy."""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.cartpole import make_cartpole_env
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.train import make_runner
    register_env("CartPole-v1", make_cartpole_env)
    cfg = default_cfg(env="CartPole-v1", use_rnn=False, nonlinearity="tanh", normalize_input=True, encoder_mlp_layers=[64, 64],
                      rollout=32, batch_size=1024, num_batches_per_epoch=2, num_epochs=4, num_workers=1,
                      num_envs_per_worker=1, async_rl=False, seed=3, serial_mode=True, cartpole_num_agents=64,
                      learning_rate=1e-3, exploration_loss_coeff=0.001, gamma=0.99, value_bootstrap=True,
                      reward_scale=0.1, shuffle_minibatches=True)
    cfg, runner = make_runner(cfg)
    runner.init()
    lens = []
    for it in range(60):
        runner.sampler.ep_stats.zero_()
        runner.iteration()
        s = runner.sampler.episode_stats()
        if s["episodes"] > 0:
            lens.append(s["mean_len"])
    first, last = np.mean(lens[:5]), np.mean(lens[-5:])
    assert first < 60, first
    assert last > 150 and last > 3 * first, (first, last)


@pytest.mark.parametrize("async_rl", [False, True])
def test_env_instances_on_split_streams_match_one_instance(lib, async_rl):
    """num_envs_per_worker = 2 instances of 512 envs, worker_num_splits = 2 (each instance's rollout on its own HIP
    stream, steps interleaved: the reference's double-buffered sampling) against ONE instance of 1024 envs: the env and
    the action sampler are keyed by the global env row, so both runs see the same frames; step 0 of the first rollout is
    computed from identical weights and inputs (different launch sizes may pick different split-K plans: tolerance on
    the values, a handful of sampled actions may flip)."""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_synthetic_env
    from sample_factory_amd.train import make_runner
    register_env("synthetic_split", make_synthetic_env)
    runs = []
    for inst in (1, 2):
        cfg = default_cfg(env="synthetic_split", use_rnn=False, rollout=8, recurrence=1, batch_size=2048,
                          num_batches_per_epoch=4, num_epochs=1, num_workers=1, num_envs_per_worker=inst,
                          worker_num_splits=2, async_rl=async_rl, seed=5, serial_mode=not async_rl,
                          synthetic_num_agents=1024 // inst)
        cfg, runner = make_runner(cfg)
        runner.init()
        # BufferMgr (shared_buffers.py:184-189): agents x env instances rows, twice that when rollouts overlap training
        assert len(runner.samplers) == inst and runner.traj["rewards"].shape == (2048 if async_rl else 1024, 8)
        stats = None
        for _ in range(3):
            stats = runner.iteration() or stats
        torch.cuda.synchronize()
        assert stats["learner_env_steps"] >= 1024 * 8
        assert torch.isfinite(runner.learner.actor_critic.flat_params).all()
        assert runner.episode_stats()["episodes"] >= 0
        runs.append(runner)
    if async_rl:
        return  # slabs are ping-ponged: the state checks below are for the synchronous schedule
    a, b = runs[0].traj, runs[1].traj
    assert torch.equal(a["obs"]["obs"][:, 0], b["obs"]["obs"][:, 0]), "same env rows, same frames"
    # the third rollout used weights after two updates on (nearly) the same data: still close
    assert (a["values"][:, 0] - b["values"][:, 0]).abs().max().item() < 5e-2
    runs2 = []
    for inst in (1, 2):  # fresh runners: the very first rollout step (identical weights) must agree tightly
        cfg = default_cfg(env="synthetic_split", use_rnn=False, rollout=8, recurrence=1, batch_size=2048,
                          num_batches_per_epoch=4, num_epochs=1, num_workers=1, num_envs_per_worker=inst,
                          worker_num_splits=2, async_rl=False, seed=5, serial_mode=True, synthetic_num_agents=1024 // inst)
        cfg, runner = make_runner(cfg)
        runner.init()
        runner._rollout_all(0.0)
        torch.cuda.synchronize()
        runs2.append(runner.traj)
    a, b = runs2
    assert torch.equal(a["obs"]["obs"], b["obs"]["obs"]) or (a["actions"] == b["actions"]).float().mean() > 0.99
    assert (a["values"][:, 0] - b["values"][:, 0]).abs().max().item() < 1e-5
    assert (a["action_logits"][:, 0] - b["action_logits"][:, 0]).abs().max().item() < 1e-5
    assert (a["actions"][:, 0] == b["actions"][:, 0]).float().mean().item() > 0.995
    assert torch.equal(a["rewards"][:, 0][a["actions"][:, 0, 0] == b["actions"][:, 0, 0]],
                       b["rewards"][:, 0][a["actions"][:, 0, 0] == b["actions"][:, 0, 0]])


def test_runner_plugin_hooks_observer_msg_handlers_training_info(lib):
    """The plugin points of the reference runner (algo/runners/runner.py:52-73, 232-249, 481-495) and of the env
    (envs/env_utils.py:110-133): an AlgoObserver sees init / start / every training step / stop, a registered episodic
    stats handler receives {EPISODIC: {reward, len, ...}, policy_id}, a message handler keyed on "train" sees the
    learner's report, and an env implementing TrainingInfoInterface is told approx_total_training_steps before every
    rollout."""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import TrainingInfoInterface, register_env
    from sample_factory_amd.envs.synthetic import SyntheticVecEnv
    from sample_factory_amd.train import EPISODIC, AlgoObserver, make_runner

    class CurriculumEnv(SyntheticVecEnv, TrainingInfoInterface):
        def __init__(self, **kw):
            SyntheticVecEnv.__init__(self, **kw)
            TrainingInfoInterface.__init__(self)
            self.seen = []

        def set_training_info(self, training_info):
            super().set_training_info(training_info)
            self.seen.append(training_info["approx_total_training_steps"])

    register_env("synthetic_curriculum", lambda name, cfg=None, env_config=None, render_mode=None:
                 CurriculumEnv(num_agents=256, seed=1))

    class Obs(AlgoObserver):
        def __init__(self):
            self.calls = []

        def on_init(self, runner): self.calls.append("init")
        def on_connect_components(self, runner): self.calls.append("connect")
        def on_start(self, runner): self.calls.append("start")
        def on_training_step(self, runner, it): self.calls.append(("step", it))
        def extra_summaries(self, runner, policy_id, env_steps, writer): self.calls.append("summaries")
        def on_stop(self, runner): self.calls.append("stop")

    cfg = default_cfg(env="synthetic_curriculum", use_rnn=False, rollout=8, recurrence=1, batch_size=1024,
                      num_batches_per_epoch=2, num_epochs=1, num_workers=1, num_envs_per_worker=1, async_rl=False, seed=1,
                      serial_mode=True, train_for_env_steps=3 * 256 * 8, train_for_seconds=600, stats_avg=10)
    cfg, runner = make_runner(cfg)
    obs = Obs()
    runner.register_observer(obs)
    episodic, train_msgs = [], []
    runner.register_episodic_stats_handler(lambda r, msg, pid: episodic.append((dict(msg[EPISODIC]), pid)))
    runner.register_msg_handler("train", lambda r, msg: train_msgs.append(msg["train"]["loss"]))
    runner.report_interval_sec = 0.0  # report after every iteration
    assert runner.init() == 0
    assert obs.calls == ["init", "connect"]
    assert runner.run() == 0
    assert obs.calls[2] == "start" and obs.calls[-1] == "stop"
    assert [c for c in obs.calls if isinstance(c, tuple)] == [("step", 1), ("step", 2), ("step", 3)]
    assert obs.calls.count("summaries") == 3
    assert len(train_msgs) == 3 and all(np.isfinite(x) for x in train_msgs)
    assert runner.env.seen == [0, 256 * 8, 2 * 256 * 8], runner.env.seen
    # Bernoulli(1/1024) terminations over 3 * 2048 env steps: a few episodes finish; their stats arrive as the
    # reference's message and land in policy_avg_stats through the default handler
    assert episodic and all(pid == 0 and set(m) >= {"reward", "len", "episodes"} for m, pid in episodic)
    assert sum(m["episodes"] for m, _ in episodic) >= 1
    assert len(runner.policy_avg_stats["reward"][0]) == len(episodic)
    assert all(m["len"] >= 1 for m, _ in episodic)


@pytest.mark.parametrize("n,act", [(512, 1), (777, 2), (4099, 1), (600, 0)])
def test_lds_image_forward_conv3_vs_torch(lib, n, act):
    """sf_nn_img.h (persistent work-groups, sample images resident in an LDS ring, im2col in the LDS address, weights in
    registers) takes the 64 x 9 x 9 / 3x3 geometry from n = 512 up: ragged last fragment (n * 49 is not a multiple of
    16), fewer fragments than resident work-groups, every activation kind, result equal to the im2col kernel's."""
    Cin, H, W, Cout, K, S = 64, 9, 9, 64, 3, 1
    g = torch.Generator().manual_seed(n)
    x = torch.randn((n, Cin, H, W), generator=g)
    d = desc(lib, Cin, H, W, Cout, K, S)
    d.relu = act
    assert lib.conv_kernel_name(3, n, d) == "k_fwd_img<64, 9, 9, 3, 1, 2, 1, 7>", lib.conv_kernel_name(3, n, d)
    x_dev = x.permute(0, 2, 3, 1).contiguous().cuda()
    w_ref = torch.randn((Cout, Cin, K, K), generator=g) / np.sqrt(Cin * K * K)
    b = torch.randn(Cout, generator=g) * 0.1
    wk = to_kmajor(w_ref, 0).cuda()
    wt = wk.t().contiguous()
    out = torch.full((n * d.OH * d.OW + 7, Cout), 7.0, device="cuda")  # guard rows behind the last one
    assert lib.conv_fwd_t_supported(n, d) and lib.conv_fwd_t_workspace(n, d) == 0
    lib.conv_fwd_t(x_dev, Cin * H * W, wt, b.cuda(), out, n, d)
    pre = F.conv2d(x, w_ref, b, stride=S)
    ref = F.relu(pre) if act == 1 else torch.tanh(pre) if act == 2 else pre
    got = out[:n * d.OH * d.OW].view(n, d.OH, d.OW, Cout).permute(0, 3, 1, 2).cpu()
    assert (got - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
    assert (out[n * d.OH * d.OW:] == 7.0).all(), "rows past the end must not be written"
    out_old = torch.empty((n * d.OH * d.OW, Cout), device="cuda")
    lib.conv_fwd(x_dev, Cin * H * W, None, 0, wk, b.cuda(), out_old, n, d)
    assert (out[:n * d.OH * d.OW] - out_old).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
    # strided samples (a view into a larger buffer) and no bias
    big = torch.randn((n, 2, H, W, Cin), generator=g).cuda()
    out2 = torch.empty((n * d.OH * d.OW, Cout), device="cuda")
    lib.conv_fwd_t(big[:, 1], 2 * Cin * H * W, wt, None, out2, n, d)
    pre2 = F.conv2d(big[:, 1].permute(0, 3, 1, 2).cpu(), w_ref, None, stride=S)
    ref2 = F.relu(pre2) if act == 1 else torch.tanh(pre2) if act == 2 else pre2
    got2 = out2.view(n, d.OH, d.OW, Cout).permute(0, 3, 1, 2).cpu()
    assert (got2 - ref2).abs().max().item() < 3e-5 * max(1.0, ref2.abs().max().item())


@pytest.mark.parametrize("native", [True, False], ids=["native_towers", "torch_path"])
def test_multi_key_observation_dict_end_to_end(lib, tmp_path, monkeypatch, native):
    """Observation dicts with several keys (image + vector: the reference's MultiInputEncoder, model/encoder.py:33-69):
    every key lives in the slab, the default architecture (one encoder per sorted key, concatenated) runs on the native
    towers of model/actor_critic_multikey.py (SF_NATIVE_MULTIKEY=0: the torch path) with the reference's parameter names;
    the policy learns a bandit whose context is only in the VECTOR key; per-key input normalisation and checkpoints
    round-trip."""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_dict_obs_bandit_env
    from sample_factory_amd.model.actor_critic_multikey import MultiKeyActorCritic
    from sample_factory_amd.model.torch_policy import TorchPolicyAdapter
    from sample_factory_amd.train import make_runner
    monkeypatch.setenv("SF_NATIVE_MULTIKEY", "1" if native else "0")
    TorchPolicyAdapter = MultiKeyActorCritic if native else TorchPolicyAdapter  # noqa: F811 (the class this run must build)
    register_env("dict_bandit", make_dict_obs_bandit_env)
    cfg = default_cfg(env="dict_bandit", use_rnn=False, nonlinearity="relu", normalize_input=True,
                      normalize_input_keys=["measurements"], obs_scale=255.0, encoder_conv_architecture="convnet_impala",
                      encoder_conv_mlp_layers=[32], encoder_mlp_layers=[32], rollout=8, batch_size=256,
                      num_batches_per_epoch=2, num_epochs=2, num_workers=1, num_envs_per_worker=1, async_rl=False, seed=4,
                      serial_mode=True, synthetic_num_agents=64, learning_rate=3e-3, gamma=0.0, normalize_returns=False,
                      train_dir=str(tmp_path), experiment="dict")
    cfg, runner = make_runner(cfg)
    runner.init()
    ac = runner.learner.actor_critic
    assert isinstance(ac, TorchPolicyAdapter) and ac.multi_key and ac.obs_keys == ["measurements", "obs"]
    names = [n for n, _ in ac.ref_param_shapes()]
    assert "encoder.encoders.measurements.mlp_head.0.weight" in names
    assert "encoder.encoders.obs.enc.conv_head.0.weight" in names and "encoder.encoders.obs.enc.mlp_layers.0.bias" in names
    assert runner.traj["obs"]["obs"].shape == (64, 9, 4, 36, 36) and runner.traj["obs"]["obs"].dtype == torch.uint8
    assert runner.traj["obs"]["measurements"].shape == (64, 9, 4)
    first = None
    for it in range(30):
        runner.iteration()
        r = float(runner.traj["rewards"].mean())
        first = r if first is None else first
    assert r > first + 0.3 and r > 0.75, (first, r)  # 4 actions: a blind policy gets 0.25
    # the slab really carries both keys of the step the action was taken on: one-hot context <-> reward
    tr = runner.traj
    tgt = tr["obs"]["measurements"][:, 1:-1].argmax(-1)
    assert torch.equal((tr["actions"][:, 1:, 0].long() == tgt).float(), tr["rewards"][:, 1:])
    sd = ac.state_dict()
    assert "obs_normalizer.running_mean_std.running_mean_std.measurements.running_mean" in sd   # normalised key
    assert "obs_normalizer.running_mean_std.running_mean_std.obs.running_mean" not in sd        # not in the key list
    assert float(sd["obs_normalizer.running_mean_std.running_mean_std.measurements.count"]) > 64 * 9
    runner.learner.save()
    cfg2, runner2 = make_runner(cfg)  # restart_behavior = resume: picks the checkpoint up
    runner2.init()
    ac2 = runner2.learner.actor_critic
    assert torch.equal(ac2.flat_params, ac.flat_params), "resume loads the multi-key checkpoint"


@pytest.mark.parametrize("native", [True, False], ids=["native_towers", "torch_path"])
def test_multi_input_model_forward_matches_reference(lib, golden, monkeypatch, native):
    """the default architecture for an image + vector observation dict vs the REFERENCE model (model_fwd_multi.npz,
    MultiInputEncoder inside ActorCriticSharedWeights): identical parameter names and shapes, same logits / values
    from the same seeded weights (obs_scale only on the "obs" key) — on the native towers and on the torch path."""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    from sample_factory_amd.model.model_factory import create_actor_critic
    monkeypatch.setenv("SF_NATIVE_MULTIKEY", "1" if native else "0")
    g = golden("model_fwd_multi")
    cfg = default_cfg(encoder_conv_architecture="convnet_impala", nonlinearity="relu", obs_scale=255.0,
                      normalize_input=False, encoder_conv_mlp_layers=[32], encoder_mlp_layers=[16, 16], use_rnn=False)
    cfg.dp_world = 1
    obs_space = spaces.Dict({"obs": spaces.Box(0, 255, (4, 36, 36), np.uint8),
                             "measurements": spaces.Box(-1, 1, (5,), np.float32)})
    ac = create_actor_critic(cfg, obs_space, spaces.Discrete(6), torch.device("cuda"))
    assert type(ac).__name__ == ("MultiKeyActorCritic" if native else "TorchPolicyAdapter")
    assert [(n, str(tuple(s))) for n, s in ac.ref_param_shapes()] == \
        [(str(n), str(tuple(eval(str(s))))) for n, s in zip(g["param_names"], g["param_shapes"])]
    load_seeded(ac, g["param_names"], g["param_shapes"], int(g["param_seed"]))
    ac.eval()
    res = ac.forward({"obs": torch.from_numpy(g["obs"]).cuda(), "measurements": torch.from_numpy(g["measurements"]).cuda()},
                     None)
    np.testing.assert_allclose(res["action_logits"].cpu().numpy(), g["action_logits"], atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(res["values"].cpu().numpy(), g["values"], atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("tag", ["ff", "gru", "lstm"])
def test_native_separate_weights_forward_matches_reference(lib, tag):
    """cfg.actor_critic_share_weights=False on the native towers (model/actor_critic_separate.py): the reference's parameter
    names / shapes, and — with the same seeded weights — its logits, values and new recurrent state [actor | critic]
    (tests/golden/model_fwd_separate_*.npz, generated by running ActorCriticSeparateWeights)"""
    import os
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    from sample_factory_amd.model.actor_critic import get_rnn_size
    from sample_factory_amd.model.actor_critic_separate import SeparateActorCritic
    from sample_factory_amd.model.model_factory import create_actor_critic
    from oracle.weights import seeded_state
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"model_fwd_separate_{tag}.npz"), allow_pickle=True)
    kw = dict(ff=dict(use_rnn=False), gru=dict(use_rnn=True, rnn_type="gru", rnn_size=12, recurrence=4),
              lstm=dict(use_rnn=True, rnn_type="lstm", rnn_size=10, recurrence=4, decoder_mlp_layers=[14]))[tag]
    cfg = default_cfg(actor_critic_share_weights=False, encoder_mlp_layers=[16, 12], nonlinearity="tanh",
                      normalize_input=False, normalize_returns=False, **kw)
    cfg.dp_world = 1
    obs_space = spaces.Dict({"obs": spaces.Box(-10, 10, (8,), np.float32)})
    ac = create_actor_critic(cfg, obs_space, spaces.Discrete(5), torch.device("cuda"))
    assert isinstance(ac, SeparateActorCritic)
    assert [(n, tuple(s)) for n, s in ac.ref_param_shapes()] == \
        [(str(n), tuple(eval(str(s)))) for n, s in zip(g["param_names"], g["param_shapes"])]
    st = seeded_state([(str(n), eval(str(s))) for n, s in zip(g["param_names"], g["param_shapes"])], int(g["param_seed"]))
    ac.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=True)
    sd = ac.state_dict()
    for k, v in st.items():
        np.testing.assert_array_equal(sd[k].numpy(), v)  # round trip under the reference's names
    ac.eval()
    rnn = torch.from_numpy(g["rnn_states"]).cuda()
    assert rnn.shape[1] == get_rnn_size(cfg) == ac.rnn_S
    res = ac.forward({"obs": torch.from_numpy(g["obs"]).cuda()}, rnn)
    np.testing.assert_allclose(res["action_logits"].cpu().numpy(), g["action_logits"], atol=3e-6, rtol=2e-5)
    np.testing.assert_allclose(res["values"].cpu().numpy(), g["values"], atol=3e-6, rtol=2e-5)
    if tag != "ff":
        np.testing.assert_allclose(res["new_rnn_states"].cpu().numpy(), g["new_rnn_states"], atol=3e-6, rtol=2e-5)


def test_native_separate_weights_async_rollout_and_training(lib):
    """separate actor / critic weights end to end on the native towers in the reference's default mode (async_rl): published
    weight snapshots of both towers, recurrent state [actor | critic] carried through the slab, one flat buffer under Adam"""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_synthetic_continuous_env
    from sample_factory_amd.model.actor_critic_separate import SeparateActorCritic
    from sample_factory_amd.train import make_runner
    register_env("synthetic_ant", make_synthetic_continuous_env)
    cfg = default_cfg(env="synthetic_ant", actor_critic_share_weights=False, use_rnn=True, rnn_type="lstm", rnn_size=64,
                      nonlinearity="tanh", normalize_input=True, encoder_mlp_layers=[64, 64], rollout=8, recurrence=8,
                      batch_size=512, num_batches_per_epoch=2, num_epochs=1, num_workers=1, num_envs_per_worker=1, async_rl=True,
                      seed=4, serial_mode=False, synthetic_num_agents=128, kl_loss_coeff=0.1)
    cfg, runner = make_runner(cfg)
    runner.init()
    ac = runner.learner.actor_critic
    assert isinstance(ac, SeparateActorCritic) and ac.rnn_S == 4 * 64 and runner.traj["rnn_states"].shape[-1] == 256
    p0 = ac.flat_params.clone()
    for _ in range(4):
        stats = runner.iteration()
    torch.cuda.synchronize()
    runner.stop_sampler_thread()
    assert np.isfinite(stats["train"]["loss"]) and torch.isfinite(ac.flat_params).all() and not torch.equal(p0, ac.flat_params)
    na = ac.actor.num_flat
    assert not torch.equal(p0[:na], ac.flat_params[:na]) and not torch.equal(p0[na:], ac.flat_params[na:])  # both towers train
    st = runner.traj["rnn_states"]
    assert torch.isfinite(st).all() and st[:, 1:, :128].abs().sum() > 0 and st[:, 1:, 128:].abs().sum() > 0
    assert float(ac.actor.layers[-1].w[:, 0].abs().max()) == 0.0 and float(ac.critic.layers[-1].w[:, 1:].abs().max()) == 0.0


@pytest.mark.parametrize("native", [True, False], ids=["native_towers", "torch_path"])
@pytest.mark.parametrize("name", ["sep_gru", "sep_mlp"])
def test_learner_train_matches_reference_separate_actor_critic_weights(lib, golden, tmp_path, name, native, monkeypatch):
    """cfg.actor_critic_share_weights=False (ActorCriticSeparateWeights, model/actor_critic.py:198-334): the reference's
    Learner.train replayed — two encoders / cores, recurrent state [actor | critic] carried and chunk-started through the
    native slab kernels, loss / clip / Adam native.  native_towers (the default since round 6): two native towers on one flat
    parameter buffer (model/actor_critic_separate.py); torch_path (SF_NATIVE_SEPARATE_WEIGHTS=0): the torch module."""
    monkeypatch.setenv("SF_NATIVE_SEPARATE_WEIGHTS", "1" if native else "0")
    from sample_factory_amd.algo.learning.learner import Learner, ParameterServer
    from sample_factory_amd.algo.utils.env_info import EnvInfo
    from sample_factory_amd.algo.utils.shared_buffers import alloc_trajectory_tensors
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    from sample_factory_amd.model.actor_critic import get_rnn_size
    from sample_factory_amd.model.torch_policy import TorchPolicyAdapter
    g = golden("train_" + name)
    E, T, A, nb = int(g["E"]), int(g["T"]), int(g["A"]), int(g["num_batches"])
    rnn = name == "sep_gru"
    cfg = default_cfg(actor_critic_share_weights=False, use_rnn=rnn, rnn_type="gru", rnn_size=32, recurrence=8 if rnn else 1,
                      nonlinearity="relu", normalize_input=False, encoder_mlp_layers=[32], rollout=T, batch_size=E * T // nb,
                      num_batches_per_epoch=nb, num_epochs=int(g["num_epochs"]), kl_loss_coeff=0.0 if rnn else 0.1, seed=0,
                      serial_mode=True, train_dir=str(tmp_path), experiment="t", record_grad_norm=True)
    obs_space = spaces.Dict({"obs": spaces.Box(-10, 10, (8,), np.float32)})
    env_info = EnvInfo(obs_space, spaces.Discrete(A), E)
    pv = torch.zeros(1, dtype=torch.int32)
    learner = Learner(cfg, env_info, pv, 0, ParameterServer(0, pv))
    learner.init()
    ac = learner.actor_critic
    from sample_factory_amd.model.actor_critic_separate import SeparateActorCritic
    assert isinstance(ac, SeparateActorCritic if native else TorchPolicyAdapter)
    assert get_rnn_size(cfg) == (64 if rnn else 2) == int(g["in_rnn_states"].shape[-1])
    assert [n for n, _ in ac.ref_param_shapes()] == list(g["param_names"])
    load_seeded(ac, g["param_names"], g["param_shapes"], int(g["param_seed"]))
    batch = alloc_trajectory_tensors(env_info, E, T, get_rnn_size(cfg), "cuda")
    for k in ["rnn_states", "actions", "action_logits", "log_prob_actions", "values", "policy_version", "rewards",
              "dones", "time_outs", "policy_id", "valids"]:
        batch[k].copy_(torch.from_numpy(g["in_" + k]))
    batch["obs"]["obs"].copy_(torch.from_numpy(g["in_obs_obs"]))
    before = {k: v.clone() for k, v in ac.state_dict().items()}
    stats = learner.train(batch)
    assert stats["learner_env_steps"] == int(g["env_steps"]) and learner.train_step == int(g["train_step"])
    np.testing.assert_allclose(learner._grad_norms, g["grad_norms"], rtol=5e-4)
    compare_post_train(learner, g, before, name + ("" if native else "_torch_path"), **TIGHT)
    if native:  # the heads columns a tower does not own never move
        assert float(ac.actor.layers[-1].w[:, 0].abs().max()) == 0.0 and float(ac.critic.layers[-1].w[:, 1:].abs().max()) == 0.0


@pytest.mark.parametrize("name", ["gru", "lstm_inv", "gru2", "lstm2"])
def test_learner_train_matches_reference_rnn_on_the_torch_model_path(lib, golden, tmp_path, name):
    """The recurrent goldens once more, through the TORCH model path (a registered encoder, here the default
    architecture itself; also what observation dicts with several keys use): default one-layer GRU / LSTM core with
    the reference's parameter names, BPTT as a masked time loop under autograd, everything around the network native —
    against the reference's Learner.train (PackedSequence BPTT).  gru2 / lstm2: TWO stacked recurrent layers
    (cfg.rnn_num_layers = 2, model/core.py:19-64) on this path too (the native model takes them since round 6)."""
    from sample_factory_amd.algo.learning.learner import Learner, ParameterServer
    from sample_factory_amd.algo.utils.env_info import EnvInfo
    from sample_factory_amd.algo.utils.shared_buffers import alloc_trajectory_tensors
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    from sample_factory_amd.model.actor_critic import get_rnn_size
    from sample_factory_amd.model.model_factory import global_model_factory
    from sample_factory_amd.model.encoder import MultiInputEncoder as _TorchMultiInputEncoder
    from sample_factory_amd.model.torch_policy import TorchPolicyAdapter
    g = golden("train_" + name)
    E, T, A, nb = int(g["E"]), int(g["T"]), int(g["A"]), int(g["num_batches"])
    rnn_type = "gru" if name.startswith("gru") else "lstm"
    layers = 2 if name.endswith("2") else 1
    cfg = default_cfg(use_rnn=True, rnn_type=rnn_type, rnn_size=32, rnn_num_layers=layers, recurrence=8, nonlinearity="relu",
                      normalize_input=False, encoder_mlp_layers=[32], rollout=T, batch_size=E * T // nb,
                      num_batches_per_epoch=nb, num_epochs=int(g["num_epochs"]),
                      kl_loss_coeff=0.1 if rnn_type == "lstm" else 0.0, seed=0, serial_mode=True, train_dir=str(tmp_path),
                      experiment="t", record_grad_norm=True)
    obs_space = spaces.Dict({"obs": spaces.Box(-10, 10, (8,), np.float32)})
    env_info = EnvInfo(obs_space, spaces.Discrete(A), E)
    pv = torch.zeros(1, dtype=torch.int32)
    global_model_factory().register_encoder_factory(lambda c, o: _TorchMultiInputEncoder(c, o))  # -> the torch model path
    try:
        learner = Learner(cfg, env_info, pv, 0, ParameterServer(0, pv))
        learner.init()
    finally:
        global_model_factory().reset()
    ac = learner.actor_critic
    assert isinstance(ac, TorchPolicyAdapter) and ac.rnn_kind == (0 if rnn_type == "gru" else 1)
    assert get_rnn_size(cfg) == 32 * layers * (2 if rnn_type == "lstm" else 1) == int(g["in_rnn_states"].shape[-1])
    assert [n for n, _ in ac.ref_param_shapes()] == list(g["param_names"])
    load_seeded(ac, g["param_names"], g["param_shapes"], int(g["param_seed"]))
    batch = alloc_trajectory_tensors(env_info, E, T, get_rnn_size(cfg), "cuda")
    for k in ["rnn_states", "actions", "action_logits", "log_prob_actions", "values", "policy_version", "rewards",
              "dones", "time_outs", "policy_id", "valids"]:
        batch[k].copy_(torch.from_numpy(g["in_" + k]))
    batch["obs"]["obs"].copy_(torch.from_numpy(g["in_obs_obs"]))
    before = {k: v.clone() for k, v in ac.state_dict().items()}
    stats = learner.train(batch)
    assert stats["learner_env_steps"] == int(g["env_steps"]) and learner.train_step == int(g["train_step"])
    np.testing.assert_allclose(learner._grad_norms, g["grad_norms"], rtol=5e-4)
    compare_post_train(learner, g, before, name + "_torch_path", **TIGHT)


@pytest.mark.parametrize("native", [True, False], ids=["native_towers", "torch_path"])
def test_multi_key_observations_with_recurrent_core_rollout_and_training(lib, tmp_path, monkeypatch, native):
    """the reference's DEFAULT configuration for a dict observation: MultiInputEncoder + GRU core (use_rnn=True) — state
    carried through the slab by the native rollout runner, BPTT through the trunk tower's sequence passes (torch path: the
    masked time loop of model/torch_policy.py)"""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_dict_obs_bandit_env
    from sample_factory_amd.train import make_runner
    monkeypatch.setenv("SF_NATIVE_MULTIKEY", "1" if native else "0")
    register_env("dict_bandit_rnn", make_dict_obs_bandit_env)
    cfg = default_cfg(env="dict_bandit_rnn", use_rnn=True, rnn_type="gru", rnn_size=32, recurrence=8, nonlinearity="relu",
                      normalize_input=False, obs_scale=255.0, encoder_conv_architecture="convnet_impala",
                      encoder_conv_mlp_layers=[32], encoder_mlp_layers=[32], rollout=8, batch_size=256,
                      num_batches_per_epoch=2, num_epochs=1, num_workers=1, num_envs_per_worker=1, async_rl=False, seed=4,
                      serial_mode=True, synthetic_num_agents=64, learning_rate=3e-3, gamma=0.0, normalize_returns=False,
                      train_dir=str(tmp_path), experiment="dict_rnn")
    cfg, runner = make_runner(cfg)
    runner.init()
    ac = runner.learner.actor_critic
    assert type(ac).__name__ == ("MultiKeyActorCritic" if native else "TorchPolicyAdapter")
    assert ac.rnn_kind == 0 and "core.core.weight_hh_l0" in [n for n, _ in ac.ref_param_shapes()]
    p0 = ac.flat_params.clone()
    first = None
    for _ in range(30):
        stats = runner.iteration()
        r = float(runner.traj["rewards"].mean())
        first = r if first is None else first
    tr = runner.traj
    assert tr["rnn_states"].shape == (64, 9, 32)
    # every episode is one step long: the state handed to the next step is zero, the state the policy produced is not
    assert (tr["rnn_states"][:, 1:] == 0).all() and ac.new_rnn_states.abs().max() > 0
    assert torch.isfinite(ac.flat_params).all() and not torch.equal(p0, ac.flat_params)
    assert np.isfinite(stats["train"]["loss"]) and r > first + 0.3 and r > 0.75, (first, r)


@pytest.mark.parametrize("core", ["ff", "gru", "lstm_decoder", "ff_odd_widths", "gru_odd_widths", "ff_conv_last", "gru_conv_last",
                                  "separate_ff", "separate_gru"])
def test_native_multi_key_towers_against_the_torch_path(lib, monkeypatch, core):
    """model/actor_critic_multikey.py against the torch construction of the same architecture (model/torch_policy.py builds
    the reference's modules: MultiInputEncoder -> core -> decoder -> heads, model/encoder.py:33-69) on the same seeded
    weights and the same slab leaves: heads of a training pass (dataset-row addressing into [E, T + 1, ...] leaves, an
    index gather), of a one-step pass on a slab column, and EVERY parameter gradient under the reference's names; per-key
    normalisation (running statistics on the vector key only, mean shift / scale on "obs" only) included."""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    from sample_factory_amd.model.model_factory import create_actor_critic
    kw = dict(ff=dict(use_rnn=False), gru=dict(use_rnn=True, rnn_type="gru", rnn_size=32, recurrence=4),
              lstm_decoder=dict(use_rnn=True, rnn_type="lstm", rnn_size=32, recurrence=4, decoder_mlp_layers=[24]),
              # encoder widths 15 + 32 = 47 columns: no 16-byte alignment anywhere in the feature batch
              ff_odd_widths=dict(use_rnn=False, encoder_mlp_layers=[16, 15]),
              gru_odd_widths=dict(use_rnn=True, rnn_type="gru", rnn_size=32, recurrence=4, encoder_mlp_layers=[16, 15],
                                  encoder_conv_mlp_layers=[33])).get(core)
    # an image encoder WITHOUT fully connected layers: the 32 x 3 x 3 conv output is concatenated in the reference's CHW order
    kw_more = dict(ff_conv_last=dict(use_rnn=False, encoder_conv_mlp_layers=[]),
                   gru_conv_last=dict(use_rnn=True, rnn_type="gru", rnn_size=32, recurrence=4, encoder_conv_mlp_layers=[]),
                   # separate actor / critic weights x several keys: each tower of model/actor_critic_separate.py is the multi-key
                   # composite (actor_encoder.encoders.<key>.*, critic_encoder.encoders.<key>.*, ...)
                   separate_ff=dict(use_rnn=False, actor_critic_share_weights=False),
                   separate_gru=dict(use_rnn=True, rnn_type="gru", rnn_size=32, recurrence=4, actor_critic_share_weights=False))
    base = dict(encoder_conv_mlp_layers=[32], encoder_mlp_layers=[16, 16])
    base.update(kw if kw is not None else kw_more[core])
    cfg = default_cfg(encoder_conv_architecture="convnet_impala", nonlinearity="relu", obs_scale=255.0, obs_subtract_mean=3.0,
                      normalize_input=True, normalize_input_keys=["measurements"], normalize_returns=False, **base)
    cfg.dp_world = 1
    obs_space = spaces.Dict({"obs": spaces.Box(0, 255, (4, 36, 36), np.uint8),
                             "measurements": spaces.Box(-1, 1, (5,), np.float32)})
    models = {}
    for native in (True, False):
        monkeypatch.setenv("SF_NATIVE_MULTIKEY", "1" if native else "0")
        models[native] = create_actor_critic(cfg, obs_space, spaces.Discrete(6), torch.device("cuda"))
    nat, ref = models[True], models[False]
    assert type(nat).__name__ == ("SeparateActorCritic" if core.startswith("separate") else "MultiKeyActorCritic")
    assert type(ref).__name__ == "TorchPolicyAdapter" and nat.multi_key and nat.obs_keys == ["measurements", "obs"]
    shapes = ref.ref_param_shapes()
    assert [(n, tuple(s)) for n, s in nat.ref_param_shapes()] == [(n, tuple(s)) for n, s in shapes]
    st = seeded_state([(n, tuple(s)) for n, s in shapes], 21)
    for m in (nat, ref):
        m.load_state_dict({k: torch.from_numpy(v) for k, v in st.items()}, strict=False)
    E, T = 16, 4
    g = torch.Generator().manual_seed(5)
    obs = {"obs": torch.randint(0, 256, (E, T + 1, 4, 36, 36), generator=g, dtype=torch.int32).to(torch.uint8).cuda(),
           "measurements": (torch.rand((E, T + 1, 5), generator=g) * 2 - 1).cuda()}
    for m in (nat, ref):   # one statistics update of the normalised key, as the Learner does per dataset
        m.obs_normalizer.update(obs, m.obs_elems, E * (T + 1))
    sd_n, sd_r = nat.state_dict(), ref.state_dict()
    k_ = "obs_normalizer.running_mean_std.running_mean_std.measurements.running_mean"
    np.testing.assert_allclose(sd_n[k_].numpy(), sd_r[k_].numpy(), rtol=1e-6, atol=1e-7)
    assert "obs_normalizer.running_mean_std.running_mean_std.obs.running_mean" not in sd_n
    n = E * T
    S = nat.rnn_S if nat.rnn_kind is not None else 1
    rnn = None
    if nat.rnn_kind is not None:
        R, Cn = 4, n // 4
        keep = (torch.rand((R, Cn), generator=g) > 0.2).float().cuda()
        h0 = (torch.randn((Cn, S), generator=g) * 0.3).cuda()
        rnn = dict(R=R, h0=h0, keep_tm=keep)
    g_heads = torch.zeros((n, nat.heads_ld), device="cuda")
    g_heads[:, :1 + nat.num_action_params] = torch.randn((n, 1 + nat.num_action_params), generator=g).cuda() / n
    outs = {}
    for name, m in (("native", nat), ("torch", ref)):
        m.train()
        m.flat_grads.zero_()
        acts = m.forward_heads(obs, n, sample_stride=m.obs_elems, index=None, offset=0, traj_T=T, tag="train",
                               rnn=None if rnn is None else dict(rnn))
        heads = acts[-1].clone()
        m.backward(acts, g_heads.clone(), obs, n, sample_stride=m.obs_elems, index=None, offset=0, traj_T=T)
        torch.cuda.synchronize()
        grads = m.flat_to_ref(m.flat_grads)
        # one inference step on slab column 2 (strided views), recurrent state included
        col = {k: v[:, 2] for k, v in obs.items()}
        st_in = (torch.randn((E, S), generator=torch.Generator().manual_seed(8)) * 0.3).cuda()
        h1 = m.forward_heads(col, E, sample_stride=0, tag="inf", rnn=dict(states=st_in) if nat.rnn_kind is not None else None)[-1]
        new_state = m.new_rnn_states_of("inf").clone() if nat.rnn_kind is not None else None
        outs[name] = (heads, grads, h1.clone(), new_state)
    A1 = 1 + nat.num_action_params
    np.testing.assert_allclose(outs["native"][0][:, :A1].cpu().numpy(), outs["torch"][0][:, :A1].cpu().numpy(), atol=3e-5, rtol=1e-4)
    np.testing.assert_allclose(outs["native"][2][:, :A1].cpu().numpy(), outs["torch"][2][:, :A1].cpu().numpy(), atol=3e-5, rtol=1e-4)
    if nat.rnn_kind is not None:
        np.testing.assert_allclose(outs["native"][3].cpu().numpy(), outs["torch"][3].cpu().numpy(), atol=2e-5, rtol=1e-4)
    for name, _ in shapes:
        a, b = outs["native"][1][name].numpy(), outs["torch"][1][name].numpy()
        assert a.shape == b.shape, name
        scale = max(1e-6, float(np.abs(b).max()))
        assert np.abs(a - b).max() <= 2e-4 * scale + 1e-7, (name, float(np.abs(a - b).max()), scale)
        assert np.abs(b).max() > 0, f"{name}: the torch path's gradient is zero — the check would be vacuous"


def test_separate_weights_with_several_keys_end_to_end(lib, tmp_path):
    """cfg.actor_critic_share_weights=False on an image + vector observation dict: SeparateActorCritic whose two towers are
    multi-key composites (reference: ActorCriticSeparateWeights with a MultiInputEncoder per tower, model/actor_critic.py:198-334
    + model/encoder.py:33-69) — learns the vector-key bandit, names are the reference's, the checkpoint round-trips"""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_dict_obs_bandit_env
    from sample_factory_amd.train import make_runner
    register_env("dict_bandit_sep", make_dict_obs_bandit_env)
    cfg = default_cfg(env="dict_bandit_sep", use_rnn=False, nonlinearity="relu", normalize_input=True,
                      normalize_input_keys=["measurements"], obs_scale=255.0, encoder_conv_architecture="convnet_impala",
                      encoder_conv_mlp_layers=[32], encoder_mlp_layers=[32], rollout=8, batch_size=256,
                      num_batches_per_epoch=2, num_epochs=2, num_workers=1, num_envs_per_worker=1, async_rl=False, seed=4,
                      serial_mode=True, synthetic_num_agents=64, learning_rate=3e-3, gamma=0.0, normalize_returns=False,
                      actor_critic_share_weights=False, train_dir=str(tmp_path), experiment="dict_sep")
    cfg, runner = make_runner(cfg)
    runner.init()
    ac = runner.learner.actor_critic
    assert type(ac).__name__ == "SeparateActorCritic" and type(ac.actor).__name__ == "MultiKeyActorCritic" and ac.multi_key
    names = [n for n, _ in ac.ref_param_shapes()]
    for n in ("actor_encoder.encoders.measurements.mlp_head.0.weight", "critic_encoder.encoders.obs.enc.conv_head.0.weight",
              "critic_encoder.encoders.obs.enc.mlp_layers.0.bias", "critic_linear.weight",
              "action_parameterization.distribution_linear.weight"):
        assert n in names, n
    first = None
    for _ in range(30):
        runner.iteration()
        r = float(runner.traj["rewards"].mean())
        first = r if first is None else first
    assert r > first + 0.3 and r > 0.75, (first, r)
    sd = ac.state_dict()
    assert "obs_normalizer.running_mean_std.running_mean_std.measurements.running_mean" in sd
    assert "obs_normalizer.running_mean_std.running_mean_std.obs.running_mean" not in sd
    runner.learner.save()
    cfg2, runner2 = make_runner(cfg)
    runner2.init()
    assert torch.equal(runner2.learner.actor_critic.flat_params, ac.flat_params)
    ns2 = runner2.learner.actor_critic.state_dict()
    k_ = "obs_normalizer.running_mean_std.running_mean_std.measurements.count"
    assert float(ns2[k_]) == float(sd[k_]) > 64 * 9


def test_native_multi_key_towers_in_async_mode(lib, tmp_path):
    """async_rl with several observation keys on the native towers: inference reads the PUBLISHED snapshot of every tower
    (weights and the per-key normalisation tables), the learner trains the live buffers; the bandit is still learned and a
    checkpoint written mid-run loads into a fresh model bit for bit"""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_dict_obs_bandit_env
    from sample_factory_amd.train import make_runner
    register_env("dict_bandit_async", make_dict_obs_bandit_env)
    cfg = default_cfg(env="dict_bandit_async", use_rnn=False, nonlinearity="relu", normalize_input=True,
                      normalize_input_keys=["measurements"], obs_scale=255.0, encoder_conv_architecture="convnet_impala",
                      encoder_conv_mlp_layers=[32], encoder_mlp_layers=[32], rollout=8, batch_size=256,
                      num_batches_per_epoch=2, num_epochs=1, num_workers=1, num_envs_per_worker=1, async_rl=True,
                      serial_mode=False, seed=4, synthetic_num_agents=64, learning_rate=3e-3, gamma=0.0,
                      normalize_returns=False, train_dir=str(tmp_path), experiment="dict_async")
    cfg, runner = make_runner(cfg)
    runner.init()
    ac = runner.learner.actor_critic
    assert type(ac).__name__ == "MultiKeyActorCritic" and ac._snap is not None
    assert all(t._snap is not None for t in ac.towers)
    first = None
    for _ in range(40):
        runner.iteration()
        r = float(runner.traj["rewards"].mean())
        first = r if first is None else first
    torch.cuda.synchronize()
    assert r > first + 0.3 and r > 0.7, (first, r)
    # the snapshot the sampler reads is a copy of the live parameters as of the last publish, tower by tower
    slot = ac.snap_read
    for t in ac.towers:
        assert t._snap[slot].shape == t.flat_params.shape and torch.isfinite(t._snap[slot]).all()
    runner.learner.save()
    cfg2, runner2 = make_runner(cfg)
    runner2.init()
    assert torch.equal(runner2.learner.actor_critic.flat_params, ac.flat_params)


def test_normalize_input_keys_subset_on_the_native_model(lib):
    """cfg.normalize_input_keys (running_mean_std.py:113-131): running statistics only for the listed keys — a list
    without "obs" leaves the native single-key model without a running normaliser, exactly like normalize_input=False"""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    from sample_factory_amd.model.actor_critic import ActorCritic
    obs_space = spaces.Dict({"obs": spaces.Box(-1, 1, (8,), np.float32)})
    for keys, expect in ((None, True), (["obs"], True), (["measurements"], False)):
        cfg = default_cfg(use_rnn=False, nonlinearity="relu", normalize_input=True, normalize_input_keys=keys,
                          encoder_mlp_layers=[16])
        ac = ActorCritic(cfg, obs_space, spaces.Discrete(4), "cuda")
        assert (ac.obs_normalizer is not None) == expect, keys
        assert ("obs_normalizer.running_mean_std.running_mean_std.obs.running_mean" in ac.state_dict()) == expect


def test_fused_input_normalisation_on_a_frame_view_at_an_odd_address(lib):
    """normalize_input=True on 84x84 frames runs inside conv1's loader, which fetches the bytes as 32-bit words.  A frame view
    whose base is not 4-byte aligned (a custom slab offset) used to fail the launch; it now goes through one aligned u8 copy of
    the batch's frames — same numbers as the aligned view, with and without an index gather"""
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs import spaces
    from sample_factory_amd.model.actor_critic import ActorCritic
    obs_space = spaces.Dict({"obs": spaces.Box(0, 255, (4, 84, 84), np.uint8)})
    cfg = default_cfg(use_rnn=False, nonlinearity="relu", normalize_input=True, obs_scale=255.0,
                      encoder_conv_architecture="convnet_atari", encoder_conv_mlp_layers=[512])
    ac = ActorCritic(cfg, obs_space, spaces.Discrete(6), "cuda")
    assert ac._fused_norm
    n, E = 96, 28224
    g = torch.Generator(device="cuda").manual_seed(3)
    frames = torch.randint(0, 256, (n, E), generator=g, device="cuda", dtype=torch.int32).to(torch.uint8)
    ac.obs_normalizer.update(frames, E, n)
    ref = ac.forward_heads(frames, n, sample_stride=E, tag="inf")[-1].clone()
    raw = torch.empty(n * E + 1, dtype=torch.uint8, device="cuda")
    odd = raw[1:].view(n, E)
    odd.copy_(frames)
    assert odd.data_ptr() % 4 == 1
    got = ac.forward_heads(odd, n, sample_stride=E, tag="inf")[-1]
    assert torch.equal(got, ref)
    idx = torch.randperm(n, device="cuda", generator=g)[:64].to(torch.int32)
    ref_i = ac.forward_heads(frames, 64, sample_stride=E, index=idx, tag="inf")[-1].clone()
    got_i = ac.forward_heads(odd, 64, sample_stride=E, index=idx, tag="inf")[-1]
    assert torch.equal(got_i, ref_i)


@pytest.mark.parametrize("rnn_type", ["lstm", "gru"])
def test_stacked_recurrent_layers_native_rollout_and_fused_passes(lib, rnn_type, monkeypatch):
    """cfg.rnn_num_layers = 2 at rnn_size = 256 on the native model, end to end: the rollout carries [h0 | c0 | h1 | c1] rows
    through the slab (zeroed after a done), both layers take the fused persistent BPTT passes, and the run on the per-step cell
    kernels gives the same weights; the one-step forward equals torch's two-layer nn.LSTM / nn.GRU on the same weights"""
    import sample_factory_amd.model.actor_critic as acm
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_synthetic_continuous_env
    from sample_factory_amd.train import make_runner
    register_env("synthetic_ant", make_synthetic_continuous_env)
    S = 256 * 2 * (2 if rnn_type == "lstm" else 1)

    def run(fused):
        monkeypatch.setattr(acm, "_LSTM_SEQ", fused)
        cfg = default_cfg(env="synthetic_ant", use_rnn=True, rnn_type=rnn_type, rnn_size=256, rnn_num_layers=2, nonlinearity="tanh",
                          normalize_input=True, encoder_mlp_layers=[64, 64], rollout=8, recurrence=8, batch_size=1024,
                          num_batches_per_epoch=2, num_epochs=1, num_workers=1, num_envs_per_worker=1, async_rl=False,
                          seed=3, serial_mode=True, synthetic_num_agents=256, with_vtrace=True, normalize_returns=False,
                          kl_loss_coeff=0.1)
        cfg, runner = make_runner(cfg)
        runner.init()
        ac = runner.learner.actor_critic
        assert isinstance(ac, acm.ActorCritic) and ac.rnn_L == 2 and ac.rnn_S == S
        lib.PROFILE = {}
        try:
            for _ in range(2):
                runner.iteration()
            torch.cuda.synchronize()
            names = [k[-1] for k in lib.PROFILE for _ in lib.PROFILE[k]]
        finally:
            lib.PROFILE = None
        return runner, names

    runner, names = run(True)
    ac = runner.learner.actor_critic
    assert sum(f"k_{rnn_type}_seq_fwd" in n for n in names) >= 2 * 2 and any(f"k_{rnn_type}_seq_bwd" in n for n in names), set(names)
    tr = runner.traj
    assert tr["rnn_states"].shape == (256, 9, S) and torch.isfinite(tr["rnn_states"]).all()
    assert tr["rnn_states"][:, 1:].abs().sum(-1).gt(0).any()
    assert (tr["rnn_states"][:, 1:-1][tr["dones"][:, :-1]].abs().sum() == 0)      # every layer's state zeroed after a done step
    # one inference step against torch's stacked core on the same weights (model/core.py:37-64)
    sd = ac.state_dict()
    core = (torch.nn.LSTM if rnn_type == "lstm" else torch.nn.GRU)(64, 256, 2).cuda().double()
    core.load_state_dict({k[len("core.core."):]: v.double() for k, v in sd.items() if k.startswith("core.core.")})
    n = 64
    feats = torch.randn(n, 64, device="cuda")
    st = torch.randn(n, S, device="cuda") * 0.5
    per = st.double().reshape(n, 2, -1).transpose(0, 1).contiguous()
    if rnn_type == "lstm":
        out_ref, (h, c) = core(feats.double().unsqueeze(0), (per[..., :256].contiguous(), per[..., 256:].contiguous()))
        new_ref = torch.cat((h, c), 2)
    else:
        out_ref, new_ref = core(feats.double().unsqueeze(0), per)
    new_ref = new_ref.transpose(0, 1).reshape(n, -1)
    li0 = [i for i, L in enumerate(ac.layers) if L.role == "rnn_ih"]
    x = feats
    for li in li0:  # the model's own one-step path, layer by layer
        gx = torch.empty((n, ac.layers[li].N), device="cuda")
        ac._gemm(li, x, x.shape[1], None, 0, 0, gx, n, "boot")
        x = ac._rnn_step(li, gx, n, dict(states=st), "boot")
    assert (x.double() - out_ref.squeeze(0)).abs().max() < 2e-5
    assert (ac.new_rnn_states_of("boot").double() - new_ref).abs().max() < 2e-5
    p_fused = ac.flat_params.clone()
    runner2, names2 = run(False)
    assert not any("_seq_" in n for n in names2)
    p_step = runner2.learner.actor_critic.flat_params
    scale = float(p_step.abs().max())
    assert torch.isfinite(p_fused).all() and (p_fused - p_step).abs().max().item() < 1e-4 * scale


@pytest.mark.parametrize("rnn_type", ["lstm", "gru"])
def test_fused_sequence_passes_at_width_256_equal_the_per_step_path(lib, rnn_type, monkeypatch):
    """rnn_size = 256 takes the fused persistent BPTT passes (sf_*_seq_fwd/bwd, instantiated for H in {256, 512}); the
    same run on the per-step cell kernels (the path every other width takes) must give the same weights."""
    import sample_factory_amd.model.actor_critic as acm
    from sample_factory_amd.cfg.arguments import default_cfg
    from sample_factory_amd.envs.env_utils import register_env
    from sample_factory_amd.envs.synthetic import make_synthetic_continuous_env
    from sample_factory_amd.train import make_runner
    register_env("synthetic_ant", make_synthetic_continuous_env)

    def run(fused):
        monkeypatch.setattr(acm, "_LSTM_SEQ", fused)
        cfg = default_cfg(env="synthetic_ant", use_rnn=True, rnn_type=rnn_type, rnn_size=256, nonlinearity="tanh",
                          normalize_input=True, encoder_mlp_layers=[64, 64], rollout=8, recurrence=8, batch_size=1024,
                          num_batches_per_epoch=2, num_epochs=1, num_workers=1, num_envs_per_worker=1, async_rl=False,
                          seed=3, serial_mode=True, synthetic_num_agents=256, with_vtrace=True, normalize_returns=False,
                          kl_loss_coeff=0.1)
        cfg, runner = make_runner(cfg)
        runner.init()
        lib.PROFILE = {}
        try:
            for _ in range(2):
                runner.iteration()
            torch.cuda.synchronize()
            names = {k[-1] for k in lib.PROFILE}
        finally:
            lib.PROFILE = None
        return runner.learner.actor_critic.flat_params.clone(), names

    p_fused, names = run(True)
    assert any(f"k_{rnn_type}_seq_fwd" in n for n in names) and any(f"k_{rnn_type}_seq_bwd" in n for n in names), names
    p_step, names2 = run(False)
    assert not any("_seq_" in n for n in names2)
    assert torch.isfinite(p_fused).all()
    scale = float((p_step).abs().max())
    assert float((p_fused - p_step).abs().max()) < 2e-5 * scale


def test_conv1_relu_sign_bits_between_forward_and_weight_gradient(lib):
    """sf_conv_fwd_relu_mask records one sign bit per output element (u32 per pixel, bit c = channel c > 0) next to the
    activations; sf_conv_wgrad_relu_mask applies them to an UNMASKED output gradient inside the kernel: bit-equal to the
    plain kernels on the same frames / a pre-masked gradient (slab addressing through an index included)."""
    E, T, n = 40, 8, 300
    g = torch.Generator().manual_seed(21)
    slab = torch.randint(0, 256, (E, T + 1, 4, 84, 84), generator=g, dtype=torch.uint8).cuda()
    idx = torch.randperm(E * T, generator=g)[:n].to(torch.int32).cuda()
    inv = float(np.float32(1 / 255.0))
    d = desc(lib, 4, 84, 84, 32, 8, 4, in_u8=1, inv_scale=inv, traj_T=T)
    assert lib.conv_relu_mask_supported(n, d) and not lib.conv_relu_mask_supported(100, d)
    assert not lib.conv_relu_mask_supported(n, desc(lib, 4, 84, 84, 32, 8, 4, in_u8=1, inv_scale=inv, relu=2))
    w = (torch.randn((256, 32), generator=g) / 16).cuda()
    b = (torch.randn(32, generator=g) * 0.3).cuda()
    S = 4 * 84 * 84
    out1, out2 = torch.empty((n * 400, 32), device="cuda"), torch.empty((n * 400, 32), device="cuda")
    mask = torch.full((n * 400,), -1, dtype=torch.int32, device="cuda")
    lib.conv_fwd(slab, S, idx, 0, w, b, out1, n, d)
    lib.conv_fwd_relu_mask(slab, S, idx, 0, w, b, out2, mask, n, d)
    assert torch.equal(out1, out2)
    bits = ((mask.view(-1, 1) >> torch.arange(32, device="cuda", dtype=torch.int32)) & 1).bool()
    assert torch.equal(bits, out1 > 0) and 0.2 < float(bits.float().mean()) < 0.8
    dy = torch.randn((n * 400, 32), generator=g).cuda()
    ws = torch.empty(lib.conv_wgrad_workspace(n, d), dtype=torch.uint8, device="cuda")
    dw1, db1, dw2, db2 = torch.zeros_like(w), torch.zeros(32, device="cuda"), torch.zeros_like(w), torch.zeros(32, device="cuda")
    lib.conv_wgrad(slab, S, idx, 0, (dy * (out1 > 0)).contiguous(), dw1, db1, n, d, ws)
    lib.conv_wgrad_relu_mask(slab, S, idx, 0, dy, mask, dw2, db2, n, d, ws)
    assert torch.equal(dw1, dw2) and torch.equal(db1, db2) and dw1.abs().max() > 0
    with pytest.raises(lib.SfHipError):   # a launch the mask kernels do not take
        lib.conv_fwd_relu_mask(slab, S, idx[:64].contiguous(), 0, w, b, out2, mask, 64, d)


def test_conv1_exact_product_kernels_on_tiny_gradients(lib):
    """Domain of the "every product exact" statement, MEASURED on the matrix pipe (it was asserted before): an f32 splits
    exactly into three bf16 terms while all 24 bits sit above bf16's smallest normal, i.e. for |v| >= 2^-102
    (tests/test_abi_and_host.py restricts its proof to |v| > 1e-30 for that reason).  Output gradients of 2^-95 (every term
    normal): weight gradient exact to accumulation level.  Output gradients of 2^-118 .. 2^-114 (mid / lo terms are bf16
    denormals or below the format): the result must degrade no further than to the hi term alone — relative error of the
    sums <= 2^-7 — whatever the pipe does with denormal operands; the observed figure is written to
    gpurun_out/conv1_tiny_gradients.json."""
    import json
    import os
    n, Cin, H, W, Cout, K, S = 256, 4, 84, 84, 32, 8, 4
    g = torch.Generator().manual_seed(6)
    x = torch.randint(0, 256, (n, Cin, H, W), generator=g, dtype=torch.uint8)
    d = desc(lib, Cin, H, W, Cout, K, S, in_u8=1, sub_mean=0.0, inv_scale=1.0, relu=0)
    assert lib.conv_kernel_name(1, n, d).startswith("k_conv1_wgrad_bf16")
    mant = 1.0 + torch.rand((n, Cout, 20, 20), generator=g)                      # full 24-bit mantissas in [1, 2)
    sign = torch.where(torch.rand((n, Cout, 20, 20), generator=g) < 0.5, -1.0, 1.0)
    ws = torch.empty(lib.conv_wgrad_workspace(n, d), dtype=torch.uint8, device="cuda")
    rep = {}
    for tag, e_lo, e_hi, tol in (("normal_terms_2^-95", -95, -95, 2e-6), ("denormal_terms_2^-118..-114", -118, -114, 2.0 ** -7)):
        expo = torch.randint(e_lo, e_hi + 1, (n, Cout, 20, 20), generator=g).double()
        dy64 = mant.double() * sign.double() * torch.pow(torch.tensor(2.0, dtype=torch.float64), expo)
        dy = dy64.float()
        assert torch.equal(dy.double(), dy64)                                    # representable: normal f32 values
        dyd = dy.permute(0, 2, 3, 1).contiguous().cuda().view(n * 400, Cout)
        dw, db = torch.zeros((Cin * K * K, Cout), device="cuda"), torch.zeros(Cout, device="cuda")
        lib.conv_wgrad(x.cuda(), Cin * H * W, None, 0, dyd, dw, db, n, d, ws)
        cols = F.unfold(x.double(), K, stride=S)                                 # [n, 256, 400]
        ref = torch.einsum("nkp,npc->kc", cols, dy64.permute(0, 2, 3, 1).reshape(n, 400, Cout))
        mag = torch.einsum("nkp,npc->kc", cols, dy64.abs().permute(0, 2, 3, 1).reshape(n, 400, Cout))
        err = float(((dw.cpu().double() - ref).abs() / mag).max())              # relative to sum |x dy| of the element
        rep[tag] = err
        assert torch.isfinite(dw).all() and err <= tol, (tag, err)
    try:
        out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(out, exist_ok=True)
        json.dump(rep, open(os.path.join(out, "conv1_tiny_gradients.json"), "w"), indent=1)
    except OSError:
        pass


@pytest.mark.parametrize("n,K,N,act", [(4096, 512, 8, 0), (2048, 512, 20, 0), (77, 512, 20, 1), (1000, 64, 32, 2),
                                       (33, 272, 5, 0), (5, 1024, 17, 1), (16, 16, 1, 0)])
def test_narrow_linear_forward_vs_torch(lib, n, K, N, act):
    """sf_nn_narrow.h (Cout <= 32: the fused heads matrix; one wave per 16 rows, both operands straight from memory into
    MFMA fragments, 256-deep chunks double-buffered in registers) through sf_conv_fwd_t: ragged last row tile, column
    tiles with missing columns, K not a multiple of the chunk, strided rows, no bias, every activation kind; rows past
    the end are not written; result equal (to rounding) to the tiled kernel's."""
    g = torch.Generator().manual_seed(n + K + N)
    d = desc(lib, K, 1, 1, N, 1, 1, relu=act)
    assert lib.conv_fwd_t_supported(n, d) and lib.conv_fwd_t_workspace(n, d) == 0
    assert lib.conv_kernel_name(3, n, d) == f"k_linear_narrow<{1 if N <= 16 else 2}>"
    x, w, b = torch.randn((n, K), generator=g), torch.randn((K, N), generator=g) / np.sqrt(K), torch.randn(N, generator=g) * 0.1
    f = lambda p: F.relu(p) if act == 1 else torch.tanh(p) if act == 2 else p
    xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
    wt = wd.t().contiguous()
    out = torch.full((n + 3, N), 7.0, device="cuda")
    lib.conv_fwd_t(xd, K, wt, bd, out, n, d)
    ref = f(x.double() @ w.double() + b.double())
    tol = 2e-6 * max(1.0, ref.abs().max().item()) * np.sqrt(K / 64)
    assert (out[:n].cpu().double() - ref).abs().max().item() < tol
    assert (out[n:] == 7.0).all(), "rows past the end must not be written"
    old = torch.empty((n, N), device="cuda")
    lib.conv_fwd(xd, K, None, 0, wd, bd, old, n, d)
    assert (out[:n] - old).abs().max().item() < tol
    big = torch.randn((n, 2, K), generator=g).cuda()  # strided rows (a view into a larger buffer), no bias
    out2 = torch.empty((n, N), device="cuda")
    lib.conv_fwd_t(big[:, 1], 2 * K, wt, None, out2, n, d)
    ref2 = f(big[:, 1].cpu().double() @ w.double())
    assert (out2.cpu().double() - ref2).abs().max().item() < tol


@pytest.mark.parametrize("n,N,K1,K2", [(2048, 2048, 64, 512), (4096, 1024, 32, 256), (2050, 2000, 64, 512), (32768, 64, 96, 32)])
def test_dual_linear_forward_vs_torch(lib, n, N, K1, K2):
    """sf_linear_fwd_dual (k_fwd_glds2: two linear layers into one accumulator, the segment switched per 32-chunk of the
    LDS-DMA pipeline; one LSTM inference step's x W_ih^T + h W_hh^T + b_ih + b_hh): against float64, row-strided second
    operand (a slab column), ragged row / column tiles, NULL biases; rows past the end are not written."""
    g = torch.Generator().manual_seed(n + N + K1)
    a1 = torch.randn((n, K1), generator=g)
    big = torch.randn((n, 3, K2 + 8), generator=g)  # a2 = big[:, 1, :K2]: row stride 3 * (K2 + 8)
    w1, w2 = torch.randn((N, K1), generator=g) / np.sqrt(K1), torch.randn((N, K2), generator=g) / np.sqrt(K2)
    b1, b2 = torch.randn(N, generator=g) * 0.1, torch.randn(N, generator=g) * 0.1
    assert lib.linear_fwd_dual_supported(n, N, K1, K2) and not lib.linear_fwd_dual_supported(n, N, K1 + 8, K2)
    bigd = big.cuda()
    a2v = bigd[:, 1, :K2]
    out = torch.full((n + 2, N), 7.0, device="cuda")
    lib.linear_fwd_dual(a1.cuda(), K1, w1.cuda(), b1.cuda(), a2v, a2v.stride(0), w2.cuda(), b2.cuda(), out, n)
    ref = a1.double() @ w1.double().t() + big[:, 1, :K2].double() @ w2.double().t() + b1.double() + b2.double()
    tol = 3e-6 * max(1.0, ref.abs().max().item())
    assert (out[:n].cpu().double() - ref).abs().max().item() < tol
    assert (out[n:] == 7.0).all(), "rows past the end must not be written"
    out2 = torch.empty((n, N), device="cuda")
    lib.linear_fwd_dual(a1.cuda(), K1, w1.cuda(), None, a2v, a2v.stride(0), w2.cuda(), None, out2, n)
    assert (out2.cpu().double() - (ref - b1.double() - b2.double())).abs().max().item() < tol


@pytest.mark.parametrize("n,H,K1", [(2048, 512, 64), (2050, 256, 32)])
def test_dual_linear_gru_layout_and_cell(lib, n, H, K1):
    """sf_linear_fwd_dual with gru_H = H: [n, 4H] = {r, z pre-activations (x and h parts + both biases), x W_in^T + b_in,
    h W_hn^T + b_hn}, and sf_rnn_cell_fwd(kind 0, gh = NULL) on it equals torch.nn.GRUCell in float64"""
    g = torch.Generator().manual_seed(n + H)
    x, h = torch.randn((n, K1), generator=g), torch.randn((n, H), generator=g) * 0.5
    wih, whh = torch.randn((3 * H, K1), generator=g) / np.sqrt(K1), torch.randn((3 * H, H), generator=g) / np.sqrt(H)
    bih, bhh = torch.randn(3 * H, generator=g) * 0.1, torch.randn(3 * H, generator=g) * 0.1
    assert lib.linear_fwd_dual_supported(n, 4 * H, K1, H)
    pre = torch.full((n, 4 * H), 7.0, device="cuda")
    hd = h.cuda()
    lib.linear_fwd_dual(x.cuda(), K1, wih.cuda(), bih.cuda(), hd, H, whh.cuda(), bhh.cuda(), pre, n, gru_H=H)
    gx, gh = x.double() @ wih.double().t() + bih.double(), h.double() @ whh.double().t() + bhh.double()
    ref = torch.cat([gx[:, :2 * H] + gh[:, :2 * H], gx[:, 2 * H:], gh[:, 2 * H:]], dim=1)
    assert (pre.cpu().double() - ref).abs().max().item() < 3e-6 * max(1.0, ref.abs().max().item())
    h_out = torch.empty((n, H), device="cuda")
    lib.rnn_cell_fwd(0, pre, None, hd, H, None, 0, None, n, H, None, h_out, None, None, None)
    cell = torch.nn.GRUCell(K1, H).double()
    with torch.no_grad():
        cell.weight_ih.copy_(wih.double()); cell.weight_hh.copy_(whh.double()); cell.bias_ih.copy_(bih.double()); cell.bias_hh.copy_(bhh.double())
        want = cell(x.double(), h.double())
    assert (h_out.cpu().double() - want).abs().max().item() < 3e-6


@pytest.mark.parametrize("n,traj_T,use_index", [(1, 0, False), (3, 0, False), (64, 0, True), (255, 5, False), (1000, 8, True),
                                                (2049, 0, False)])
def test_conv1_loader_fused_normalisation_vs_float64(lib, n, traj_T, use_index):
    """sf_conv_fwd_norm / sf_conv_wgrad_norm (cfg.normalize_input on raw u8 frames, normalisation inside conv1's loader)
    against utils/normalize.py:40-70 + running_mean_std.py:79-110 + F.conv2d evaluated in float64: every launch size
    (odd n, n < 256, one sample), dataset -> slab addressing (traj_T), index gather, obs_subtract_mean / obs_scale."""
    import torch.nn.functional as F
    assert lib.load().sf_abi_version() >= 18
    g = torch.Generator().manual_seed(1000 + n)
    C_, H, W, N, KH, S = 4, 84, 84, 32, 8, 4
    d = lib.sf_conv_desc(Cin=C_, H=H, W=W, Cout=N, KH=KH, KW=KH, stride=S, OH=20, OW=20, in_u8=1, relu=1, traj_T=traj_T,
                         sub_mean=3.0, inv_scale=float(np.float32(1.0 / 255.0)))
    assert lib.conv_norm_supported(n, d)
    rows = n + (n // traj_T + 2 if traj_T else 0) + 7
    frames = torch.randint(0, 256, (rows, C_, H, W), dtype=torch.uint8, generator=g).cuda()
    mu = (torch.rand(C_ * H * W, generator=g) * 0.6 + 0.2).cuda()
    rstd = (1.0 / torch.sqrt(torch.rand(C_ * H * W, generator=g) * 0.2 + 1e-3)).cuda()   # some pixels clamp at +-5
    w = (torch.randn(C_ * KH * KH, N, generator=g) / 16).cuda()
    b = (torch.randn(N, generator=g) * 0.1).cuda()
    index = torch.randperm(n, generator=g).int().cuda() if use_index else None
    offset = 0 if use_index else 2
    dsel = index.long() if use_index else torch.arange(offset, offset + n, device="cuda")
    if traj_T:
        dsel = dsel + dsel // traj_T
    x = frames[dsel].double()
    xn = (((x - 3.0) * float(np.float32(1.0 / 255.0)) - mu.double().view(1, C_, H, W)) * rstd.double().view(1, C_, H, W)).clamp(-5, 5)
    assert float((xn.abs() == 5).double().mean()) > 1e-3        # the clamp is exercised
    w4 = w.double().t().reshape(N, C_, KH, KH).clone().requires_grad_(True)     # k = (c*KH + kh)*KW + kw
    b64 = b.double().clone().requires_grad_(True)
    pre = F.conv2d(xn, w4, b64, stride=S)                       # [n, N, 20, 20], before the ReLU
    out = torch.empty((n * 400, N), device="cuda")
    stride = C_ * H * W
    lib.conv_fwd_norm(frames, stride, index, offset, mu, rstd, w, b, out, n, d)
    ref = F.relu(pre).detach().permute(0, 2, 3, 1).reshape(n * 400, N)
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 2e-6, err
    dy = torch.randn((n * 400, N), generator=g).cuda()
    dy_masked = dy * (out > 0)   # the mask of OUR forward, applied by the caller (as sf_conv_dgrad's epilogue does): the
    # float64 gradient below is that of the LINEAR layer against this very dout — no ReLU-flip ambiguity in the comparison
    pre.backward(dy_masked.double().reshape(n, 20, 20, N).permute(0, 3, 1, 2))
    dw, db = torch.empty_like(w), torch.empty_like(b)
    ws = torch.empty(lib.conv_wgrad_workspace(n, d), dtype=torch.uint8, device="cuda")
    lib.conv_wgrad_norm(frames, stride, index, offset, mu, rstd, dy_masked, dw, db, n, d, ws)
    gw = w4.grad.reshape(N, -1).t()
    e_w = (dw.double() - gw).abs().max().item() / gw.abs().max().item()
    e_b = (db.double() - b64.grad).abs().max().item() / b64.grad.abs().max().item()
    assert e_w < 5e-6 and e_b < 5e-6, (e_w, e_b)
    # misuse is reported, not executed
    bad = lib.sf_conv_desc.from_buffer_copy(d)
    bad.H = 80
    assert not lib.conv_norm_supported(n, bad)

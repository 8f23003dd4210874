"""GPU parity of CrossScan/CrossMerge (bit-exact), dwconv+SiLU, SS2D and VSSM against the oracle and
the golden outputs of the reference's own modules."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2, rel_max

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,D,H,W", [(2, 3, 3, 5), (1, 40, 7, 7), (2, 33, 17, 20), (1, 192, 56, 56), (1, 8, 128, 128),
                                     (2, 130, 7, 7),      # small-image tile (128 x 8 x 8), two channel tiles, scalar path
                                     (1, 132, 8, 8),      # small-image tile, vector path
                                     (2, 36, 9, 8),       # just above the small-image limit: 16 x 16 tile with padding
                                     (3, 64, 14, 14)])    # stage-2 shape
def test_cross_scan_and_merge_bit_exact(B, D, H, W):
    from mamba_unet_amd import cross_merge, cross_scan
    from oracle.ss2d_ref import cross_merge_ref, cross_scan_ref
    g = torch.Generator().manual_seed(H * 100 + W)
    x = torch.randn(B, D, H, W, generator=g)
    xs = cross_scan(x.cuda())
    assert torch.equal(xs.cpu(), cross_scan_ref(x))          # pure permutation: bit-exact
    ys = torch.randn(B, 4, D, H * W, generator=g)
    want = cross_merge_ref(ys, H, W)                           # (B, D, L), summed k0 + k2 + k1 + k3
    got_cf = cross_merge(ys.cuda(), H, W, channel_last=False)
    assert torch.equal(got_cf.cpu(), want)                     # same summation order: bit-exact
    got_cl = cross_merge(ys.cuda(), H, W, channel_last=True)
    assert torch.equal(got_cl.cpu(), want.transpose(1, 2).reshape(B, H, W, D))


@pytest.mark.parametrize("H,W", [(3, 5), (7, 7), (56, 56)])
def test_cross_scan_index_tables_bit_exact(H, W):
    """Push an arange image through the kernel: the result IS the src table (int64 equality)."""
    from mamba_unet_amd import cross_scan
    from oracle.ss2d_ref import cross_scan_index
    L = H * W
    x = torch.arange(L, dtype=torch.float32).view(1, 1, H, W).cuda()
    got = cross_scan(x)[0, :, 0].round().to(torch.int64).cpu().numpy()
    assert np.array_equal(got, cross_scan_index(H, W))


def test_cross_ops_gradients_are_adjoints():
    from mamba_unet_amd import cross_merge, cross_scan
    from oracle.ss2d_ref import cross_merge_ref, cross_scan_ref
    x = torch.randn(2, 5, 6, 9)
    g = torch.randn(2, 4, 5, 54)
    xg = x.clone().cuda().requires_grad_(True)
    (cross_scan(xg) * g.cuda()).sum().backward()
    xr = x.clone().requires_grad_(True)
    (cross_scan_ref(xr) * g).sum().backward()
    assert torch.equal(xg.grad.cpu(), xr.grad) or rel_max(xg.grad.cpu(), xr.grad) < 1e-6
    ys = torch.randn(2, 4, 5, 54)
    gy = torch.randn(2, 6, 9, 5)
    yg = ys.clone().cuda().requires_grad_(True)
    (cross_merge(yg, 6, 9, True) * gy.cuda()).sum().backward()
    yr = ys.clone().requires_grad_(True)
    (cross_merge_ref(yr, 6, 9).transpose(1, 2).reshape(2, 6, 9, 5) * gy).sum().backward()
    assert torch.equal(yg.grad.cpu(), yr.grad)


@pytest.mark.parametrize("B,D,H,W,cl", [(2, 5, 4, 6, True), (1, 40, 9, 33, True), (2, 64, 14, 14, False),
                                        (1, 192, 56, 56, True),
                                        (24, 40, 7, 7, True),      # narrow variant (W <= 16), several units per workgroup
                                        (3, 33, 17, 16, True),     # narrow variant, ragged channels and rows
                                        (2, 24, 20, 40, False)])   # wide variant, (B, D, H, W) input
@pytest.mark.parametrize("reps", [None, "3"])
def test_dwconv_silu_fwd_bwd(B, D, H, W, cl, reps, monkeypatch):
    """reps: force 3 (image, row-tile) units per backward workgroup (SS2D_DWCONV_REPS), so the multi-unit walk with its
    ragged last group runs on small shapes too (the default picks > 1 only from ~770 tiles up)."""
    from mamba_unet_amd import dwconv3x3_silu
    if reps is not None:
        monkeypatch.setenv("SS2D_DWCONV_REPS", reps)
    from oracle.ss2d_ref import dwconv3x3_silu_ref
    g = torch.Generator().manual_seed(D)
    wide = torch.randn(B, H, W, 2 * D, generator=g)            # emulates xz; the kernel reads its first half in place
    w = torch.randn(D, 1, 3, 3, generator=g) * 0.3
    b = torch.randn(D, generator=g) * 0.1
    gy = torch.randn(B, D, H, W, generator=g)
    # oracle (fp64 for the gradient sums)
    xr = wide[..., :D].permute(0, 3, 1, 2).double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = dwconv3x3_silu_ref(xr, wr, br)
    (yr * gy.double()).sum().backward()
    # device
    wide_d = wide.cuda().requires_grad_(True)
    wd, bd = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    xin = wide_d[..., :D] if cl else wide_d[..., :D].permute(0, 3, 1, 2).contiguous()
    yd = dwconv3x3_silu(xin, wd, bd, channel_last_in=cl)
    (yd * gy.cuda()).sum().backward()
    assert rel_max(yd.detach().cpu(), yr.detach()) < 1e-5
    assert rel_max(wide_d.grad[..., :D].cpu(), xr.grad.permute(0, 2, 3, 1)) < 1e-5
    assert float(wide_d.grad[..., D:].abs().max()) == 0.0
    assert rel_max(wd.grad.cpu(), wr.grad) < 1e-5 and rel_max(bd.grad.cpu(), br.grad) < 1e-5


@pytest.mark.parametrize("B,D,H,W", [(2, 8, 4, 6), (1, 40, 9, 33), (2, 64, 14, 14), (1, 192, 56, 56), (24, 40, 7, 7),
                                     (3, 36, 17, 16), (2, 24, 20, 40), (2, 100, 31, 61),
                                     # degenerate planes, the narrow / wide switch (W = 14 | 15), the 30-column tile edge
                                     (1, 4, 1, 1), (2, 8, 1, 5), (1, 4, 3, 2), (1, 12, 2, 17), (1, 68, 9, 15), (1, 8, 5, 14),
                                     (1, 8, 16, 30), (1, 8, 8, 31), (1, 8, 9, 60)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("reps", [None, "3"])
def test_dwconv_silu_channel_last_both_sides(B, D, H, W, dtype, reps, monkeypatch):
    """The channel-last convolution (csrc/ss2d_dwconv_cl.hip: float4 of channels per lane, no transposing tile) against
    the fp64 oracle: output, dx (written into the x half of a wide tensor, the other half untouched), dw, dbias; ragged
    channel tiles (D % 32 != 0), ragged column tiles and row bands, several units per workgroup (reps)."""
    from mamba_unet_amd import dwconv3x3_silu
    from mamba_unet_amd.ss2d_ops import _DwConvSiLUCL
    if reps is not None:
        monkeypatch.setenv("SS2D_DWCONV_REPS", reps)
    from oracle.ss2d_ref import dwconv3x3_silu_ref
    g = torch.Generator().manual_seed(D + W)
    wide = torch.randn(B, H, W, 2 * D, generator=g).to(dtype)
    w = torch.randn(D, 1, 3, 3, generator=g) * 0.3
    b = torch.randn(D, generator=g) * 0.1
    gy = torch.randn(B, H, W, D, generator=g).to(dtype)
    xr = wide[..., :D].permute(0, 3, 1, 2).double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    yr = dwconv3x3_silu_ref(xr, wr, br)
    (yr * gy.double().permute(0, 3, 1, 2)).sum().backward()
    wide_d = wide.cuda().requires_grad_(True)
    wd, bd = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    yd = dwconv3x3_silu(wide_d[..., :D], wd, bd, channel_last_in=True, channel_last_out=True)
    assert yd.shape == (B, H, W, D) and yd.dtype == dtype and yd.grad_fn.name().startswith(_DwConvSiLUCL.__name__)
    (yd * gy.cuda()).sum().backward()
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_max(yd.detach().float().cpu().permute(0, 3, 1, 2), yr.detach()) < tol
    assert rel_max(wide_d.grad[..., :D].float().cpu(), xr.grad.permute(0, 2, 3, 1)) < tol
    assert float(wide_d.grad[..., D:].abs().max()) == 0.0
    assert rel_max(wd.grad.cpu(), wr.grad) < tol and rel_max(bd.grad.cpu(), br.grad) < tol


@pytest.mark.parametrize("name", ["ss2d_2x8x5x7_s0.npz", "ss2d_1x16x8x8_s1.npz", "ss2d_2x96x14x14_s0.npz"])
def test_ss2d_module_matches_reference_golden(name):
    """Outputs and ALL gradients of the reference's own SS2D (tests/golden, made by make_golden.py)."""
    from mamba_unet_amd import SS2D
    g = np.load(os.path.join(GOLDEN, name))
    C = g["x"].shape[-1]
    m = SS2D(d_model=C)
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")})
    m = m.cuda()
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    # gates at ~10x the measured errors (tools/parity_margins.py: outputs <= 4.2e-7, dx <= 4.0e-7, parameter gradients <= 8.1e-7)
    core = m.forward_corev0(torch.from_numpy(g["core_in"]).cuda())
    assert rel_max(core.detach().cpu(), g["core_out"]) < 5e-6
    y = m(x)
    assert rel_max(y.detach().cpu(), g["y"]) < 5e-6 and rel_l2(y.detach().cpu(), g["y"]) < 5e-6
    (y * torch.from_numpy(g["g"]).cuda()).sum().backward()
    assert rel_l2(x.grad.cpu(), g["dx"]) < 5e-6
    for k, p in m.named_parameters():
        assert rel_l2(p.grad.cpu(), g["grad." + k]) < 1e-5, (k, rel_l2(p.grad.cpu(), g["grad." + k]))


@pytest.mark.parametrize("switch", ["SS2D_NO_GATHER6", "SS2D_NO_CL_CONV", "SS2D_NO_BLOCK_PARAMS", "SS2D_NO_PACKED_DXDBL"])
def test_ss2d_golden_with_one_fusion_switched_off(switch, monkeypatch):
    """The routes behind the A/B switches stay correct (README): the channel-last convolution output with the plain gathers
    (no 6-slot gather), the (B, D, H, W) convolution, per-parameter transforms, the separate dB / dC packing pass -- each
    against the reference's SS2D fixture, output and every gradient."""
    from mamba_unet_amd import SS2D
    from mamba_unet_amd.selective_scan_interface import ScanCoreFn
    monkeypatch.setenv(switch, "1")
    if switch == "SS2D_NO_PACKED_DXDBL":
        monkeypatch.setattr(ScanCoreFn, "PACKED_DXDBL", False)      # class-wide switch, read from the environment at import
    g = np.load(os.path.join(GOLDEN, "ss2d_2x96x14x14_s0.npz"))
    m = SS2D(d_model=g["x"].shape[-1])
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")})
    m = m.cuda()
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    y = m(x)
    assert rel_max(y.detach().cpu(), g["y"]) < 5e-6
    (y * torch.from_numpy(g["g"]).cuda()).sum().backward()
    assert rel_l2(x.grad.cpu(), g["dx"]) < 5e-6
    for k, p in m.named_parameters():
        assert rel_l2(p.grad.cpu(), g["grad." + k]) < 1e-5, (k, rel_l2(p.grad.cpu(), g["grad." + k]))


def test_ss2d_bf16_autocast_with_fp32_scan_io_matches_the_fixture():
    """bf16 autocast with SS2D.bf16_scan_io off (the reference's casts around the scan, mamba_sys.py:423-426): the channel-last
    convolution output goes back to (B, D, H, W) for the literal composition.  Output and dx against the fp32 fixture at
    the bf16 gate of north_star (1e-2, relative L2)."""
    from mamba_unet_amd import SS2D
    g = np.load(os.path.join(GOLDEN, "ss2d_2x96x14x14_s0.npz"))
    m = SS2D(d_model=g["x"].shape[-1])
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")})
    m = m.cuda()
    old = SS2D.bf16_scan_io
    try:
        SS2D.bf16_scan_io = False
        x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = m(x)
        (y.float() * torch.from_numpy(g["g"]).cuda()).sum().backward()
    finally:
        SS2D.bf16_scan_io = old
    assert rel_l2(y.detach().float().cpu(), g["y"]) < 1e-2
    assert rel_l2(x.grad.float().cpu(), g["dx"]) < 1e-2


def test_vssm_small_logits_match_reference_golden():
    from mamba_unet_amd import VSSM
    g = np.load(os.path.join(GOLDEN, "vssm_small.npz"))
    m = VSSM(patch_size=4, in_chans=3, num_classes=4, depths=[1, 1, 1, 1], dims=[16, 32, 64, 128], drop_path_rate=0.2)
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")})
    m = m.cuda().eval()
    with torch.no_grad():
        y = m(torch.from_numpy(g["x"]).cuda())
    assert rel_max(y.cpu(), g["logits"]) < 1.5e-5 and rel_l2(y.cpu(), g["logits"]) < 1e-5      # measured 1.1e-6 / 5.8e-7


def test_config0_vssm_tiny_224_logits_match_reference_golden():
    """BASELINE.json configs[0]: MambaUNet (VSSM tiny) forward on one 1x1x224x224 slice, 4 classes, against the
    logits the reference itself produced on CPU with its selective_scan_ref path (tests/golden/make_golden.py).
    The weights are not in the fixture: seed 1337 reproduces the reference's initialisation bit for bit."""
    from mamba_unet_amd import MambaUnet
    g = np.load(os.path.join(GOLDEN, "config0_vssm_tiny_224.npz"))
    torch.manual_seed(1337)
    net = MambaUnet(img_size=224, num_classes=4).cuda().eval()
    with torch.no_grad():
        y = net(torch.from_numpy(g["x"]).cuda())          # 1 channel in: MambaUnet repeats it to 3 (vision_mamba.py:43-44)
    assert y.shape == (1, 4, 224, 224)
    assert rel_l2(y.cpu(), g["logits"]) < 1e-5 and rel_max(y.cpu(), g["logits"]) < 1.5e-5      # measured 1.1e-6 / 1.4e-6


def test_mambaunet_train_step_runs():
    """One fully-supervised step (train_fully_supervised_2D_VIM.py:147-160) at a small size: finite loss,
    every parameter receives a gradient."""
    from mamba_unet_amd import MambaUnet, MambaUnetConfig
    torch.manual_seed(1337)
    net = MambaUnet(MambaUnetConfig(depths=[1, 1, 1, 1]), img_size=64, num_classes=4).cuda().train()
    opt = torch.optim.SGD(net.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    x = torch.rand(2, 1, 64, 64, device="cuda")
    lab = torch.randint(0, 4, (2, 64, 64), device="cuda")
    out = net(x)
    loss = torch.nn.functional.cross_entropy(out, lab)
    opt.zero_grad()
    loss.backward()
    opt.step()
    assert torch.isfinite(loss)
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())


@pytest.mark.parametrize("rows,M,N,dtype", [(100, 96, 96, torch.float32), (3137, 384, 96, torch.float32),
                                            (5000, 50, 200, torch.float32), (4096, 96, 192, torch.bfloat16),
                                            (75264, 384, 96, torch.bfloat16), (777, 33, 7, torch.bfloat16)])
def test_linear_wgrad_mfma(rows, M, N, dtype):
    """dW = G^T X on the split-token MFMA kernel against an fp64 product of the same values."""
    from mamba_unet_amd import linear_wgrad
    gen = torch.Generator().manual_seed(rows)
    G = torch.randn(rows, M, generator=gen).to(dtype)
    X = torch.randn(rows, N, generator=gen).to(dtype)
    want = G.double().t() @ X.double()
    got = linear_wgrad(G.cuda(), X.cuda())
    assert got.dtype == torch.float32 and got.shape == (M, N)
    assert rel_l2(got.cpu(), want) < 1e-5          # fp32 products and sums in both storage types
    # strided operands: the two halves of a wider tensor (how dz / dx halves of xz would be read)
    wide = torch.randn(rows, 2 * N, generator=gen).to(dtype).cuda()
    got2 = linear_wgrad(G.cuda(), wide[:, N:])
    assert rel_l2(got2.cpu(), G.double().t() @ wide[:, N:].double().cpu()) < 1e-5


def test_linear_nobias_matches_f_linear():
    from mamba_unet_amd import linear_nobias
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(2, 56, 56, 96, generator=gen).cuda().requires_grad_(True)
    w = (torch.randn(384, 96, generator=gen) * 0.05).cuda().requires_grad_(True)
    g = torch.randn(2, 56, 56, 384, generator=gen).cuda()
    y = linear_nobias(x, w)
    y.backward(g)
    xr, wr = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    yr = torch.nn.functional.linear(xr, wr)
    yr.backward(g.double())
    assert rel_l2(y.detach().cpu(), yr.detach().cpu()) < 1e-5
    assert rel_l2(x.grad.cpu(), xr.grad.cpu()) < 1e-5 and rel_l2(w.grad.cpu(), wr.grad.cpu()) < 1e-5


@pytest.mark.parametrize("rows_shape,D,dtype", [((2, 5, 7), 192, torch.float32), ((1, 14, 14), 768, torch.float32),
                                                 ((3, 7, 7), 1536, torch.float32), ((777,), 20, torch.float32),
                                                 ((2, 28, 28), 384, torch.bfloat16), ((2, 9, 9), 1536, torch.bfloat16)])
def test_outnorm_gate_fused(rows_shape, D, dtype):
    """LayerNorm(D)(y) * SiLU(z) fused fwd/bwd against torch in fp64; z is the second half of a wider tensor."""
    from mamba_unet_amd import outnorm_gate
    gen = torch.Generator().manual_seed(D)
    y = (torch.randn(*rows_shape, D, generator=gen) * 2 + 0.3).to(dtype)
    xz = torch.randn(*rows_shape, 2 * D, generator=gen).to(dtype)
    w = torch.rand(D, generator=gen) + 0.5
    b = torch.randn(D, generator=gen) * 0.1
    go = torch.randn(*rows_shape, D, generator=gen).to(dtype)
    yr, xzr = y.double().requires_grad_(True), xz.double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(yr, (D,), wr, br, 1e-5) * torch.nn.functional.silu(xzr[..., D:])
    ref.backward(go.double())
    yd, xzd = y.cuda().requires_grad_(True), xz.cuda().requires_grad_(True)
    wd, bd = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    out = outnorm_gate(yd, xzd[..., D:], wd, bd, 1e-5)
    assert out.dtype == dtype and out.shape == y.shape
    out.backward(go.cuda())
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_l2(out.detach().float().cpu(), ref.detach()) < tol
    assert rel_l2(yd.grad.float().cpu(), yr.grad) < tol
    assert rel_l2(xzd.grad[..., D:].float().cpu(), xzr.grad[..., D:]) < tol
    assert float(xzd.grad[..., :D].abs().max()) == 0.0
    ptol = 2e-5 if dtype == torch.float32 else 1e-2
    assert rel_l2(wd.grad.cpu(), wr.grad) < ptol and rel_l2(bd.grad.cpu(), br.grad) < ptol


@pytest.mark.parametrize("rows_shape,D,dtype", [((3, 5, 7), 192, torch.float32), ((2, 14, 14), 768, torch.float32), ((4, 9, 9), 384, torch.bfloat16)])
def test_outnorm_gate_with_per_sample_scale(rows_shape, D, dtype):
    """out = LayerNorm(y) * SiLU(z) * scale[b] (the block's stochastic-depth factor mask / keep, mamba_sys.py:581-585, applied in
    front of the bias-free out_proj): forward and all gradients against torch fp64; a dropped sample (factor 0) gets no gradient."""
    from mamba_unet_amd import outnorm_gate
    gen = torch.Generator().manual_seed(D + 11)
    B = rows_shape[0]
    y = (torch.randn(*rows_shape, D, generator=gen) * 2 + 0.3).to(dtype)
    xz = torch.randn(*rows_shape, 2 * D, generator=gen).to(dtype)
    w, b = torch.rand(D, generator=gen) + 0.5, torch.randn(D, generator=gen) * 0.1
    go = torch.randn(*rows_shape, D, generator=gen).to(dtype)
    scale = torch.tensor([0.0] + [1.0 / 0.9] * (B - 1))
    yr, xzr = y.double().requires_grad_(True), xz.double().requires_grad_(True)
    wr, br = w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(yr, (D,), wr, br, 1e-5) * torch.nn.functional.silu(xzr[..., D:]) * scale.double().view(B, 1, 1, 1)
    ref.backward(go.double())
    yd, xzd = y.cuda().requires_grad_(True), xz.cuda().requires_grad_(True)
    wd, bd = w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    out = outnorm_gate(yd, xzd[..., D:], wd, bd, 1e-5, scale=scale.cuda())
    out.backward(go.cuda())
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_l2(out.detach().float().cpu(), ref.detach()) < tol and not out[0].any()
    assert rel_l2(yd.grad.float().cpu(), yr.grad) < tol and not yd.grad[0].any()
    assert rel_l2(xzd.grad[..., D:].float().cpu(), xzr.grad[..., D:]) < tol
    ptol = 2e-5 if dtype == torch.float32 else 1e-2
    assert rel_l2(wd.grad.cpu(), wr.grad) < ptol and rel_l2(bd.grad.cpu(), br.grad) < ptol


@pytest.mark.parametrize("T_shape,N,K,dtype", [((2, 56, 56), 96, 192, torch.bfloat16), ((2, 56, 56), 96, 192, torch.float32),
                                                ((24, 28, 28), 192, 384, torch.bfloat16), ((3, 19, 23), 104, 72, torch.bfloat16),
                                                ((24, 56, 56), 96, 192, torch.float32)])
def test_linear_residual_equals_linear_then_add(T_shape, N, K, dtype):
    """res + x W^T on the token GEMM (ss2d_token_gemm_res: out_proj + the residual connection, mamba_sys.py:562, :585): the
    forward bit for bit what `linear` followed by `res + y` gives (the product is rounded before the sum), the gradients
    those of the two-op form."""
    from mamba_unet_amd.ss2d_ops import linear, linear_residual
    gen = torch.Generator().manual_seed(N + K)
    lin = torch.nn.Linear(K, N, bias=False).cuda().to(dtype)
    x = torch.randn(*T_shape, K, generator=gen).cuda().to(dtype)
    res = torch.randn(*T_shape, N, generator=gen).cuda().to(dtype)
    g = torch.randn(*T_shape, N, generator=gen).cuda().to(dtype)
    outs = []
    for fused in (True, False):
        xi, ri = x.clone().requires_grad_(True), res.clone().requires_grad_(True)
        lin.weight.grad = None
        o = linear_residual(lin, xi, ri) if fused else ri + linear(xi, lin.weight, None)
        o.backward(g)
        outs.append((o.detach(), xi.grad, ri.grad, lin.weight.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    # atomically accumulated in fp32 (order-dependent last bit), then rounded to the weight's dtype
    assert rel_l2(outs[0][3].float().cpu(), outs[1][3].float().cpu()) < (1e-6 if dtype == torch.float32 else 1e-3)
    ref = res.double() + x.double() @ lin.weight.detach().double().t()
    assert rel_l2(outs[0][0].float().cpu(), ref.cpu()) < (1e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("rows_shape,D,dtype,autocast", [((2, 56, 56), 96, torch.float32, False), ((1, 14, 14), 768, torch.float32, False),
                                                          ((3, 7, 7), 1536, torch.float32, False), ((777,), 20, torch.float32, False),
                                                          ((2, 28, 28), 384, torch.bfloat16, False), ((2, 28, 28), 192, torch.float32, True)])
def test_layer_norm_kernel(rows_shape, D, dtype, autocast):
    """HIP LayerNorm fwd/bwd against torch fp64; under bf16 autocast fp32 input -> bf16 output, fp32 dx."""
    from mamba_unet_amd import layer_norm
    gen = torch.Generator().manual_seed(D + 1)
    x = (torch.randn(*rows_shape, D, generator=gen) * 1.7 - 0.2).to(dtype)
    w, b = torch.rand(D, generator=gen) + 0.5, torch.randn(D, generator=gen) * 0.1
    go = torch.randn(*rows_shape, D, generator=gen).to(torch.bfloat16 if autocast else dtype)
    xr, wr, br = x.double().requires_grad_(True), w.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), wr, br, 1e-5)
    ref.backward(go.double())
    xd, wd, bd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True), b.cuda().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        out = layer_norm(xd, wd, bd, 1e-5)
    assert out.dtype == (torch.bfloat16 if autocast else dtype) and xd.dtype == dtype
    out.backward(go.cuda())
    assert xd.grad.dtype == dtype
    tol = 1e-5 if (dtype == torch.float32 and not autocast) else 1e-2
    assert rel_l2(out.detach().float().cpu(), ref.detach()) < tol
    assert rel_l2(xd.grad.float().cpu(), xr.grad) < (1e-5 if dtype == torch.float32 else 1e-2)
    assert rel_l2(wd.grad.cpu(), wr.grad) < 2e-5 and rel_l2(bd.grad.cpu(), br.grad) < 2e-5


@pytest.mark.parametrize("rows_shape,D,dtype,autocast", [((2, 56, 56), 96, torch.float32, False), ((3, 7, 7), 1536, torch.float32, False),
                                                          ((777,), 20, torch.float32, False), ((2, 28, 28), 384, torch.bfloat16, False),
                                                          ((2, 28, 28), 192, torch.float32, True)])
def test_layer_norm_skip_adds_the_residual_gradient(rows_shape, D, dtype, autocast):
    """x + f(LN(x)) (VSSBlock, mamba_sys.py:583-585) with x's residual gradient added inside the LayerNorm backward kernel
    (ss2d_layernorm_bwd_add) instead of by autograd's accumulation: against torch fp64 on `3 * x + LN(x) * m`."""
    from mamba_unet_amd.ss2d_ops import apply_norm_skip
    gen = torch.Generator().manual_seed(D + 7)
    x = (torch.randn(*rows_shape, D, generator=gen) * 1.7 - 0.2).to(dtype)
    m = torch.randn(*rows_shape, D, generator=gen).to(torch.bfloat16 if autocast else dtype)
    go = torch.randn(*rows_shape, D, generator=gen).to(dtype)
    ln = torch.nn.LayerNorm(D)
    with torch.no_grad():
        ln.weight.copy_(torch.rand(D, generator=gen) + 0.5)
        ln.bias.copy_(torch.randn(D, generator=gen) * 0.1)
    lr = torch.nn.LayerNorm(D).double()
    lr.load_state_dict(ln.state_dict())
    xr = x.double().requires_grad_(True)
    ref = 3.0 * xr + lr(xr) * m.double()
    ref.backward(go.double())
    ld, xd = ln.cuda(), x.cuda().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        y, skip = apply_norm_skip(ld, xd)
    assert skip.data_ptr() == xd.data_ptr() and skip.dtype == dtype
    out = 3.0 * skip + (y * m.cuda()).to(dtype)
    out.backward(go.cuda())
    tol = 1e-5 if (dtype == torch.float32 and not autocast) else 1e-2
    assert xd.grad.dtype == dtype and rel_l2(xd.grad.float().cpu(), xr.grad) < tol
    assert rel_l2(ld.weight.grad.cpu(), lr.weight.grad) < (2e-5 if not autocast and dtype == torch.float32 else 1e-2)
    # the pass-through output alone, and the norm output alone
    x2 = x.cuda().requires_grad_(True)
    y2, s2 = apply_norm_skip(ld, x2)
    (s2 * 2.0).sum().backward()
    assert torch.equal(x2.grad.cpu(), torch.full_like(x, 2.0))
    x3 = x.cuda().requires_grad_(True)
    y3, _ = apply_norm_skip(ld, x3)
    y3.backward(go.cuda().to(y3.dtype))
    x4 = x.double().requires_grad_(True)
    lr(x4).backward(go.double())
    assert rel_l2(x3.grad.float().cpu(), x4.grad) < (1e-5 if dtype == torch.float32 else 1e-2)


def test_linear_with_bias_matches_f_linear():
    from mamba_unet_amd.ss2d_ops import linear
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(2, 56, 56, 192, generator=gen).cuda().requires_grad_(True)
    w = (torch.randn(96, 192, generator=gen) * 0.05).cuda().requires_grad_(True)
    b = torch.randn(96, generator=gen).cuda().requires_grad_(True)
    g = torch.randn(2, 56, 56, 96, generator=gen).cuda()
    y = linear(x, w, b)
    y.backward(g)
    xr, wr, br = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yr = torch.nn.functional.linear(xr, wr, br)
    yr.backward(g.double())
    assert rel_l2(y.detach().cpu(), yr.detach().cpu()) < 1e-5
    for a, r in ((x, xr), (w, wr), (b, br)):
        assert rel_l2(a.grad.cpu(), r.grad.cpu()) < 1e-5


@pytest.mark.parametrize("B,C,H,W", [(2, 8, 12, 20), (1, 24, 9, 5), (2, 96, 28, 28)])
def test_fused_directions_equal_literal_path(B, C, H, W):
    """SS2D with the direction permutations folded into the scan (rev_mask / shared u) against the literal
    cross_scan -> scan -> cross_merge path of the same module: output and every gradient."""
    from mamba_unet_amd import SS2D
    torch.manual_seed(C + H)
    m = SS2D(d_model=C).cuda()
    with torch.no_grad():
        m.A_logs.add_(torch.randn_like(m.A_logs) * 0.3)      # direction-specific parameters must stay with their direction
        m.Ds.mul_(torch.rand_like(m.Ds) + 0.5)
    x = torch.randn(B, H, W, C, device="cuda")
    g = torch.randn(B, H, W, C, device="cuda")
    res = {}
    try:
        for flag in (False, True):
            SS2D.fused_directions = flag
            xi = x.clone().requires_grad_(True)
            m.zero_grad(set_to_none=True)
            y = m(xi)
            y.backward(g)
            res[flag] = (y.detach(), xi.grad, {k: p.grad.clone() for k, p in m.named_parameters()})
    finally:
        SS2D.fused_directions = True
    assert rel_l2(res[True][0].cpu(), res[False][0].cpu()) < 1e-5
    assert rel_l2(res[True][1].cpu(), res[False][1].cpu()) < 2e-5
    for k in res[True][2]:
        assert rel_l2(res[True][2][k].cpu(), res[False][2][k].cpu()) < 5e-5, k


@pytest.mark.parametrize("B,C,H,W,dtype", [(2, 768, 7, 7, torch.float32), (3, 768, 7, 7, torch.bfloat16), (2, 192, 7, 7, torch.bfloat16),
                                           (2, 96, 5, 9, torch.float32)])
def test_padded_rows_equal_unpadded(B, C, H, W, dtype, monkeypatch):
    """An odd H*W (the 7 x 7 stage): SS2D pads the rows of the direction side to a multiple of 8 (zero u, zero dy in the pad)
    so that x_proj / dt_proj / the scan keep their aligned kernels -- output and every gradient against the same module
    without the pad (SS2D_NO_ROW_PAD=1: unaligned rows, library GEMMs, the general scan backward)."""
    from mamba_unet_amd import SS2D
    torch.manual_seed(C + H)
    m = SS2D(d_model=C).cuda()
    with torch.no_grad():
        m.A_logs.add_(torch.randn_like(m.A_logs) * 0.3)
        m.Ds.mul_(torch.rand_like(m.Ds) + 0.5)
    x = torch.randn(B, H, W, C, device="cuda")
    g = torch.randn(B, H, W, C, device="cuda")
    res = {}
    old = SS2D.bf16_scan_io
    try:
        SS2D.bf16_scan_io = dtype == torch.bfloat16
        for pad in (False, True):
            if pad:
                monkeypatch.delenv("SS2D_NO_ROW_PAD", raising=False)
            else:
                monkeypatch.setenv("SS2D_NO_ROW_PAD", "1")
            xi = x.clone().requires_grad_(True)
            m.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
                y = m(xi)
            y.float().backward(g)
            res[pad] = (y.detach().float(), xi.grad, {k: p.grad.clone() for k, p in m.named_parameters()})
    finally:
        SS2D.bf16_scan_io = old
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2        # bf16: two roundings of different intermediate tensors
    assert rel_l2(res[True][0].cpu(), res[False][0].cpu()) < tol
    assert rel_l2(res[True][1].cpu(), res[False][1].cpu()) < 2 * tol
    for k in res[True][2]:
        assert torch.isfinite(res[True][2][k]).all(), k
        assert rel_l2(res[True][2][k].cpu(), res[False][2][k].cpu()) < 3 * tol, k


@pytest.mark.parametrize("B,G,D,R,L,dtype", [(2, 4, 192, 6, 3136, torch.float32), (1, 4, 384, 12, 784, torch.float32),
                                             (3, 4, 16, 6, 20, torch.float32), (2, 4, 192, 6, 3136, torch.bfloat16)])
def test_dt_proj_kernel(B, G, D, R, L, dtype):
    """dts = einsum('bgrl,gdr->bgdl') on the streaming kernel, x read in place from the first R rows of x_dbl."""
    from mamba_unet_amd.ss2d_ops import dt_proj
    gen = torch.Generator().manual_seed(L)
    Cx = R + 32
    x_dbl = torch.randn(B, G, Cx, L, generator=gen).to(dtype)
    w = torch.randn(G, D, R, generator=gen) * 0.3
    go = torch.randn(B, G, D, L, generator=gen).to(dtype)
    xr, wr = x_dbl.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = torch.einsum("bgrl,gdr->bgdl", xr[:, :, :R], wr)
    ref.backward(go.double())
    xd, wd = x_dbl.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    out = dt_proj(xd[:, :, :R], wd)
    assert out.dtype == dtype
    out.backward(go.cuda())
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_l2(out.detach().float().cpu(), ref.detach()) < tol
    assert rel_l2(xd.grad[:, :, :R].float().cpu(), xr.grad[:, :, :R]) < tol
    assert rel_l2(wd.grad.cpu(), wr.grad) < (1e-5 if dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("Dm,N,R,regroup", [(192, 16, 6, True), (192, 16, 6, False), (24, 5, 3, True), (1536, 16, 48, True)])
def test_scan_params_kernel(Dm, N, R, regroup):
    """A = -exp(A_logs), Ds, dt_projs_bias, dt_projs_weight (mamba_sys.py:428-431, :420) in one launch each way,
    optionally regrouped to the scan-group order (reference direction k = [0, 2, 1, 3][group]); fp32 tolerance."""
    from mamba_unet_amd import scan_params
    K = 4
    gen = torch.Generator().manual_seed(Dm + N)
    A_logs = torch.log(torch.rand(K * Dm, N, generator=gen) * 15 + 1)
    Ds, bias, w = torch.randn(K * Dm, generator=gen), torch.randn(K, Dm, generator=gen), torch.randn(K, Dm, R, generator=gen)
    grads = [torch.randn(K * Dm, N, generator=gen), torch.randn(K * Dm, generator=gen), torch.randn(K * Dm, generator=gen),
             torch.randn(K, Dm, R, generator=gen)]
    order = [0, 2, 1, 3] if regroup else [0, 1, 2, 3]
    ref_in = [t.double().requires_grad_(True) for t in (A_logs, Ds, bias, w)]
    ref_out = [(-torch.exp(ref_in[0])).view(K, Dm, N)[order].reshape(K * Dm, N), ref_in[1].view(K, Dm)[order].reshape(-1),
               ref_in[2][order].reshape(-1), ref_in[3][order]]
    torch.autograd.backward(ref_out, [g.double() for g in grads])
    dev_in = [t.cuda().requires_grad_(True) for t in (A_logs, Ds, bias, w)]
    dev_out = scan_params(*dev_in, regroup=regroup)
    torch.autograd.backward(list(dev_out), [g.cuda() for g in grads])
    for got, want in zip(dev_out, ref_out):
        assert got.shape == want.shape and rel_l2(got.detach().cpu(), want.detach()) < 1e-6
    for got, want in zip(dev_in, ref_in):
        assert got.grad.shape == want.grad.shape and rel_l2(got.grad.cpu(), want.grad) < 1e-6


@pytest.mark.parametrize("Dm,N,R,io", [(192, 16, 6, torch.bfloat16), (192, 16, 6, torch.float32), (1536, 16, 48, torch.bfloat16),
                                       (20, 16, 3, torch.bfloat16), (20, 16, 3, torch.float32)])
def test_block_params_kernel(Dm, N, R, io):
    """Every parameter-derived operand of a fused-direction SS2D call from one launch (mamba_sys.py:407, :420, :428-431):
    scan_params(regroup=True) plus the stacked x_proj weight, its transpose, dt_w transposed and dt_w, rounded to the
    activations' dtype in zero-padded rows; the backward regroups the stacked weight gradient into x_proj_weight's layout.
    Permutations and casts: bit-exact against the torch composition; A = -exp: fp32 tolerance."""
    from mamba_unet_amd.ss2d_ops import block_params
    K, Cx = 4, R + 2 * N
    gen = torch.Generator().manual_seed(Dm + R)
    A_logs = torch.log(torch.rand(K * Dm, N, generator=gen) * 15 + 1)
    Ds, bias, w = torch.randn(K * Dm, generator=gen), torch.randn(K, Dm, generator=gen), torch.randn(K, Dm, R, generator=gen)
    xw = torch.randn(K, Cx, Dm, generator=gen)
    order = [0, 2, 1, 3]
    dev_in = [t.cuda().requires_grad_(True) for t in (A_logs, Ds, bias, w, xw)]
    A, D, b, dtw, sxw, sxwT, dtwT, dtw_io, holder = block_params(*dev_in, io, want_dtw_io=True)
    assert rel_l2(A.detach().cpu(), (-torch.exp(A_logs.double())).view(K, Dm, N)[order].reshape(K * Dm, N)) < 1e-6
    assert torch.equal(D.detach().cpu(), Ds.view(K, Dm)[order].reshape(-1)) and torch.equal(b.detach().cpu(), bias[order].reshape(-1))
    assert torch.equal(dtw.detach().cpu(), w[order])
    w_cat = xw.view(2, 2, Cx, Dm).transpose(0, 1).reshape(2, 2 * Cx, Dm)          # [orientation][rev*Cx + c]
    assert sxw.dtype == io and torch.equal(sxw.cpu(), w_cat.to(io))
    assert torch.equal(sxwT.cpu(), w_cat.transpose(1, 2).to(io))
    assert torch.equal(dtwT.cpu(), w[order].transpose(1, 2).to(io)) and torch.equal(dtw_io.cpu(), w[order].to(io))
    q = 4 if io == torch.float32 else 8
    for t, k in ((sxw, Dm), (sxwT, 2 * Cx), (dtwT, Dm), (dtw_io, R)):              # rows padded to 16 bytes with zeros
        ld = t.stride(1)
        assert ld % q == 0 and ld >= k and t.data_ptr() % 16 == 0
        full = torch.as_strided(t, (t.shape[0], t.shape[1], ld), (t.stride(0), ld, 1))
        assert not full[:, :, k:].any()
    assert not any(t.requires_grad for t in (sxw, sxwT, dtwT, dtw_io))
    grads = [torch.randn(K * Dm, N, generator=gen), torch.randn(K * Dm, generator=gen), torch.randn(K * Dm, generator=gen),
             torch.randn(K, Dm, R, generator=gen)]
    d_xw = torch.randn(2, 2 * Cx, Dm, generator=gen)
    holder.d_xw = d_xw.cuda()                                                       # what x_proj's backward leaves there
    torch.autograd.backward([A, D, b, dtw], [g.cuda() for g in grads])
    assert holder.d_xw is None
    inv = [0, 2, 1, 3]
    assert rel_l2(dev_in[0].grad.cpu(), (grads[0].double() * (-torch.exp(A_logs.double())).view(K, Dm, N)[order].reshape(K * Dm, N))
                  .view(K, Dm, N)[inv].reshape(K * Dm, N)) < 1e-6
    assert torch.equal(dev_in[1].grad.cpu(), grads[1].view(K, Dm)[inv].reshape(-1))
    assert torch.equal(dev_in[2].grad.cpu(), grads[2].view(K, Dm)[inv])
    assert torch.equal(dev_in[3].grad.cpu(), grads[3][inv])
    assert torch.equal(dev_in[4].grad.cpu(), d_xw.view(2, 2, Cx, Dm).transpose(0, 1).reshape(K, Cx, Dm))


@pytest.mark.parametrize("B,H,W,c,p,dtype", [(2, 5, 7, 24, 2, torch.float32), (1, 14, 14, 96, 4, torch.float32),
                                             (2, 7, 7, 384, 2, torch.float32), (2, 14, 14, 96, 4, torch.bfloat16)])
def test_layer_norm_over_pixel_unshuffle(B, H, W, c, p, dtype):
    """LayerNorm(rearrange(x, 'b h w (p1 p2 c) -> b (h p1) (w p2) c')) (mamba_sys.py:244-247, :262-265) with the
    rearrangement done by the kernel's row addressing: against torch on the rearranged copy, fp64."""
    from mamba_unet_amd.ss2d_ops import apply_norm_unshuffle, pixel_unshuffle_cl
    gen = torch.Generator().manual_seed(c + p)
    x = torch.randn(B, H, W, p * p * c, generator=gen).to(dtype).float()
    ln = torch.nn.LayerNorm(c)
    with torch.no_grad():
        ln.weight.copy_(torch.rand(c, generator=gen) + 0.5)
        ln.bias.copy_(torch.randn(c, generator=gen) * 0.1)
    go = torch.randn(B, H * p, W * p, c, generator=gen).to(dtype).float()
    xr = x.double().requires_grad_(True)
    lr = torch.nn.LayerNorm(c).double()
    lr.load_state_dict(ln.state_dict())
    ref = lr(pixel_unshuffle_cl(xr, p))
    ref.backward(go.double())
    ld = torch.nn.LayerNorm(c).cuda()
    ld.load_state_dict(ln.state_dict())
    xd = x.cuda().to(dtype).requires_grad_(True)
    out = apply_norm_unshuffle(ld, xd, p)
    assert out.shape == (B, H * p, W * p, c) and out.dtype == dtype
    out.backward(go.cuda().to(dtype))
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_l2(out.detach().float().cpu(), ref.detach()) < tol
    assert rel_l2(xd.grad.float().cpu(), xr.grad) < tol
    assert rel_l2(ld.weight.grad.cpu(), lr.weight.grad) < (1e-5 if dtype == torch.float32 else 2e-3)
    assert rel_l2(ld.bias.grad.cpu(), lr.bias.grad) < (1e-5 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("B,H,W,c,p,nc,dtype", [(2, 5, 7, 24, 2, 4, torch.float32), (1, 14, 14, 96, 4, 4, torch.float32),
                                                (2, 6, 5, 96, 4, 9, torch.float32), (1, 4, 4, 128, 4, 16, torch.float32),
                                                (2, 14, 14, 96, 4, 4, torch.bfloat16)])
def test_ln_unshuffle_head_fused(B, H, W, c, p, nc, dtype):
    """1x1 conv (bias-free) of LayerNorm(rearrange(x)) in one pass each way against torch on the materialised
    tensors in fp64 (the tail of VSSM.up_x4, mamba_sys.py:859-866)."""
    from mamba_unet_amd.ss2d_ops import ln_unshuffle_head, pixel_unshuffle_cl
    gen = torch.Generator().manual_seed(c + p + nc)
    x = torch.randn(B, H, W, p * p * c, generator=gen).to(dtype).float()
    ln, conv = torch.nn.LayerNorm(c), torch.nn.Conv2d(c, nc, 1, bias=False)
    with torch.no_grad():
        ln.weight.copy_(torch.rand(c, generator=gen) + 0.5)
        ln.bias.copy_(torch.randn(c, generator=gen) * 0.1)
        conv.weight.copy_(torch.randn(nc, c, 1, 1, generator=gen) * 0.2)
    go = torch.randn(B, nc, H * p, W * p, generator=gen).to(dtype).float()
    xr = x.double().requires_grad_(True)
    lr, cr = torch.nn.LayerNorm(c).double(), torch.nn.Conv2d(c, nc, 1, bias=False).double()
    lr.load_state_dict(ln.state_dict()); cr.load_state_dict(conv.state_dict())
    ref = cr(lr(pixel_unshuffle_cl(xr, p)).permute(0, 3, 1, 2))
    ref.backward(go.double())
    ld, cd = torch.nn.LayerNorm(c).cuda(), torch.nn.Conv2d(c, nc, 1, bias=False).cuda()
    ld.load_state_dict(ln.state_dict()); cd.load_state_dict(conv.state_dict())
    xd = x.cuda().to(dtype).requires_grad_(True)
    out = ln_unshuffle_head(ld, xd, p, cd)
    assert out.shape == (B, nc, H * p, W * p) and out.dtype == dtype
    out.backward(go.cuda().to(dtype))
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    ptol = 1e-5 if dtype == torch.float32 else 3e-3
    assert rel_l2(out.detach().float().cpu(), ref.detach()) < tol
    assert rel_l2(xd.grad.float().cpu(), xr.grad) < tol
    assert rel_l2(ld.weight.grad.cpu(), lr.weight.grad) < ptol and rel_l2(ld.bias.grad.cpu(), lr.bias.grad) < ptol
    assert rel_l2(cd.weight.grad.cpu(), cr.weight.grad) < ptol


def test_graphed_forward_matches_eager():
    """Eval forward replayed from a HIP graph (batch-1 latency path) equals the eager forward bit for bit, also on a
    new input copied into the static buffer."""
    from mamba_unet_amd import MambaUnet
    from mamba_unet_amd.train import GraphedForward
    torch.manual_seed(3)
    net = MambaUnet(img_size=64, num_classes=4).cuda().eval()
    x0, x1 = torch.rand(1, 1, 64, 64, device="cuda"), torch.rand(1, 1, 64, 64, device="cuda")
    g = GraphedForward(net, x0)
    with torch.no_grad():
        want = net(x1)
    assert torch.equal(g(x1), want)


@pytest.mark.parametrize("rows,M,ldx,dtype", [(5000, 96, 96, torch.float32), (75264, 96, 96, torch.bfloat16), (777, 33, 40, torch.float32),
                                              (1, 768, 768, torch.float32), (4704, 384, 384, torch.bfloat16)])
def test_colsum_matches_fp64(rows, M, ldx, dtype):
    from mamba_unet_amd.ss2d_ops import colsum
    gen = torch.Generator().manual_seed(rows + M)
    wide = torch.randn(rows, ldx, generator=gen).to(dtype)
    got = colsum(wide.cuda()[:, :M])
    assert got.dtype == torch.float32 and rel_l2(got.cpu(), wide[:, :M].double().sum(0)) < 1e-5
    assert torch.equal(got, colsum(wide.cuda()[:, :M]))          # deterministic


@pytest.mark.parametrize("T,N,K1,K2,bias", [(3136, 96, 96, 96, True), (777, 192, 192, 192, True), (130, 384, 384, 384, False),
                                            (1000, 104, 64, 40, True), (257, 96, 264, 0, True), (64, 8, 8, 0, True)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_token_gemm_cat(T, N, K1, K2, bias, dtype):
    """[a | a2] @ w.T + bias (the decoder's concat Linear, mamba_sys.py:835, and a plain biased Linear with a2 = None)
    against an fp64 product of the same stored values + the fp32 bias: bf16 -- result rounded once; fp32 (three-way bf16
    split, six MFMAs per product) -- an fp32 GEMM's accuracy."""
    from mamba_unet_amd.ss2d_ops import token_gemm_cat
    gen = torch.Generator().manual_seed(T + N + K1 + K2)
    a = torch.randn(T, K1, generator=gen).to(dtype)
    a2 = torch.randn(T, K2, generator=gen).to(dtype) if K2 else None
    w = (torch.randn(N, K1 + K2, generator=gen) * 0.2).to(dtype)
    b = torch.randn(N, generator=gen) if bias else None
    got = token_gemm_cat(a.cuda(), None if a2 is None else a2.cuda(), w.cuda(), None if b is None else b.cuda())
    assert got is not None and got.shape == (T, N) and got.dtype == dtype
    full = a.double() if a2 is None else torch.cat([a.double(), a2.double()], -1)
    want = full @ w.double().t() + (0 if b is None else b.double())
    if dtype == torch.bfloat16:
        assert rel_l2(got.float().cpu(), want) < 4e-3
        assert (got.float().cpu() - want).abs().max() <= 2.0 ** -8 * want.abs().max() + 1e-6
    else:
        assert rel_l2(got.cpu(), want) < 2e-6
        assert (got.cpu() - want).abs().max() <= 2.0 ** -19 * want.abs().max()
    if K2:   # K1 must be a multiple of the kernel's k chunk: anything else is refused, not computed wrongly
        assert token_gemm_cat(a.cuda()[:, :K1 - 8].contiguous(), a2.cuda(), w.cuda()[:, 8:].contiguous(), None) is None


@pytest.mark.parametrize("B,H,W,C,dtype", [(2, 56, 56, 96, torch.float32), (2, 14, 14, 384, torch.float32), (3, 28, 28, 192, torch.bfloat16),
                                           (3, 56, 56, 96, torch.bfloat16), (2, 14, 14, 384, torch.bfloat16)])
def test_concat_linear_matches_cat_then_linear(B, H, W, C, dtype):
    """The decoder's skip connection (mamba_sys.py:835): Linear(cat([x, skip])) without the concatenated tensor."""
    from mamba_unet_amd.ss2d_ops import concat_linear
    gen = torch.Generator().manual_seed(C)
    lin = torch.nn.Linear(2 * C, C)
    x, skip = torch.randn(B, H, W, C, generator=gen).to(dtype), torch.randn(B, H, W, C, generator=gen).to(dtype)
    g = torch.randn(B, H, W, C, generator=gen).to(dtype)
    ref = torch.nn.Linear(2 * C, C).double()
    ref.load_state_dict({k: v.double() for k, v in lin.state_dict().items()})
    xr, sr = x.double().requires_grad_(True), skip.double().requires_grad_(True)
    yr = ref(torch.cat([xr, sr], -1))
    yr.backward(g.double())
    lin = lin.cuda()
    xd, sd = x.cuda().requires_grad_(True), skip.cuda().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
        y = concat_linear(lin, xd, sd)
    y.backward(g.cuda())
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_l2(y.detach().float().cpu(), yr.detach()) < tol
    assert rel_l2(xd.grad.float().cpu(), xr.grad) < tol and rel_l2(sd.grad.float().cpu(), sr.grad) < tol
    assert rel_l2(lin.weight.grad.cpu(), ref.weight.grad) < tol and rel_l2(lin.bias.grad.cpu(), ref.bias.grad) < tol


@pytest.mark.parametrize("B,G,M,N,L,dtype", [(3, 4, 192, 6, 3136, torch.float32), (2, 2, 76, 192, 784, torch.float32),
                                             (24, 4, 192, 6, 3136, torch.bfloat16), (24, 2, 76, 192, 3136, torch.bfloat16),
                                             (5, 4, 768, 24, 196, torch.bfloat16), (4, 2, 112, 768, 196, torch.float32),
                                             (2, 1, 33, 100, 20, torch.float32), (3, 2, 160, 1536, 64, torch.bfloat16)])
def test_chan_wgrad_mfma(B, G, M, N, L, dtype):
    """out[g] = sum_b P[b, g] Q[b, g]^T for channel-major operands (the dt_proj / x_proj weight gradients), operands taken
    as row ranges of wider blocks like the dts rows of x_dbl, against an fp64 einsum of the same stored values."""
    from mamba_unet_amd.ss2d_ops import chan_wgrad
    gen = torch.Generator().manual_seed(M + N + L)
    Pw = torch.randn(B, G, M + 8, L, generator=gen).to(dtype)
    Qw = torch.randn(B, G, N + 4, L, generator=gen).to(dtype)
    Pd, Qd = Pw.cuda()[:, :, 8:], Qw.cuda()[:, :, :N]
    got = chan_wgrad(Pd, Qd)
    assert got is not None and got.shape == (G, M, N) and got.dtype == torch.float32
    want = torch.einsum("bgml,bgnl->gmn", Pw[:, :, 8:].double(), Qw[:, :, :N].double())
    assert rel_l2(got.cpu(), want) < 1e-5
    assert chan_wgrad(Pd[..., :L - 1], Qd[..., :L - 1]) is None        # rows no longer L apart: the caller falls back


@pytest.mark.parametrize("B,G,M,K,L", [(3, 2, 76, 192, 3136), (2, 2, 192, 76, 784), (2, 4, 6, 192, 784), (3, 4, 768, 24, 196),
                                       (2, 2, 112, 768, 196), (1, 1, 97, 65, 132), (2, 3, 33, 7, 12), (2, 4, 48, 1536, 52)])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_chan_gemm_mfma(B, G, M, K, L, dtype):
    """Y[b, g] = W[g] @ X[b, g] for a channel-major X taken as a row range of a wider block (like the dts rows of x_dbl):
    x_proj / dt_proj and their input gradients, against an fp64 product of the same stored values.  fp32: three-way bf16
    split on the MFMA, the accuracy of an fp32 GEMM; 1, 2 or 3 row tiles per workgroup by M."""
    from mamba_unet_amd.ss2d_ops import chan_gemm
    gen = torch.Generator().manual_seed(M * 7 + K + L)
    W = (torch.randn(G, M, K, generator=gen) * 0.2).to(dtype)
    Xw = torch.randn(B, G, K + 8, L, generator=gen).to(dtype)
    Xd = Xw.cuda()[:, :, 4:4 + K]
    guard = torch.full((B * G * M + 2, L), 7.0, dtype=dtype, device="cuda")             # nothing may be written around the result
    got = chan_gemm(W.cuda(), Xd)
    assert got is not None and got.shape == (B, G, M, L) and got.dtype == dtype
    want = torch.matmul(W.double().unsqueeze(0), Xw[:, :, 4:4 + K].double())
    if dtype == torch.bfloat16:
        assert rel_l2(got.float().cpu(), want) < 4e-3                 # bf16 rounding of the result only
        assert (got.float().cpu() - want).abs().max() <= 2.0 ** -8 * want.abs().max() + 1e-6
    else:
        assert rel_l2(got.cpu(), want) < 2e-6
        assert (got.cpu() - want).abs().max() <= 2.0 ** -19 * want.abs().max()
    assert bool((guard == 7.0).all())
    assert chan_gemm(W.cuda().half(), Xd.half()) is None                # other dtypes: library


@pytest.mark.parametrize("T,N,K", [(3136, 384, 96), (3136, 96, 192), (777, 192, 384), (130, 1536, 768), (49, 768, 1536), (1, 8, 8),
                                   (1000, 104, 40), (257, 96, 264)])
@pytest.mark.parametrize("k_major", [False, True])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_token_gemm_mfma(T, N, K, k_major, dtype):
    """The token-major Linear forward (a @ w.T) and input gradient (a @ w) of in_proj / out_proj / merge / expand against an
    fp64 product of the same stored values: whole and ragged token tiles, column tiles and k chunks, both chunk sizes, a row
    range of a wider buffer as the token operand.  bf16: the result is rounded once.  fp32: every value enters the bf16 MFMA
    as the exact sum of three bf16 values, six products kept -- the accuracy of an fp32 GEMM."""
    from mamba_unet_amd.ss2d_ops import token_gemm
    gen = torch.Generator().manual_seed(T + 3 * N + 5 * K + int(k_major))
    a_wide = torch.randn(T, K + 16, generator=gen).to(dtype)
    w = (torch.randn(*((K, N) if k_major else (N, K)), generator=gen) * 0.2).to(dtype)
    a = a_wide.cuda()[:, 8:8 + K]
    guard = torch.full((T + 2, N), 7.0, dtype=dtype, device="cuda")                     # nothing may be written around the result
    got = token_gemm(a, w.cuda(), k_major)
    assert got is not None and got.shape == (T, N) and got.dtype == dtype
    want = a_wide[:, 8:8 + K].double() @ (w.double() if k_major else w.double().t())
    if dtype == torch.bfloat16:
        assert rel_l2(got.float().cpu(), want) < 4e-3                 # bf16 rounding of the result only
        assert (got.float().cpu() - want).abs().max() <= 2.0 ** -8 * want.abs().max() + 1e-6
    else:
        assert rel_l2(got.cpu(), want) < 2e-6
        assert (got.cpu() - want).abs().max() <= 2.0 ** -19 * want.abs().max()
    assert bool((guard == 7.0).all())
    assert token_gemm(a, w.cuda().to(torch.float16), k_major) is None   # mixed / other dtypes: library
    a3 = a.reshape(1, T, K)                                             # leading batch axes are flattened
    got3 = token_gemm(a3, w.cuda(), k_major)
    assert got3.shape == (1, T, N) and torch.equal(got3[0], got)


@pytest.mark.parametrize("scale", [1e-30, 1.0, 1e30])
def test_token_gemm_fp32_split_is_scale_free(scale):
    """The three-way split is by bits, not by magnitude: the same product at 1e-30, 1 and 1e30 times the operands (results
    1e-30 ... 1e30 around 1) keeps the fp32-GEMM accuracy; zeros stay zeros."""
    from mamba_unet_amd.ss2d_ops import token_gemm
    gen = torch.Generator().manual_seed(5)
    a = torch.randn(300, 64, generator=gen) * scale
    a[:, 7] = 0.0
    w = torch.randn(40, 64, generator=gen) / scale
    got = token_gemm(a.cuda(), w.cuda(), False)
    assert rel_l2(got.cpu(), a.double() @ w.double().t()) < 2e-6
    z = token_gemm(torch.zeros(300, 64, device="cuda"), w.cuda(), False)
    assert bool((z == 0).all())


def test_linear_function_bf16():
    """ss2d_ops.linear (forward on the token GEMM, dX on its k-major mode, dW on the split-token kernel) against fp64
    autograd of the same bf16 values, and bit-identical output with the library switch."""
    from mamba_unet_amd import ss2d_ops
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(2, 28, 28, 192, generator=gen).to(torch.bfloat16)
    w = (torch.randn(768, 192, generator=gen) * 0.1).to(torch.bfloat16)
    g = torch.randn(2, 28, 28, 768, generator=gen).to(torch.bfloat16)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    (xr @ wr.t()).backward(g.double())
    xd, wd = x.cuda().requires_grad_(True), w.cuda().requires_grad_(True)
    y = ss2d_ops._Linear.apply(xd, wd, None)
    y.backward(g.cuda())
    assert rel_l2(y.detach().float().cpu(), (x.double() @ w.double().t())) < 4e-3
    assert rel_l2(xd.grad.float().cpu(), xr.grad) < 4e-3 and rel_l2(wd.grad.float().cpu(), wr.grad) < 4e-3


@pytest.mark.parametrize("bias_dtype,autocast", [(torch.bfloat16, False), (torch.float32, True), (torch.float32, False)])
def test_linear_function_keeps_its_bias(bias_dtype, autocast):
    """A biased Linear through ss2d_ops.linear: a pure-bf16 module (model.bfloat16(), no autocast: bf16 bias) must not lose
    its bias on the way to the own kernel (whose bias input is fp32); fp32 master parameters under autocast take the own
    kernel with the bias; an fp32 module without autocast takes the fp32 kernel or the library by shape.  All against F.linear
    in fp64."""
    from mamba_unet_amd import ss2d_ops
    gen = torch.Generator().manual_seed(12)
    pdt = bias_dtype
    x = torch.randn(2, 28, 28, 96, generator=gen).to(torch.bfloat16 if (autocast or pdt == torch.bfloat16) else torch.float32)
    w = (torch.randn(192, 96, generator=gen) * 0.1).to(pdt)
    b = (torch.randn(192, generator=gen) * 3.0).to(pdt)               # large against x W^T: a dropped bias cannot hide
    g = torch.randn(2, 28, 28, 192, generator=gen).to(x.dtype)
    xd, wd, bd = (t.cuda().requires_grad_(True) for t in (x, w, b))
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
        y = ss2d_ops.linear(xd, wd, bd)
    y.backward(g.cuda().to(y.dtype))
    wq = w.to(torch.bfloat16) if autocast else w
    want = x.double() @ wq.double().t() + b.double()
    tol = 1e-5 if y.dtype == torch.float32 else 6e-3
    assert rel_l2(y.detach().float().cpu(), want) < tol
    assert bd.grad is not None and rel_l2(bd.grad.float().cpu(), g.double().sum((0, 1, 2))) < tol
    assert rel_l2(xd.grad.float().cpu(), g.double() @ wq.double()) < tol


@pytest.mark.parametrize("dtype,L", [(torch.float32, 784), (torch.bfloat16, 784), (torch.bfloat16, 196), (torch.bfloat16, 49)])
def test_x_proj_function(dtype, L):
    """x_proj forward, dX and dW (MFMA kernels in bf16 where the layout allows, library otherwise) against fp64 autograd of
    the same stored values."""
    from mamba_unet_amd.ss2d_ops import x_proj
    gen = torch.Generator().manual_seed(5 + L)
    w = (torch.randn(2, 76, 192, generator=gen) * 0.1).to(dtype)
    xs = torch.randn(3, 2, 192, L, generator=gen).to(dtype)
    g = torch.randn(3, 2, 76, L, generator=gen).to(dtype)
    wr, xr = w.double().requires_grad_(True), xs.double().requires_grad_(True)
    yr = torch.matmul(wr.unsqueeze(0), xr)
    yr.backward(g.double())
    wd, xd = w.cuda().requires_grad_(True), xs.cuda().requires_grad_(True)
    y = x_proj(wd, xd)
    y.backward(g.cuda())
    tol = 1e-5 if dtype == torch.float32 else 4e-3
    assert y.dtype == dtype and rel_l2(y.detach().float().cpu(), yr.detach()) < tol
    assert rel_l2(wd.grad.float().cpu(), wr.grad) < tol and rel_l2(xd.grad.float().cpu(), xr.grad) < tol


@pytest.mark.parametrize("B,H,W,C,dtype", [(2, 56, 56, 96, torch.float32), (3, 28, 28, 192, torch.bfloat16), (2, 14, 14, 384, torch.float32),
                                            (1, 6, 10, 8, torch.float32), (2, 7, 14, 96, torch.float32)])
def test_layernorm_merge_gather(B, H, W, C, dtype):
    """PatchMerging2D's 2x2 gather + LayerNorm(4C) by the kernel's addressing (mamba_sys.py:213-221) against the literal
    slices + cat + fp64 LayerNorm, forward, dx, dgamma, dbeta; an odd size takes the literal path of the reference."""
    from mamba_unet_amd.ss2d_ops import apply_norm_merge
    gen = torch.Generator().manual_seed(H * W + C)
    ln = torch.nn.LayerNorm(4 * C)
    with torch.no_grad():
        ln.weight.copy_(torch.randn(4 * C, generator=gen)); ln.bias.copy_(torch.randn(4 * C, generator=gen))
    x = torch.randn(B, H, W, C, generator=gen).to(dtype)
    h2, w2 = H // 2, W // 2
    g = torch.randn(B, h2, w2, 4 * C, generator=gen).to(dtype)
    ref = torch.nn.LayerNorm(4 * C).double()
    ref.load_state_dict({k: v.double() for k, v in ln.state_dict().items()})
    xr = x.double().requires_grad_(True)
    parts = [xr[:, i::2, j::2, :][:, :h2, :w2, :] for (i, j) in ((0, 0), (1, 0), (0, 1), (1, 1))]
    yr = ref(torch.cat(parts, -1))
    yr.backward(g.double())
    ln = ln.cuda()
    xd = x.cuda().requires_grad_(True)
    y = apply_norm_merge(ln, xd)
    y.backward(g.cuda())
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert y.shape == (B, h2, w2, 4 * C)
    assert rel_l2(y.detach().float().cpu(), yr.detach()) < tol and rel_l2(xd.grad.float().cpu(), xr.grad) < tol
    assert rel_l2(ln.weight.grad.cpu(), ref.weight.grad) < tol and rel_l2(ln.bias.grad.cpu(), ref.bias.grad) < tol


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_patch_embed_as_linear_equals_conv(dtype, monkeypatch):
    """PatchEmbed2D (mamba_sys.py:166-188): the stride-4 4x4 convolution as a Linear over gathered patches against the
    convolution itself (SS2D_PATCH_CONV=1): output, weight and bias gradients."""
    from mamba_unet_amd.vssm import PatchEmbed2D
    torch.manual_seed(3)
    m = PatchEmbed2D(4, 3, 96, torch.nn.LayerNorm).cuda()
    x = torch.randn(6, 3, 224, 224, device="cuda")
    g = torch.randn(6, 56, 56, 96, device="cuda")
    res = {}
    for conv in (True, False):
        if conv:
            monkeypatch.setenv("SS2D_PATCH_CONV", "1")
        else:
            monkeypatch.delenv("SS2D_PATCH_CONV", raising=False)
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
            y = m(x)
        y.float().backward(g)
        res[conv] = (y.detach().float(), m.proj.weight.grad.clone(), m.proj.bias.grad.clone(), m.norm.weight.grad.clone())
    tol = 2e-5 if dtype == torch.float32 else 1.5e-2
    for a, b in zip(res[False], res[True]):
        assert a.shape == b.shape and rel_l2(a.cpu(), b.cpu()) < tol

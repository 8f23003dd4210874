import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mamba_unet_amd import MambaUnet, SS2D
import mamba_unet_amd as M
from mamba_unet_amd.ss2d_ops import dirs_scatter2, dt_proj, scan_params, apply_norm
torch.manual_seed(0)

def check(name, fn, *inputs):
    """fn(*inputs) eager vs captured-and-replayed (inputs static); forward only"""
    ref = fn(*inputs)
    ref = [r.clone() for r in (ref if isinstance(ref, (tuple, list)) else [ref])]
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(*inputs)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn(*inputs)
    out = list(out if isinstance(out, (tuple, list)) else [out])
    errs = []
    for k in range(3):
        for o in out: o.fill_(float("nan")) if o.is_floating_point() else None
        g.replay(); torch.cuda.synchronize()
        errs.append(max(float((o.float() - r.float()).abs().max()) if torch.isfinite(o.float()).all() else float("inf") for o, r in zip(out, ref)))
    print(f"{name:28s} max abs diff over 3 replays: {errs}", flush=True)

B, C, H = 24, 96, 56
D = 2 * C
with torch.no_grad():
    x = torch.randn(B, H, H, C, device="cuda")
    ss = SS2D(d_model=C).cuda()
    xz = torch.randn(B, H, H, 2 * D, device="cuda")
    check("dwconv3x3_silu", lambda: M.dwconv3x3_silu(xz[..., :D], ss.conv2d.weight, ss.conv2d.bias, channel_last_in=True))
    xc = torch.randn(B, D, H, H, device="cuda")
    check("dirs_scatter2", lambda: dirs_scatter2(xc))
    check("cross_scan", lambda: M.cross_scan(xc))
    ys = torch.randn(B, 4, D, H * H, device="cuda")
    check("cross_merge", lambda: M.cross_merge(ys, H, H, channel_last=True))
    check("scan_params", lambda: scan_params(ss.A_logs, ss.Ds, ss.dt_projs_bias, ss.dt_projs_weight, regroup=True))
    xdbl = torch.randn(B, 4, 38, H * H, device="cuda")
    A, Dv, bias, dtw = scan_params(ss.A_logs, ss.Ds, ss.dt_projs_bias, ss.dt_projs_weight, regroup=True)
    check("dt_proj", lambda: dt_proj(xdbl[:, :, :6], dtw))
    xs2 = dirs_scatter2(xc).reshape(B, 2 * D, H * H)
    check("scan_core_fn", lambda: M.scan_core_fn(xs2, xdbl, dtw, A, Dv, bias, H, H))
    delta = torch.rand(B, 4 * D, H * H, device="cuda") * 0.1
    Bm, Cm = torch.randn(B, 4, 16, H * H, device="cuda"), torch.randn(B, 4, 16, H * H, device="cuda")
    u4 = torch.randn(B, 4 * D, H * H, device="cuda")
    check("selective_scan_fn", lambda: M.selective_scan_fn(u4, delta, A, Bm, Cm, Dv, None, bias, True))
    y = torch.randn(B, H, H, D, device="cuda")
    check("outnorm_gate", lambda: M.outnorm_gate(y, xz[..., D:], ss.out_norm.weight, ss.out_norm.bias))
    ln = torch.nn.LayerNorm(C).cuda()
    check("layer_norm", lambda: apply_norm(ln, x))
    check("SS2D.forward", lambda: ss(x))
    net = MambaUnet(img_size=224, num_classes=4).cuda().eval()
    img = torch.rand(B, 1, 224, 224, device="cuda")
    check("MambaUnet.forward (eval)", lambda: net(img))

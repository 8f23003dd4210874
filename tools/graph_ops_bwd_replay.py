import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mamba_unet_amd import MambaUnet, SS2D
import mamba_unet_amd as M
from mamba_unet_amd.ss2d_ops import dirs_scatter2, dt_proj, scan_params, apply_norm, linear
torch.manual_seed(0)

def check(name, fn, inputs):
    """grads of sum(fn(*inputs) * w) w.r.t. inputs: eager vs captured-and-replayed"""
    def run():
        out = fn(*inputs)
        outs = out if isinstance(out, (tuple, list)) else [out]
        loss = sum((o.float() * 0.37).sum() for o in outs)
        return [o.detach() for o in outs] + list(torch.autograd.grad(loss, inputs, allow_unused=True))
    ref = [None if r is None else r.clone() for r in run()]
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = run()
    errs = []
    for k in range(3):
        for o in out:
            if o is not None: o.fill_(float("nan"))
        g.replay(); torch.cuda.synchronize()
        e = []
        for o, r in zip(out, ref):
            if o is None: continue
            d = (o.float() - r.float()).abs().max() / (r.float().abs().max() + 1e-20)
            e.append(float(d) if torch.isfinite(o.float()).all() else float("inf"))
        errs.append(["%.1e" % v for v in e])
    print(f"{name:24s} rel diffs [outs..., grads...] per replay: {errs}", flush=True)

B, C, H = 24, 96, 56
D = 2 * C
rg = lambda t: t.requires_grad_(True)
ss = SS2D(d_model=C).cuda()
xz = rg(torch.randn(B, H, H, 2 * D, device="cuda"))
check("dwconv3x3_silu", lambda a, w, b: M.dwconv3x3_silu(a[..., :D], w, b, channel_last_in=True), [xz, ss.conv2d.weight, ss.conv2d.bias])
xc = rg(torch.randn(B, D, H, H, device="cuda"))
check("dirs_scatter2", lambda a: dirs_scatter2(a), [xc])
check("scan_params", lambda a, b, c, d: scan_params(a, b, c, d, regroup=True), [ss.A_logs, ss.Ds, ss.dt_projs_bias, ss.dt_projs_weight])
xdbl = rg(torch.randn(B, 4, 38, H * H, device="cuda"))
A, Dv, bias, dtw = [rg(t.detach().clone()) for t in scan_params(ss.A_logs, ss.Ds, ss.dt_projs_bias, ss.dt_projs_weight, regroup=True)]
xs2 = dirs_scatter2(xc).reshape(B, 2 * D, H * H).detach()
check("scan_core_fn", lambda xd, w, a, d, b, xx: M.scan_core_fn(xs2, xd, w, a, d, b, H, H, x=xx), [xdbl, dtw, A, Dv, bias, xc])
y = rg(torch.randn(B, H, H, D, device="cuda"))
check("outnorm_gate", lambda a, z, w, b: M.outnorm_gate(a, z[..., D:], w, b), [y, xz, ss.out_norm.weight, ss.out_norm.bias])
ln = torch.nn.LayerNorm(C).cuda()
x = rg(torch.randn(B, H, H, C, device="cuda"))
check("layer_norm", lambda a, w, b: M.layer_norm(a, w, b), [x, ln.weight, ln.bias])
check("linear(in_proj)", lambda a, w: linear(a, w, None), [x, ss.in_proj.weight])
check("SS2D.forward", lambda a: ss(a), [x] + list(ss.parameters()))

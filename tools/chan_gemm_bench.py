#!/usr/bin/env python3
"""x_proj / dt_proj products on the package's MFMA kernel (ss2d_chan_gemm) beside the library's batched GEMM, kernel
times from a rocprofv3 trace.  On the GPU box:
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/cg -o cg -- python3 $REPO/tools/chan_gemm_bench.py
    python3 $REPO/tools/chan_gemm_bench.py --parse $REPO/gpurun_out/cg
The run interleaves (own, library) for the four products at stages 0-2, six times; --parse prints the last repetition.
    python3 tools/chan_gemm_bench.py --events [fp32]
times the same products with HIP events instead (whole calls: the library's copies included) and prints the relative L2
error of both against fp64."""
import sys, os
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    import csv, glob
    f = glob.glob(sys.argv[2] + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    seq = [(r['Kernel_Name'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows if 'cm_gemm' in r['Kernel_Name'] or r['Kernel_Name'].startswith('Cijk')]
    per = 8
    names = ['xproj fwd', 'xproj dX', 'dt d_dts', 'dt fwd']
    for s in range(len(seq) // (6 * per)):
        base = s * 6 * per + 5 * per
        print('stage', s, '  '.join('%s %.1f (lib %.1f)' % (names[i], seq[base + 2 * i][1], seq[base + 2 * i + 1][1]) for i in range(4)))
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mamba_unet_amd.ss2d_ops import chan_gemm, chan_wgrad
dt = torch.float32 if sys.argv[-1] == "fp32" else torch.bfloat16
EVENTS = len(sys.argv) > 1 and sys.argv[1] == "--events"


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def rel(y, w_, x_):
    ref = torch.matmul(w_[None, :1].double(), x_[:2, :1].double())
    return float((y[:2, :1].double() - ref).norm() / ref.norm())


for (C, H) in ((96, 56), (192, 28), (384, 14)):
    B, D, L, R = 24, 2 * C, H * H, (C + 15) // 16
    Cx = R + 32
    xs2 = torch.randn(B, 2, D, L, device='cuda', dtype=dt); w = torch.randn(2, 2 * Cx, D, device='cuda', dtype=dt)
    g = torch.randn(B, 2, 2 * Cx, L, device='cuda', dtype=dt); wt = w.transpose(1, 2).contiguous()
    dd = torch.randn(B, 4, D, L, device='cuda', dtype=dt); wd = torch.randn(4, D, R, device='cuda', dtype=dt); wdt = wd.transpose(1, 2).contiguous()
    xd = torch.randn(B, 4, Cx, L, device='cuda', dtype=dt)
    if EVENTS:
        cases = [("xproj fwd", w, xs2), ("xproj dX", wt, g), ("dt d_dts", wdt, dd), ("dt fwd", wd, xd[:, :, :R])]
        print("stage C=%d L=%d  " % (C, L) + "  ".join("%s %.1f (lib %.1f) err %.1e (lib %.1e)" % (
            n, timed(lambda: chan_gemm(w_, x_)), timed(lambda: torch.matmul(w_.unsqueeze(0), x_)),
            rel(chan_gemm(w_, x_), w_, x_), rel(torch.matmul(w_.unsqueeze(0), x_), w_, x_)) for n, w_, x_ in cases))
        def lib_w(P, Q):
            return torch.matmul(P, Q.transpose(2, 3)).sum(0, dtype=torch.float32)

        def relw(y, P, Q):
            ref = torch.einsum("bml,bnl->mn", P[:, 0].double(), Q[:, 0].double())
            return float((y[0].double() - ref).norm() / ref.norm())
        wc = [("xproj dW", g, xs2), ("dt dW", dd, xd[:, :, :R])]
        print("                   " + "  ".join("%s %.1f (lib %.1f) err %.1e (lib %.1e)" % (
            n, timed(lambda: chan_wgrad(P, Q)), timed(lambda: lib_w(P, Q)), relw(chan_wgrad(P, Q), P, Q), relw(lib_w(P, Q), P, Q))
            for n, P, Q in wc))
        continue
    for it in range(6):
        a = chan_gemm(w, xs2); b_ = torch.matmul(w.unsqueeze(0), xs2)
        c = chan_gemm(wt, g); d = torch.matmul(wt.unsqueeze(0), g)
        e = chan_gemm(wdt, dd); f = torch.matmul(wdt.unsqueeze(0), dd)
        h = chan_gemm(wd, xd[:, :, :R]); i = torch.matmul(wd.unsqueeze(0), xd[:, :, :R])
    torch.cuda.synchronize()
    print(C, L, float((a.float() - b_.float()).abs().max()), float((c.float() - d.float()).abs().max()), float((e.float() - f.float()).abs().max()), float((h.float() - i.float()).abs().max()))

#!/usr/bin/env python3
"""The weight-gradient kernel (ss2d_linear_wgrad, bf16) at the Linear shapes of MambaUNet, batch 24: run under
rocprofv3 --kernel-trace --output-format csv -d DIR -o w, then --parse DIR prints time and bytes/time per shape.
    python3 tools/wgrad_shapes_bench.py --events [fp32]
times whole calls with HIP events beside the library's g^T x, with the relative L2 error of both against fp64."""
import os, sys
SHAPES = []
for s in range(4):
    T, C = 24 * (56 >> s) ** 2, 96 << s
    SHAPES += [("s%d in_proj" % s, T, 4 * C, C), ("s%d out_proj" % s, T, C, 2 * C)]
for s in range(3):
    T, C = 24 * (28 >> s) ** 2, 96 << s
    SHAPES += [("merge%d 4C->2C" % s, T, 2 * C, 4 * C), ("expand%d 2C->4C" % s, T, 4 * C, 2 * C)]
SHAPES += [("final expand C->16C", 24 * 56 * 56, 1536, 96)]
REPS = 5
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    import csv, glob
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    seq = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "wgrad" in r["Kernel_Name"]]
    for i, (name, T, M, N) in enumerate(SHAPES):
        t = seq[i * REPS + REPS - 1]
        mb = T * (M + N) * 2 / 1e6
        print("%-22s T %6d M %5d N %5d  %7.1f us  %6.1f MB  %5.2f TB/s  %6.1f TFLOP/s" % (name, T, M, N, t, mb, mb / t, 2.0 * T * M * N / t / 1e6))
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mamba_unet_amd.ss2d_ops import linear_wgrad  # noqa: E402
if len(sys.argv) > 1 and sys.argv[1] == "--events":
    dt = torch.float32 if sys.argv[-1] == "fp32" else torch.bfloat16

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
    for name, T, M, N in SHAPES:
        g = torch.randn(T, M, device="cuda").to(dt); x = torch.randn(T, N, device="cuda").to(dt)
        ref = g.double().t() @ x.double()
        rel = lambda y: float((y.double() - ref).norm() / ref.norm())  # noqa: E731
        print("%-22s T %6d M %5d N %5d  own %7.1f us  lib %7.1f us   err own %.1e lib %.1e" % (
            name, T, M, N, timed(lambda: linear_wgrad(g, x)), timed(lambda: g.t() @ x), rel(linear_wgrad(g, x)), rel(g.t() @ x)))
    sys.exit(0)
for name, T, M, N in SHAPES:
    g = torch.randn(T, M, device="cuda").bfloat16(); x = torch.randn(T, N, device="cuda").bfloat16()
    for _ in range(REPS):
        linear_wgrad(g, x)
torch.cuda.synchronize()

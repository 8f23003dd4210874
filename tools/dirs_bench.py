#!/usr/bin/env python3
"""The five direction scatter / gather passes of one SS2D block (forward + backward) at the MambaUNet stage shapes, bf16,
batch 24: run under `rocprofv3 --kernel-trace --output-format csv -d DIR -o d -- python3 tools/dirs_bench.py`, then
`python3 tools/dirs_bench.py --parse DIR` prints kernel time and achieved bandwidth per pass."""
import os
import sys

PASSES = [("scatter x(B,D,H,W) -> xs(B,2,D,L)", 3), ("gather out(B,4,D,L) -> y(B,H,W,D)", 5), ("scatter g(B,H,W,D) -> dys(B,2,D,L)", 3),
          ("gather du(B,4,D,L) -> dx(B,D,H,W)", 5), ("gather dxs(B,2,D,L) -> dx(B,D,H,W)", 3)]
STAGES = [(192, 56), (384, 28), (768, 14), (1536, 7)]
REPS = 6
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    import csv, glob
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    seq = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "cross_" in r["Kernel_Name"]]
    i = 0
    for D, H in STAGES:
        base = i + (REPS - 1) * len(PASSES)
        for j, (name, planes) in enumerate(PASSES):
            mb = 24 * D * H * H * 2 * planes / 1e6
            print("D %4d H %2d  %-36s %7.1f us  %6.0f MB  %5.2f TB/s" % (D, H, name, seq[base + j], mb, mb / seq[base + j]))
        i += REPS * len(PASSES)
    sys.exit(0)

import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mamba_unet_amd import _lib  # noqa: E402
lib, P = _lib.load(), _lib.ptr
st = None
B = 24
for D, H in STAGES:
    L = H * H
    x = torch.randn(B, D, H, H, device="cuda").bfloat16(); xs = torch.empty(B, 2, D, L, device="cuda", dtype=torch.bfloat16)
    out = torch.randn(B, 4, D, L, device="cuda").bfloat16(); y = torch.empty(B, H, H, D, device="cuda", dtype=torch.bfloat16)
    import ctypes
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(REPS):
        _lib.check(lib.ss2d_dirs_scatter(1, 2, P(x), P(xs), B, D, H, H, 0, s), "a")
        _lib.check(lib.ss2d_dirs_gather(1, 1, P(out), P(y), B, D, H, H, 1, s), "b")
        _lib.check(lib.ss2d_dirs_scatter(1, 2, P(y), P(xs), B, D, H, H, 1, s), "c")
        _lib.check(lib.ss2d_dirs_gather(1, 1, P(out), P(x), B, D, H, H, 0, s), "d")
        _lib.check(lib.ss2d_dirs_gather(1, 2, P(xs), P(x), B, D, H, H, 0, s), "e")
    torch.cuda.synchronize()

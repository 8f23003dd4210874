#!/usr/bin/env python3
"""Where the per-call floor of ss2d_linear_wgrad (bf16) comes from: kernel time against the number of 64-token chunks a
workgroup walks, at a fixed tile count.  Run under rocprofv3 --kernel-trace --output-format csv -d DIR -o w, then
--parse DIR.  (A linear fit t = c0 + c1 * chunks separates the constant part from the chunk loop.)"""
import os, sys
M, N = 768, 1536                      # 6 x 16 = 96 tiles -> 3 slabs per tile at 256 CUs
TS = [192, 384, 768, 1176, 2352, 4704, 9408, 18816]
REPS = 5
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    import csv, glob
    f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    seq = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "wgrad" in r["Kernel_Name"]]
    for i, T in enumerate(TS):
        ks = min(max((256 + 95) // 96, 1), (T + 255) // 256)
        slab = (T + ks - 1) // ks
        print("T %6d  slabs %d  chunks per workgroup %3d  %7.1f us" % (T, ks, (slab + 63) // 64, min(seq[i * REPS:(i + 1) * REPS])))
    sys.exit(0)
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mamba_unet_amd.ss2d_ops import linear_wgrad  # noqa: E402
for T in TS:
    g = torch.randn(T, M, device="cuda").bfloat16(); x = torch.randn(T, N, device="cuda").bfloat16()
    for _ in range(REPS):
        linear_wgrad(g, x)
torch.cuda.synchronize()

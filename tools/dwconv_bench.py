#!/usr/bin/env python3
"""dwconv3x3 + SiLU forward / backward and the column-sum kernel at the MambaUNet stage shapes (bf16, batch 24, the
channel-last x half of xz as input, channel-last output): run under rocprofv3 --kernel-trace, read the per-kernel
averages from the stats (the kernel trace has the four stages in order, 5 calls each)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mamba_unet_amd.ss2d_ops import dwconv3x3_silu, colsum  # noqa: E402
B = 24
CL_OUT = not os.environ.get("DWCONV_BENCH_TILE")      # the channel-last kernels SS2D uses; set the variable for the tile kernels
for D, H in ((192, 56), (384, 28), (768, 14), (1536, 7)):
    xz = torch.randn(B, H, H, 2 * D, device="cuda").bfloat16().requires_grad_(True)
    w = torch.randn(D, 1, 3, 3, device="cuda", requires_grad=True); b = torch.randn(D, device="cuda", requires_grad=True)
    for _ in range(5):
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = dwconv3x3_silu(xz[..., :D], w, b, True, channel_last_out=CL_OUT)
        y.backward(torch.randn_like(y))
    g = torch.randn(B * H * H, D // 2, device="cuda").bfloat16()
    for _ in range(5):
        colsum(g)
torch.cuda.synchronize()

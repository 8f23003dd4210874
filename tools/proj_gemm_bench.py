#!/usr/bin/env python3
"""MFMA-side report for the four projections of one SS2D call at the stage
shapes of MambaUNet 224x224, batch 24: in_proj, x_proj, dt_proj, out_proj, forward and weight/input gradients,
exactly as mamba_unet_amd.ss2d.SS2D issues them (F.linear / broadcast torch.matmul -> hipBLASLt / rocBLAS).

    python tools/proj_gemm_bench.py [--dtype fp32|bf16] [--stage 0..3] [--iters 20]
Prints one line per GEMM: time, TFLOP/s, fraction of the dense MFMA peak for the dtype
(MI355X_MICROARCH.md: bf16 2500 TFLOP/s, fp32 matrix 157.3 TFLOP/s), and the bytes/time HBM fraction
(these GEMMs are skinny: M or N is 38/6/96, so most are bandwidth-bound).
Run under `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` for the MfmaUtil counters.
"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PEAK = {"fp32": 157.3e12, "bf16": 2500e12}
STAGES = [(96, 56), (192, 28), (384, 14), (768, 7)]


def timeit(f, n):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", choices=["fp32", "bf16"], default="fp32")
    ap.add_argument("--stage", type=int, default=0)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dt = torch.float32 if a.dtype == "fp32" else torch.bfloat16
    es = 4 if a.dtype == "fp32" else 2
    C, H = STAGES[a.stage]
    B, K, N = 24, 4, 16
    D, R, L = 2 * C, (C + 15) // 16, H * H
    Cx = R + 2 * N
    dev = "cuda"
    x = torch.randn(B, H, H, C, device=dev, dtype=dt)
    w_in = torch.randn(2 * D, C, device=dev, dtype=dt)
    xs = torch.randn(B, K, D, L, device=dev, dtype=dt)
    w_x = torch.randn(K, Cx, D, device=dev, dtype=dt)
    dts = torch.randn(B, K, R, L, device=dev, dtype=dt)
    w_dt = torch.randn(K, D, R, device=dev, dtype=dt)
    y = torch.randn(B, H, H, D, device=dev, dtype=dt)
    w_out = torch.randn(C, D, device=dev, dtype=dt)
    g_xz = torch.randn(B, H, H, 2 * D, device=dev, dtype=dt)
    g_xdbl = torch.randn(B, K, Cx, L, device=dev, dtype=dt)
    g_dts = torch.randn(B, K, D, L, device=dev, dtype=dt)
    g_out = torch.randn(B, H, H, C, device=dev, dtype=dt)
    M = B * L
    cases = [
        # name, callable, flops, bytes moved (operands + result once)
        ("in_proj fwd", lambda: F.linear(x, w_in), 2 * M * C * 2 * D, es * (M * C + 2 * D * C + M * 2 * D)),
        ("in_proj dX", lambda: g_xz @ w_in, 2 * M * C * 2 * D, es * (M * 2 * D + 2 * D * C + M * C)),
        ("in_proj dW", lambda: g_xz.reshape(M, 2 * D).t() @ x.reshape(M, C), 2 * M * C * 2 * D, es * (M * 2 * D + M * C + 2 * D * C)),
        ("x_proj fwd", lambda: torch.matmul(w_x.unsqueeze(0), xs), 2 * B * K * Cx * D * L, es * (B * K * D * L + K * Cx * D + B * K * Cx * L)),
        ("x_proj dX", lambda: torch.matmul(w_x.transpose(1, 2).unsqueeze(0), g_xdbl), 2 * B * K * Cx * D * L, es * (B * K * Cx * L + K * Cx * D + B * K * D * L)),
        ("x_proj dW", lambda: torch.matmul(g_xdbl, xs.transpose(2, 3)).sum(0), 2 * B * K * Cx * D * L, es * (B * K * Cx * L + B * K * D * L + B * K * Cx * D)),
        ("dt_proj fwd", lambda: torch.matmul(w_dt.unsqueeze(0), dts), 2 * B * K * D * R * L, es * (B * K * R * L + K * D * R + B * K * D * L)),
        ("dt_proj dX", lambda: torch.matmul(w_dt.transpose(1, 2).unsqueeze(0), g_dts), 2 * B * K * D * R * L, es * (B * K * D * L + K * D * R + B * K * R * L)),
        ("dt_proj dW", lambda: torch.matmul(g_dts, dts.transpose(2, 3)).sum(0), 2 * B * K * D * R * L, es * (B * K * D * L + B * K * R * L + B * K * D * R)),
        ("out_proj fwd", lambda: F.linear(y, w_out), 2 * M * D * C, es * (M * D + C * D + M * C)),
        ("out_proj dX", lambda: g_out @ w_out, 2 * M * D * C, es * (M * C + C * D + M * D)),
        ("out_proj dW", lambda: g_out.reshape(M, C).t() @ y.reshape(M, D), 2 * M * D * C, es * (M * C + M * D + C * D)),
    ]
    rows, tot_t, tot_f = [], 0.0, 0.0
    for name, f, flops, nbytes in cases:
        t = timeit(f, a.iters)
        tot_t += t
        tot_f += flops
        rows.append({"gemm": name, "us": round(t * 1e6, 1), "GFLOP": round(flops / 1e9, 2), "TFLOPs": round(flops / t / 1e12, 1),
                     "frac_mfma_peak": round(flops / t / PEAK[a.dtype], 4), "GBs": round(nbytes / t / 1e9, 0),
                     "frac_hbm_peak": round(nbytes / t / 8e12, 3)})
        print(json.dumps(rows[-1]))
    print(json.dumps({"stage": a.stage, "dtype": a.dtype, "C": C, "D": D, "L": L, "R": R, "total_us": round(tot_t * 1e6, 1),
                      "total_GFLOP": round(tot_f / 1e9, 1), "avg_frac_mfma_peak": round(tot_f / tot_t / PEAK[a.dtype], 4)}))


if __name__ == "__main__":
    main()

"""Token-major Linear forward / input gradient at the MambaUNet shapes (batch 24, 224^2): ss2d_token_gemm against the BLAS
library (F.linear / matmul), microseconds per call by HIP events over 50 back-to-back calls.  Run on the GPU box:
    python tools/token_gemm_bench.py
"""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from mamba_unet_amd.ss2d_ops import token_gemm


def timed(fn, reps=50):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


shapes = []
for s in range(4):
    T, C = 24 * (56 >> s) ** 2, 96 << s
    shapes += [("s%d in_proj" % s, T, 4 * C, C), ("s%d out_proj" % s, T, C, 2 * C)]
shapes += [("final expand", 24 * 56 * 56, 1536, 96), ("merge s0", 24 * 28 * 28, 192, 384), ("expand s3", 24 * 7 * 7, 1536, 768)]
print("%-14s %7s %5s %5s | fwd lib / own (us) | dX lib / own (us) | ideal fwd us @6 TB/s" % ("shape", "T", "N", "K"))
for name, T, N, K in shapes:
    x = torch.randn(T, K, device="cuda").bfloat16()
    w = (torch.randn(N, K, device="cuda") * 0.1).bfloat16()
    g = torch.randn(T, N, device="cuda").bfloat16()
    lf = timed(lambda: torch.nn.functional.linear(x, w))
    of = timed(lambda: token_gemm(x, w, False))
    lb = timed(lambda: torch.matmul(g, w))
    ob = timed(lambda: token_gemm(g, w, True))
    ideal = 2.0 * (T * K + T * N + N * K) / 6e12 * 1e6
    print("%-14s %7d %5d %5d | %7.1f / %7.1f | %7.1f / %7.1f | %6.1f" % (name, T, N, K, lf, of, lb, ob, ideal))

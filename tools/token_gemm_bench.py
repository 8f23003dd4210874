"""Token-major Linear forward / input gradient at the MambaUNet shapes (batch 24, 224^2): ss2d_token_gemm against the BLAS
library (F.linear / matmul), microseconds per call by HIP events over 50 back-to-back calls.  Run on the GPU box:
    python tools/token_gemm_bench.py [fp32]
fp32: the three-way bf16 split kernels against the library's fp32 GEMM, with the relative L2 error of both against fp64.
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


F32 = len(sys.argv) > 1 and sys.argv[1] == "fp32"
DT = torch.float32 if F32 else torch.bfloat16


def rel(y, ref):
    return float((y.double() - ref).norm() / ref.norm())


shapes = []
for s in range(4):
    T, C = 24 * (56 >> s) ** 2, 96 << s
    shapes += [("s%d in_proj" % s, T, 4 * C, C), ("s%d out_proj" % s, T, C, 2 * C)]
shapes += [("final expand", 24 * 56 * 56, 1536, 96), ("merge s0", 24 * 28 * 28, 192, 384), ("expand s3", 24 * 7 * 7, 1536, 768)]
print("%-14s %7s %5s %5s | fwd lib / own (us) | dX lib / own (us) | ideal fwd us @6 TB/s" % ("shape", "T", "N", "K"))
for name, T, N, K in shapes:
    x = torch.randn(T, K, device="cuda").to(DT)
    w = (torch.randn(N, K, device="cuda") * 0.1).to(DT)
    g = torch.randn(T, N, device="cuda").to(DT)
    lf = timed(lambda: torch.nn.functional.linear(x, w))
    of = timed(lambda: token_gemm(x, w, False))
    lb = timed(lambda: torch.matmul(g, w))
    ob = timed(lambda: token_gemm(g, w, True))
    ideal = x.element_size() * (T * K + T * N + N * K) / 6e12 * 1e6
    line = "%-14s %7d %5d %5d | %7.1f / %7.1f | %7.1f / %7.1f | %6.1f" % (name, T, N, K, lf, of, lb, ob, ideal)
    if F32:
        n = min(T, 4096)
        ref = x[:n].double() @ w.double().t()
        refb = g[:n].double() @ w.double()
        line += " | err fwd lib %.1e own %.1e, dX lib %.1e own %.1e" % (
            rel(torch.nn.functional.linear(x[:n], w), ref), rel(token_gemm(x[:n], w, False), ref),
            rel(torch.matmul(g[:n], w), refb), rel(token_gemm(g[:n], w, True), refb))
    print(line)

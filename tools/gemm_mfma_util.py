#!/usr/bin/env python3
"""MFMA utilisation of the package's own GEMM kernels (x_proj / dt_proj / in_proj / out_proj and their gradients:
tok_gemm_kernel, cm_gemm_kernel, cm_wgrad_kernel, wgrad_bf16_tr_kernel; reference ops mamba_sys.py:407, :420, :555, :562)
at the MambaUNet shapes, batch 24, bf16 -- or, with a trailing `fp32` argument, the fp32 forms of the same kernels (every
value a three-way bf16 split, six MFMAs per product: tok_gemm_f32_kernel, cm_gemm_f32_kernel, cm_wgrad_kernel<float>,
wgrad_f32_tr_kernel).  The workload is run twice under rocprofv3 by tools/gemm_mfma_util.sh -- once with
--kernel-trace (durations), once with --pmc (SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_VALU_MFMA_MOPS_BF16, SQ_BUSY_CYCLES,
GRBM_GUI_ACTIVE) -- and `--parse TRACE_DIR PMC_DIR` prints one line per (kernel, shape).

  MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)     (MI355X_MICROARCH.md: the counter is in
  cycles, 32 per v_mfma_f32_32x32x16_bf16; GRBM_GUI_ACTIVE is summed over the XCDs)
  TFLOP/s and TB/s are the shape's algorithmic FLOPs / bytes over the traced duration."""
import os
import sys

REPS = 6          # per shape; the parser takes the last repetition of every launch
B = 24
FP32 = "fp32" in sys.argv[1:]
ES = 4.0 if FP32 else 2.0      # bytes per operand element
WG = "wgrad_f32" if FP32 else "wgrad_bf16"


def workload():
    """[(label, kernel-name substring, flops, bytes, thunk factory)] in launch order."""
    items = []
    for s in range(4):
        T, C = B * (56 >> s) ** 2, 96 << s
        for nm, N, K in (("in_proj", 4 * C, C), ("out_proj", C, 2 * C)):
            items.append(("s%d %s fwd" % (s, nm), "tok_gemm", ("tok", T, N, K, False)))
            items.append(("s%d %s dX" % (s, nm), "tok_gemm", ("tok", T, N, K, True)))
            items.append(("s%d %s dW" % (s, nm), WG, ("wgrad", T, N, K)))
    items.append(("final expand fwd", "tok_gemm", ("tok", B * 56 * 56, 1536, 96, False)))
    items.append(("final expand dW", WG, ("wgrad", B * 56 * 56, 1536, 96)))
    for s in range(3):
        C, H = 96 << s, 56 >> s
        D, L, R = 2 * C, H * H, (C + 15) // 16
        Cx = R + 32
        items.append(("s%d x_proj fwd" % s, "cm_gemm", ("cg", 2, 2 * Cx, D, L)))            # W (2, 2Cx, D) @ xs2 (B, 2, D, L)
        items.append(("s%d x_proj dX" % s, "cm_gemm", ("cg", 2, D, 2 * Cx, L)))
        items.append(("s%d x_proj dW" % s, "cm_wgrad", ("cw", 2, 2 * Cx, D, L)))
        items.append(("s%d dt_proj dW" % s, "cm_wgrad", ("cw", 4, D, R, L)))
        if s == 2:
            items.append(("s%d dt_proj fwd" % s, "cm_gemm", ("cg", 4, D, R, L)))
    return items


def flops_bytes(spec):
    kind = spec[0]
    if kind == "tok":
        _, T, N, K, _ = spec
        return 2.0 * T * N * K, ES * (T * K + T * N + N * K)
    if kind == "wgrad":
        _, T, M, N = spec
        return 2.0 * T * M * N, ES * T * (M + N) + 4.0 * M * N
    if kind == "cg":
        _, G, M, K, L = spec
        return 2.0 * B * G * M * K * L, ES * B * G * L * (M + K)
    _, G, M, N, L = spec
    return 2.0 * B * G * M * N * L, ES * B * G * L * (M + N) + 4.0 * G * M * N


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from mamba_unet_amd.ss2d_ops import chan_gemm, chan_wgrad, linear_wgrad, token_gemm
    dt = torch.float32 if FP32 else torch.bfloat16
    for label, _, spec in workload():
        kind = spec[0]
        if kind == "tok":
            _, T, N, K, kmaj = spec
            a = torch.randn(T, N if kmaj else K, device="cuda").to(dt)
            w = (torch.randn(N, K, device="cuda") * 0.1).to(dt)
            fn = lambda: token_gemm(a, w, kmaj)      # noqa: E731
        elif kind == "wgrad":
            _, T, M, N = spec
            g, x = torch.randn(T, M, device="cuda").to(dt), torch.randn(T, N, device="cuda").to(dt)
            fn = lambda: linear_wgrad(g, x)          # noqa: E731
        elif kind == "cg":
            _, G, M, K, L = spec
            w, x = (torch.randn(G, M, K, device="cuda") * 0.1).to(dt), torch.randn(B, G, K, L, device="cuda").to(dt)
            fn = lambda: chan_gemm(w, x)             # noqa: E731
        else:
            _, G, M, N, L = spec
            p, q = torch.randn(B, G, M, L, device="cuda").to(dt), torch.randn(B, G, N, L, device="cuda").to(dt)
            fn = lambda: chan_wgrad(p, q)            # noqa: E731
        for _ in range(REPS):
            if fn() is None:
                raise SystemExit("%s: the own kernel declined this layout" % label)
        torch.cuda.synchronize()


def parse(trace_dir, pmc_dir):
    import csv
    import glob
    own = ("tok_gemm", WG, "cm_gemm", "cm_wgrad")
    tr = sorted(csv.DictReader(open(glob.glob(trace_dir + "/**/*kernel_trace.csv", recursive=True)[0])), key=lambda r: int(r["Start_Timestamp"]))
    durs = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in tr if any(k in r["Kernel_Name"] for k in own)]
    pm = {}
    for f in glob.glob(pmc_dir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if any(k in r["Kernel_Name"] for k in own):
                pm.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
    disp = [pm[k] for k in sorted(pm)]
    items = workload()
    if len(durs) != len(items) * REPS or len(disp) != len(items) * REPS:
        raise SystemExit("launch counts do not match the workload: %d traced, %d counted, %d expected" % (len(durs), len(disp), len(items) * REPS))
    if FP32:
        print("fp32 operands: TFLOP/s counts the product once (2 T N K); the MFMA executes six bf16 products per fp32 product")
    print("%-20s %-22s %8s %9s %7s %9s %10s" % ("product", "kernel", "us", "TFLOP/s", "TB/s", "MFMA busy", "of 2.5 PF"))
    for i, (label, kname, spec) in enumerate(items):
        name, us = durs[i * REPS + REPS - 1]
        c = disp[i * REPS + REPS - 1]
        assert kname in name and kname in c["name"], (label, name, c["name"])
        fl, by = flops_bytes(spec)
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)
        print("%-20s %-22s %8.1f %9.1f %7.2f %8.1f%% %9.2f%%" % (label, next(k for k in ("tok_gemm_f32_kernel", "tok_gemm_kernel", "wgrad_f32_tr_kernel", "wgrad_bf16_tr_kernel", "wgrad_bf16_kernel", "cm_gemm_f32_kernel", "cm_gemm_kernel", "cm_wgrad_kernel", "?") if k in name or k == "?"), us, fl / us / 1e6, by / us / 1e6,
                                                          100 * busy, 100 * fl / us / 1e6 / 2500.0))


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[1] == "--parse":
        parse(sys.argv[2], sys.argv[3])
    else:
        run()

#!/usr/bin/env python3
"""Per-step kernel time breakdown from a rocprofv3 --kernel-trace --stats CSV (kernel_stats.csv).
usage: kernel_breakdown.py <stats_kernel_stats.csv> <steps profiled> [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"GPU kernel time per step: {tot / steps / 1e6:.3f} ms")
groups = {"scan (ours)": 0.0, "cross/dwconv (ours)": 0.0, "GEMM / weight gradient (ours)": 0.0, "norm / gate / head / loss / params (ours)": 0.0,
          "GEMM (rocBLAS/hipBLASLt)": 0.0, "torch: copies": 0.0, "torch: optimizer": 0.0, "torch: other elementwise/reduce": 0.0}


def own(n):      # kernels of libss2d_hip.so live in anonymous namespaces: "(anonymous namespace)::" demangled, "_GLOBAL__N_" mangled
    return ("anonymous namespace)::" in n or "_GLOBAL__N_" in n) and "at::native" not in n


for r in rows:
    n, t = r["Name"], float(r["TotalDurationNs"])
    if "scan_fwd" in n or "scan_bwd" in n: groups["scan (ours)"] += t
    elif "cross_" in n or "dwconv" in n: groups["cross/dwconv (ours)"] += t
    elif own(n) and ("gemm" in n or "wgrad" in n): groups["GEMM / weight gradient (ours)"] += t
    elif own(n): groups["norm / gate / head / loss / params (ours)"] += t
    elif n.startswith("Cijk") or "gemm" in n.lower(): groups["GEMM (rocBLAS/hipBLASLt)"] += t
    elif "multi_tensor_apply" in n: groups["torch: optimizer"] += t
    elif "copy" in n.lower() or "CatArray" in n: groups["torch: copies"] += t
    else: groups["torch: other elementwise/reduce"] += t
for k, v in groups.items():
    print(f"  {k:44s} {v / steps / 1e6:8.3f} ms  {100 * v / tot:5.1f} %")
for r in rows[:limit]:
    print(f"{float(r['TotalDurationNs']) / steps / 1e3:9.1f} us/step {float(r['Percentage']):6.2f}% calls/step {int(r['Calls']) / steps:6.1f}  {r['Name'][:120]}")

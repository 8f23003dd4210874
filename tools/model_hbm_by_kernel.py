#!/usr/bin/env python3
"""HBM bytes per kernel and training step from the PMC passes of tools/model_hbm.sh (same corrections as tools/model_hbm.py):
    tools/model_hbm_by_kernel.py gpurun_out/<tag> [rows]"""
import collections, csv, glob, re, sys
root, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
tot = collections.defaultdict(lambda: [0.0, 0.0, 0.0])
for c, idx, mul in (("FETCH_SIZE", 0, 2048.0), ("WRITE_SIZE", 1, 1024.0)):
    f = glob.glob(f"{root}/pmc_{c}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "ln_head4_fwd_kernel" in r["Kernel_Name"]]
    a, b = marks[-4], marks[-1]            # the last three complete steps
    for r in rows[a:b]:
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        k = re.sub(r"\((ScanBwdParams|.*Params|.*Args).*", "", k)[:100]
        tot[k][idx] += float(r["Counter_Value"]) * mul / 3
        if idx == 0:
            tot[k][2] += 1 / 3
print("%9s %9s %7s  kernel" % ("read MB", "write MB", "calls"))
for k, (r, w, n) in sorted(tot.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:top]:
    print("%9.0f %9.0f %7.1f  %s" % (r / 1e6, w / 1e6, n, k))
print("total per step: read %.2f GB, written %.2f GB" % (sum(v[0] for v in tot.values()) / 1e9, sum(v[1] for v in tot.values()) / 1e9))

#!/usr/bin/env python3
"""Sum the HBM bytes of one training step from the two PMC passes of tools/model_hbm.sh.  A step is delimited by the
one ln_head4_fwd_kernel dispatch it contains; the last three complete steps are averaged.  Corrections as in
tools/summarize_pmc.py (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE counts a 128-byte
request as 64 bytes on gfx950 -> read bytes = 2 x FETCH_SIZE x 1024, written bytes = WRITE_SIZE x 1024."""
import csv, glob, json, os, sys

root, ms = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.0


def per_step(counter):
    f = glob.glob(os.path.join(root, "pmc_" + counter, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "ln_head4_fwd_kernel" in r["Kernel_Name"]]
    steps = [sum(float(r["Counter_Value"]) for r in rows[a:b]) for a, b in zip(marks, marks[1:])][-3:]
    return sum(steps) / len(steps), len(marks), [b - a for a, b in zip(marks, marks[1:])][-3:]


fetch, n1, k1 = per_step("FETCH_SIZE")
write, n2, k2 = per_step("WRITE_SIZE")
rd, wr = 2.0 * fetch * 1024.0, write * 1024.0
rec = {"workload": "MambaUNet train step, batch 24, 224x224, bf16 autocast + bf16 kernel I/O", "hbm_read_bytes_per_step": rd,
       "hbm_written_bytes_per_step": wr, "hbm_bytes_per_step": rd + wr, "dispatches_per_step": k1, "steps_seen": n1}
if ms > 0:
    rec.update({"ms_per_step": ms, "achieved_GBs": (rd + wr) / ms / 1e6, "peak_GBs": 8000.0, "frac_of_hbm_peak": (rd + wr) / ms / 1e6 / 8000.0})
print(json.dumps(rec))

#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel and dispatch (reads <dir>/pmc_*/**/*counter_collection.csv and,
for the durations, <dir>/trace/**/*kernel_trace.csv), print a table, and with a second argument write the JSON that
bench.py reads (profiles/rNN/bench_pmc.json):

    tools/summarize_pmc.py gpurun_out/<tag> [profiles/rNN/bench_pmc.json]

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes
for wide coalesced reads (MI355X_MICROARCH.md, HBM section), so the corrected read traffic is 2 x FETCH_SIZE x 1024.
SQ_ACTIVE_INST_VALU counts in units of 4 cycles summed over all SIMDs; GRBM_GUI_ACTIVE is the number of GPU clock
cycles the dispatch was active, summed over the 8 XCDs, so with cyc = GRBM_GUI_ACTIVE / 8:
valu_busy = 4 x SQ_ACTIVE_INST_VALU / (n_simd x cyc)   and   clock = cyc / duration (the chip runs the scan well below its 2.4 GHz peak clock)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD = 256 * 4
N_XCD = 8          # GRBM_GUI_ACTIVE is summed over the 8 XCDs' GRBMs
SCAN_SOURCES = ("ss2d_scan_fwd.hip", "ss2d_scan_bwd.hip", "ss2d_scan_bwd_sr.hip", "ss2d_scan.h", "ss2d_common.h")   # as bench.py


def kclass(name):
    if "scan_fwd_kernel" in name:
        return "scan_fwd"
    if "scan_bwd_sr_reduce_kernel" in name:
        return "scan_bwd_reduce"
    if "scan_bwd_sr_kernel" in name:
        return "scan_bwd_sr"
    if "scan_bwd_kernel" in name:
        return "scan_bwd_general"
    return None


def main():
    root = sys.argv[1]
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = kclass(r["Kernel_Name"])
                if k:
                    a = acc[(k, r["Counter_Name"])]
                    a[0] += float(r["Counter_Value"])
                    a[1] += 1
    dur = collections.defaultdict(lambda: [0.0, 0])
    seqs = collections.defaultdict(list)       # (start, duration) per dispatch: the timed steps are the last ones of the run
    for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = kclass(r["Kernel_Name"])
                if k:
                    dur[k][0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                    dur[k][1] += 1
                    seqs[k].append((float(r["Start_Timestamp"]), float(r["End_Timestamp"]) - float(r["Start_Timestamp"])))
    avg = {kc: s / n for kc, (s, n) in acc.items()}
    print(f"{'kernel':18s} {'counter':26s} {'avg per dispatch':>20s} {'dispatches':>10s}")
    for (k, c), (s, n) in sorted(acc.items()):
        print(f"{k:18s} {c:26s} {s / n:20.1f} {n:10d}")
    kernels = sorted({k for k, _ in acc})
    traffic, valu, dur_us = {}, {}, {}
    for k in kernels:
        if k in dur:
            dur_us[k] = dur[k][0] / dur[k][1] / 1e3
            print(f"{k}: average duration under --kernel-trace {dur_us[k]:.1f} us over {dur[k][1]} dispatches")
            last = [d for _, d in sorted(seqs[k])][-20:]
            print(f"{k}: the last {len(last)} dispatches (the timed steps; the run starts with bench.py's pre-warm) average {sum(last) / len(last) / 1e3:.1f} us")
        fs, ws = avg.get((k, "FETCH_SIZE")), avg.get((k, "WRITE_SIZE"))
        if fs is not None and ws is not None:
            rd, wr = 2 * fs * 1024, ws * 1024
            traffic[k] = int(rd + wr)
            print(f"{k}: HBM traffic per launch = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 = {rd:.0f} + {wr:.0f} = {rd + wr:.0f} bytes")
        act, gui, ins = avg.get((k, "SQ_ACTIVE_INST_VALU")), avg.get((k, "GRBM_GUI_ACTIVE")), avg.get((k, "SQ_INSTS_VALU"))
        if act is not None:
            v = {"SQ_ACTIVE_INST_VALU": act, "SQ_INSTS_VALU": ins}
            if gui:
                v["GRBM_GUI_ACTIVE"] = gui
                cyc = gui / N_XCD
                v["valu_busy"] = round(4 * act / (N_SIMD * cyc), 4)
                if k in dur_us:
                    v["clock_ghz"] = round(cyc / (dur_us[k] * 1e3), 3)
                print(f"{k}: VALU busy = 4*SQ_ACTIVE_INST_VALU/({N_SIMD}*GRBM_GUI_ACTIVE/{N_XCD}) = {v['valu_busy']:.3f}"
                      + (f", clock = {v['clock_ghz']:.2f} GHz" if "clock_ghz" in v else ""))
            valu[k] = v
    if len(sys.argv) > 2:
        # bench.py speaks of two kernels: the backward = state-row kernel + its partial-tile reduction (both launches)
        def both(d, f):
            if "scan_bwd_sr" in d:
                return f(d["scan_bwd_sr"], d.get("scan_bwd_reduce"))
            return d.get("scan_bwd_general")
        tb = {"scan_fwd": traffic.get("scan_fwd"), "scan_bwd": both(traffic, lambda a, b: a + (b or 0))}
        vb = {"scan_fwd": valu.get("scan_fwd"), "scan_bwd": both(valu, lambda a, b: a)}
        h = hashlib.sha256()
        for f in SCAN_SOURCES:
            with open(os.path.join(ROOT, "mamba_unet_amd", "csrc", f), "rb") as fh:
                h.update(fh.read())
        out = {"sources_hash": h.hexdigest()[:16], "traffic_bytes": tb, "valu": vb, "durations_us": dur_us,
               "traffic_bytes_per_kernel": traffic,
               "how": "tools/profile_round.sh: separate rocprofv3 --pmc passes of `bench.py --steps 3`, averaged per dispatch; "
                      "traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024; valu_busy = 4*SQ_ACTIVE_INST_VALU/(1024*GRBM_GUI_ACTIVE/8)"}
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[2])), exist_ok=True)
        with open(sys.argv[2], "w") as fh:
            json.dump(out, fh, indent=1)
        print("wrote", sys.argv[2])


if __name__ == "__main__":
    main()

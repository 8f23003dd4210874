#!/usr/bin/env python3
"""Compile one .hip file to gfx950 assembly and print, for a kernel whose mangled name contains the given substring,
the register/spill summary and the instruction mix of its loop blocks (depth >= DEPTH).
    tools/isa_mix.py mamba_unet_amd/csrc/ss2d_scan_bwd_sr.hip 'sr_kernelILb1EfLb0' [depth]"""
import collections, re, subprocess, sys, tempfile, os
src, pat = sys.argv[1], sys.argv[2]
depth = int(sys.argv[3]) if len(sys.argv) > 3 else 3
out = os.path.join(tempfile.gettempdir(), "isa_mix.s")
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-command-line-argument",
                "-S", "--cuda-device-only", *os.environ.get("ISA_DEFS", "").split(), src, "-o", out], check=True)
s = open(out).read()
names = [m.group(1) for m in re.finditer(r"^(\w+):\s*; @", s, re.M) if pat in m.group(1)]
for name in names:
    i = s.index(name + ":"); j = s.index(".Lfunc_end", i)
    body = s[i:j].split("\n")
    meta = s[j:j + 6000]
    print(name)
    for key in ("NumVgprs", "NumAgprs", "ScratchSize", "Occupancy", "NumSgprs", "codeLenInByte"):
        m = re.search(r"; %s: (\S+)" % key, meta)
        print("  ", key, m.group(1) if m else "?")
    labels = [n for n, l in enumerate(body) if re.match(r"^\.LBB", l)] + [len(body)]
    tot = collections.Counter()
    for a, b in zip(labels, labels[1:]):
        m = re.search(r"Depth=(\d+)", body[a])
        if not m or int(m.group(1)) < depth:
            continue
        for l in body[a + 1:b]:
            t = l.strip().split(" ")[0]
            if t and t[0] not in ".;" and not t.endswith(":"):
                tot[t] += 1
    valu = sum(v for k, v in tot.items() if k.startswith("v_"))
    print("   loop blocks depth>=%d: %d instructions, %d VALU, %d exp/log/rcp, %d ds, %d scratch, %d s_nop, %d s_waitcnt, %d v_mov" % (
        depth, sum(tot.values()), valu, sum(v for k, v in tot.items() if re.match(r"v_(exp|log|rcp)", k)),
        sum(v for k, v in tot.items() if k.startswith("ds_")), sum(v for k, v in tot.items() if k.startswith("scratch_")),
        tot["s_nop"], tot["s_waitcnt"], sum(v for k, v in tot.items() if k.startswith("v_mov"))))
    if os.environ.get("ISA_MIX_FULL"):
        for k, v in tot.most_common(60):
            print("      %4d %s" % (v, k))

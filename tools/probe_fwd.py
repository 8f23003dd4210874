#!/usr/bin/env python3
"""Phase timing of the scan forward from s_memtime probes compiled into a debug build:
    make -C mamba_unet_amd/csrc variant SRC=ss2d_scan_fwd NAME=fprobe DEFS=-DSS2D_FWD_PROBE
    SS2D_HIP_LIB=scratch/libss2d_fprobe.so python tools/probe_fwd.py          (on the GPU box)"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mamba_unet_amd import _lib  # noqa: E402
from mamba_unet_amd.synthetic import make_scan_inputs  # noqa: E402

B, K, D, L, N = 24, 4, 192, 3136, 16
KD = K * D
lib, P = _lib.load(), _lib.ptr
raw = ctypes.CDLL(_lib.LIB_PATH)
raw.ss2d_debug_set_fwd_probe.argtypes = [ctypes.c_void_p]
NP = 8
inp = {k: v.cuda() for k, v in make_scan_inputs(B, KD, L, dstate=N, ngroups=K, seed=0).items()}
out = torch.empty_like(inp["u"])
ckpt = torch.empty(B, KD, lib.ss2d_scan_num_chunks(L), N, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
nwaves = B * KD // 4
probe = torch.zeros(nwaves * (NP + 1), dtype=torch.int64, device="cuda")
assert raw.ss2d_debug_set_fwd_probe(ctypes.c_void_p(probe.data_ptr())) == 0
for it in range(3):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    _lib.check(lib.ss2d_selective_scan_fwd(0, P(inp["u"]), P(inp["delta"]), P(inp["A"]), P(inp["B"]), P(inp["C"]), P(inp["D"]), None,
                                           P(inp["delta_bias"]), P(out), None, P(ckpt), B, KD, L, N, K, 1, st), "fwd")
    ev1.record()
    torch.cuda.synchronize()
print("fwd with probes: %.1f us" % (ev0.elapsed_time(ev1) * 1e3))
t = probe.view(-1, NP + 1).double().cpu()
t = t[t[:, NP] > 0]
tot = t[:, NP].mean().item()
names = ["0 loop top", "1 B/C tile: global loads + LDS stores", "2 u/delta prefetch issue + softplus", "3 barrier", "4 checkpoint store",
         "5 the 8 state pairs", "6 y sum + store (then: start clock)", "7 (end clock)"]
print("waves %d, mean total %.0f ticks" % (t.shape[0], tot))
for i, n in enumerate(names):
    print("  %-42s %12.0f  %5.1f %%" % (n, t[:, i].mean().item(), 100 * t[:, i].mean().item() / tot))

# launch rounds: start / end of every wave on the device-wide 100 MHz clock (slots 6, 7)
st0 = (t[:, 6] - t[:, 6].min()) / 100.0
en = (t[:, 7] - t[:, 6].min()) / 100.0
print("span %.1f us (event: %.1f us); core ticks per us: %.0f" % (en.max().item(), ev0.elapsed_time(ev1) * 1e3, (t[:, NP] / (en - st0)).median().item()))
late = st0 > 0.05 * en.max()
print("waves starting at once: %d, mean duration %.1f us; later: %d, mean start %.1f us, mean duration %.1f us" % (
    (~late).sum().item(), (en - st0)[~late].mean().item(), late.sum().item(), st0[late].mean().item() if late.any() else 0,
    (en - st0)[late].mean().item() if late.any() else 0))
import numpy as np
h, edges = np.histogram(st0.numpy(), bins=12)
print("start-time histogram (us):", [(round(float(e)), int(c)) for e, c in zip(edges[:-1], h)])
h, edges = np.histogram(en.numpy(), bins=12)
print("end-time histogram (us):", [(round(float(e)), int(c)) for e, c in zip(edges[:-1], h)])

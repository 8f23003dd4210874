#!/usr/bin/env python3
"""Phase timing of the state-row scan backward from s_memtime probes compiled into a debug build
(make -C mamba_unet_amd/csrc probe -> scratch/libss2d_probe.so).  Run on the GPU box:
    SS2D_HIP_LIB=scratch/libss2d_probe.so python tools/probe_bwd.py
Prints, averaged over all waves, the cycles each phase of ss2d_scan_bwd_sr.hip took (probe ids in the source)."""
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
raw.ss2d_debug_set_probe.argtypes = [ctypes.c_void_p]
NP = 16
inp = {k: v.cuda() for k, v in make_scan_inputs(B, KD, L, dstate=N, ngroups=K, seed=0).items()}
dout = torch.randn(B, KD, L, generator=torch.Generator().manual_seed(1)).cuda()
out = torch.empty_like(inp["u"])
ckpt = torch.empty(B, KD, lib.ss2d_scan_num_chunks(L), N, device="cuda")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
_lib.check(lib.ss2d_selective_scan_dir_fwd(0, P(inp["u"]), P(inp["delta"]), P(inp["A"]), P(inp["B"]), P(inp["C"]), P(inp["D"]), None,
                                           P(inp["delta_bias"]), P(out), None, P(ckpt), B, KD, L, N, K, 1, K, 0, 0, st), "fwd")
wsb = int(lib.ss2d_scan_bwd_workspace_bytes(B, KD, L, N, K))
ws = torch.empty(wsb // 4, device="cuda")
du, dd = torch.empty_like(out), torch.empty_like(out)
dA, dD, db = torch.zeros(KD, N, device="cuda"), torch.zeros(KD, device="cuda"), torch.zeros(KD, device="cuda")
dB, dC = torch.empty(B, K, N, L, device="cuda"), torch.empty(B, K, N, L, device="cuda")
probe = torch.zeros(256 * 8 * (NP + 1), dtype=torch.int64, device="cuda")
assert raw.ss2d_debug_set_probe(ctypes.c_void_p(probe.data_ptr())) == 0
for it in range(3):
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    _lib.check(lib.ss2d_selective_scan_dir_bwd_ws(0, P(inp["u"]), P(inp["delta"]), P(inp["A"]), P(inp["B"]), P(inp["C"]), P(inp["D"]), None,
                                                  P(inp["delta_bias"]), P(dout), P(ckpt), P(du), P(dd), P(dA), P(dB), P(dC), P(dD), None, P(db),
                                                  B, KD, L, N, K, 1, K, 0, 0, P(ws), wsb, st), "bwd")
    ev1.record()
    torch.cuda.synchronize()
print("bwd with probes: %.1f us" % (ev0.elapsed_time(ev1) * 1e3))
t = probe.view(-1, NP + 1).double().cpu()
t = t[t[:, NP] > 0]
tot = t[:, NP].mean().item()
names = ["0 loop top (next-input loads issue)", "1 softplus+stage+LDS reads", "2 pair0 fwd states", "3 pair0 adjoint local+scan", "4 pair0 true pass",
         "5 pair1 fwd states", "6 pair1 adjoint local+scan", "7 pair1 true pass", "8 flush: second barrier", "9 tile staging (barriers)", "10 flush: slot reads + sum + global store", "11 row sums + record RMW + carry + tile_load issue", "12 (8b) cross-row reductions", "13 (8c) final math + stores", "14 flush: first barrier (wait for all waves)", "15 flush: slot writes"]
print("waves %d, mean total %.0f ticks" % (t.shape[0], tot))
for i, n in enumerate(names):
    print("  %-40s %12.0f  %5.1f %%" % (n, t[:, i].mean().item(), 100 * t[:, i].mean().item() / tot))
print("  %-40s %12.0f  %5.1f %%" % ("unaccounted", tot - t[:, :NP].sum(1).mean().item(), 100 * (tot - t[:, :NP].sum(1).mean().item()) / tot))

"""ss2d_selective_scan_dir_bwd_static -- the entry point that allocates nothing (its dB/dC partial-tile workspace is a
fixed array inside the library's code object) -- against ss2d_selective_scan_dir_bwd_ws with a caller-allocated
workspace, through the C ABI: same kernels, so du / ddelta / dB / dC must agree bit for bit and the atomically
accumulated dA / dD / ddelta_bias to the fp32 gate.  Two calls in a row, the second on a smaller input (the shared
array still holds the first call's partial tiles), then the size / shape rejections."""
import ctypes

import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu

STATIC_WS_BYTES = 64 << 20     # include/ss2d.h: SS2D_STATIC_WS_BYTES


def _run(lib, P, entry, inp, ckpt, dout, batch, dim, L, N, G, rev_mask, u_groups, ws):
    dev = dout.device
    du, ddelta = torch.full_like(inp["delta"], float("nan")), torch.full_like(inp["delta"], float("nan"))
    dA, dD, dbias = torch.zeros(dim, N, device=dev), torch.zeros(dim, device=dev), torch.zeros(dim, device=dev)
    dB, dC = torch.full((batch, G, N, L), float("nan"), device=dev), torch.full((batch, G, N, L), float("nan"), device=dev)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    args = [0, P(inp["u"]), P(inp["delta"]), P(inp["A"]), P(inp["B"]), P(inp["C"]), P(inp["D"]), None, P(inp["delta_bias"]),
            P(dout), P(ckpt), P(du), P(ddelta), P(dA), P(dB), P(dC), P(dD), None, P(dbias), batch, dim, L, N, G, 1,
            u_groups, rev_mask, 0]
    if entry == "static":
        rc = lib.ss2d_selective_scan_dir_bwd_static(*args, st)
    else:
        rc = lib.ss2d_selective_scan_dir_bwd_ws(*args, P(ws), ws.numel() * 4, st)
    torch.cuda.synchronize()
    return rc, {"du": du, "ddelta": ddelta, "dA": dA, "dB": dB, "dC": dC, "dD": dD, "dbias": dbias}


def _case(batch, dim, L, G, rev_mask, u_groups, seed):
    from mamba_unet_amd import _lib
    from oracle.ss2d_ref import make_scan_inputs
    lib, P = _lib.load(), _lib.ptr
    N = 16
    inp = {k: v.cuda() for k, v in make_scan_inputs(batch, dim, L, dstate=N, ngroups=G, seed=seed, integer_A=False).items()}
    if u_groups != G:      # a direction and its reversal share one copy of u / dout
        inp["u"] = inp["u"].view(batch, G, dim // G, L)[:, ::G // u_groups].reshape(batch, -1, L).contiguous()
    dout = torch.randn(inp["u"].shape, generator=torch.Generator().manual_seed(seed + 1)).cuda()
    ckpt = torch.empty(batch, dim, lib.ss2d_scan_num_chunks(L), N, device="cuda")
    out = torch.empty_like(inp["delta"])
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.ss2d_selective_scan_dir_fwd(0, P(inp["u"]), P(inp["delta"]), P(inp["A"]), P(inp["B"]), P(inp["C"]), P(inp["D"]),
                                               None, P(inp["delta_bias"]), P(out), None, P(ckpt), batch, dim, L, N, G, 1,
                                               u_groups, rev_mask, 0, st), "fwd")
    need = int(lib.ss2d_scan_bwd_workspace_bytes(batch, dim, L, N, G))
    assert 0 < need <= STATIC_WS_BYTES
    ws = torch.empty(need // 4, device="cuda")
    rc, want = _run(lib, P, "ws", inp, ckpt, dout, batch, dim, L, N, G, rev_mask, u_groups, ws)
    _lib.check(rc, "bwd_ws")
    rc, got = _run(lib, P, "static", inp, ckpt, dout, batch, dim, L, N, G, rev_mask, u_groups, None)
    _lib.check(rc, "bwd_static")
    for k in ("du", "ddelta", "dB", "dC"):        # plain stores of the same arithmetic: bit for bit
        assert torch.equal(got[k], want[k]), k
    for k in ("dA", "dD", "dbias"):               # float atomics: order-dependent in the last bits
        assert rel_l2(got[k].cpu(), want[k].cpu()) < 1e-5, k
        assert torch.isfinite(got[k]).all()


def test_static_entry_matches_workspace_entry_twice_second_smaller():
    _case(1, 768, 3136, 4, 0b1010, 2, seed=3)       # one 224x224 stage-0 image, fused directions: 40 MB of partial tiles
    _case(1, 96, 196, 4, 0, 4, seed=5)              # smaller: stale partial tiles of the first call stay in the array


def test_static_entry_rejects_what_it_cannot_hold():
    from mamba_unet_amd import _lib
    lib, P = _lib.load(), _lib.ptr
    t = torch.zeros(64, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ptrs = [P(t)] * 6 + [None] + [P(t)] * 9 + [None, P(t)]
    big = (24, 768, 3136, 16, 4)                                   # BASELINE configs[1]: 141 MB of partial tiles
    assert int(lib.ss2d_scan_bwd_workspace_bytes(*big)) > STATIC_WS_BYTES
    rc = lib.ss2d_selective_scan_dir_bwd_static(0, *ptrs, *big, 1, 4, 0, 0, st)
    assert rc == -8 and b"workspace" in lib.ss2d_last_error()
    rc = lib.ss2d_selective_scan_dir_bwd_static(0, *ptrs, 1, 8, 30, 16, 4, 1, 4, 0, 0, st)     # L % 4 != 0
    assert rc == -7
    rc = lib.ss2d_selective_scan_dir_bwd_static(0, *ptrs, 1, 8, 32, 8, 4, 1, 4, 0, 0, st)      # dstate != 16
    assert rc == -7
    zp = list(ptrs)
    zp[6] = P(t)
    rc = lib.ss2d_selective_scan_dir_bwd_static(0, *zp, 1, 8, 32, 16, 4, 1, 4, 0, 0, st)       # z gate
    assert rc == -7 and b"z gate" in lib.ss2d_last_error()


def test_static_entry_under_hip_graph_capture_allocates_nothing():
    """What the entry point exists for: it can be captured in a HIP graph (no allocation, no synchronisation inside) and
    the device's free memory does not move across calls.  One eager call primes the per-device symbol lookup."""
    from mamba_unet_amd import _lib
    from mamba_unet_amd.synthetic import make_scan_inputs
    lib, P = _lib.load(), _lib.ptr
    batch, dim, L, N, G = 1, 96, 1792, 16, 4
    inp = {k: v.cuda() for k, v in make_scan_inputs(batch, dim, L, dstate=N, ngroups=G, seed=8, integer_A=False).items()}
    dout = torch.randn(batch, dim, L, generator=torch.Generator().manual_seed(9)).cuda()
    ckpt = torch.empty(batch, dim, lib.ss2d_scan_num_chunks(L), N, device="cuda")
    out = torch.empty_like(inp["delta"])
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.ss2d_selective_scan_dir_fwd(0, P(inp["u"]), P(inp["delta"]), P(inp["A"]), P(inp["B"]), P(inp["C"]), P(inp["D"]),
                                               None, P(inp["delta_bias"]), P(out), None, P(ckpt), batch, dim, L, N, G, 1, G, 0, 0, st), "fwd")
    ws = torch.empty(int(lib.ss2d_scan_bwd_workspace_bytes(batch, dim, L, N, G)) // 4, device="cuda")
    rc, want = _run(lib, P, "ws", inp, ckpt, dout, batch, dim, L, N, G, 0, G, ws)
    _lib.check(rc, "bwd_ws")
    rc, _ = _run(lib, P, "static", inp, ckpt, dout, batch, dim, L, N, G, 0, G, None)      # primes hipGetSymbolAddress
    _lib.check(rc, "bwd_static eager")
    # static output buffers for the captured call
    dev = dout.device
    du, ddelta = torch.empty_like(inp["delta"]), torch.empty_like(inp["delta"])
    acc = torch.zeros(dim * N + 2 * dim, device=dev)
    dA, dD, dbias = acc[:dim * N].view(dim, N), acc[dim * N:dim * N + dim], acc[dim * N + dim:]
    dB, dC = torch.empty(batch, G, N, L, device=dev), torch.empty(batch, G, N, L, device=dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        acc.zero_()
        gst = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = lib.ss2d_selective_scan_dir_bwd_static(0, P(inp["u"]), P(inp["delta"]), P(inp["A"]), P(inp["B"]), P(inp["C"]), P(inp["D"]), None,
                                                    P(inp["delta_bias"]), P(dout), P(ckpt), P(du), P(ddelta), P(dA), P(dB), P(dC), P(dD), None,
                                                    P(dbias), batch, dim, L, N, G, 1, G, 0, 0, gst)
    _lib.check(rc, "bwd_static under capture")      # (an allocation or a synchronisation inside the call would have broken the capture)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]            # behind the graph's instantiation, which takes memory of its own
    for _ in range(2):
        du.fill_(float("nan")); dB.fill_(float("nan"))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(du, want["du"]) and torch.equal(ddelta, want["ddelta"])
        assert torch.equal(dB, want["dB"]) and torch.equal(dC, want["dC"])
        assert rel_l2(dA.cpu(), want["dA"].cpu()) < 1e-5 and rel_l2(dbias.cpu(), want["dbias"].cpu()) < 1e-5
    assert torch.cuda.mem_get_info()[0] >= free0          # the replays allocated nothing

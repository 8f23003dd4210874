"""GPU parity of the selective-scan backward (all 7/8 gradients) against the oracle, through the C ABI."""
import numpy as np
import pytest
import torch

from conftest import rel_l2, rel_max

pytestmark = pytest.mark.gpu

# north_star: rel err <= 1e-5 in fp32 (L2-relative).  Gradients that are sums of
# ~batch*L products (dA, dD, ddelta_bias) are compared against a float64-accumulating oracle; the
# fp32 atomics that accumulate them on the GPU are order-dependent, hence the slightly wider max-norm gate.
L2_TOL = 1e-5
MAX_TOL = 5e-5


def _grads_gpu(inp, z, softplus, dout):
    from mamba_unet_amd import selective_scan_fn
    leaves = {k: (v.cuda().requires_grad_(True) if v is not None else None) for k, v in inp.items()}
    zl = z.cuda().requires_grad_(True) if z is not None else None
    y = selective_scan_fn(leaves["u"], leaves["delta"], leaves["A"], leaves["B"], leaves["C"], leaves["D"], zl,
                          leaves["delta_bias"], softplus)
    y.backward(dout.cuda())
    torch.cuda.synchronize()
    out = {k: (v.grad.cpu() if v is not None else None) for k, v in leaves.items()}
    out["z"] = zl.grad.cpu() if zl is not None else None
    return y.detach().cpu(), out


def _check(got, want, name):
    assert rel_l2(got, want) < L2_TOL, (name, rel_l2(got, want))
    assert rel_max(got, want) < MAX_TOL, (name, rel_max(got, want))


@pytest.mark.parametrize("batch,dim,L,N,G", [
    (2, 64, 35, 16, 4),
    (1, 128, 64, 16, 4),
    (3, 32, 300, 16, 4),
    (2, 24, 129, 16, 2),
    (5, 8, 515, 7, 1),
    (1, 20, 1000, 4, 4),
    (2, 384, 196, 16, 4),     # 96 channels per group -> 3 rounds of 8 waves
])
@pytest.mark.parametrize("with_z", [False, True])
def test_scan_bwd_matches_autograd_of_oracle(batch, dim, L, N, G, with_z):
    from oracle.selective_scan_ref import selective_scan_ref
    from oracle.ss2d_ref import make_scan_inputs
    inp = make_scan_inputs(batch, dim, L, dstate=N, ngroups=G, seed=batch * 1000 + L, integer_A=False)
    z = torch.randn(batch, dim, L, generator=torch.Generator().manual_seed(7)) if with_z else None
    dout = torch.randn(batch, dim, L, generator=torch.Generator().manual_seed(1))
    leaves = {k: v.clone().double().requires_grad_(True) for k, v in inp.items()}
    zl = z.clone().double().requires_grad_(True) if with_z else None
    y = selective_scan_ref(leaves["u"], leaves["delta"], leaves["A"], leaves["B"], leaves["C"], leaves["D"], zl,
                           leaves["delta_bias"], True, compute_dtype=torch.float64)
    y.backward(dout.double())
    y_gpu, g = _grads_gpu(inp, z, True, dout)
    _check(y_gpu, y.detach(), "out")
    for k in ("u", "delta", "A", "B", "C", "D", "delta_bias"):
        _check(g[k], leaves[k].grad, "d" + k)
    if with_z:
        _check(g["z"], zl.grad, "dz")


def test_scan_bwd_3d_BC_no_bias_no_D_no_softplus():
    from oracle.selective_scan_ref import selective_scan_ref
    from oracle.ss2d_ref import make_scan_inputs
    inp = make_scan_inputs(2, 16, 77, dstate=16, ngroups=1, seed=11, integer_A=False)
    inp["B"], inp["C"] = inp["B"][:, 0].contiguous(), inp["C"][:, 0].contiguous()
    inp["delta"] = inp["delta"].abs() * 0.2
    inp["D"], inp["delta_bias"] = None, None
    dout = torch.randn(2, 16, 77, generator=torch.Generator().manual_seed(1))
    leaves = {k: (v.clone().double().requires_grad_(True) if v is not None else None) for k, v in inp.items()}
    y = selective_scan_ref(leaves["u"], leaves["delta"], leaves["A"], leaves["B"], leaves["C"], None, None, None, False,
                           compute_dtype=torch.float64)
    y.backward(dout.double())
    _, g = _grads_gpu(inp, None, False, dout)
    for k in ("u", "delta", "A", "B", "C"):
        assert g[k].shape == leaves[k].grad.shape
        _check(g[k], leaves[k].grad, "d" + k)


def test_scan_bwd_config2_slice_against_c_oracle():
    """BASELINE.json configs[1] shape at batch 2 (the plain-C oracle needs ~10 s per batch pair)."""
    from oracle import c_ref
    from oracle.ss2d_ref import make_scan_inputs
    inp = make_scan_inputs(2, 768, 3136, dstate=16, ngroups=4, seed=0, integer_A=True)
    dout = torch.randn(2, 768, 3136, generator=torch.Generator().manual_seed(1))
    _, g = _grads_gpu(inp, None, True, dout)
    npi = {k: v.numpy() for k, v in inp.items()}
    want = c_ref.scan_bwd(npi["u"], npi["delta"], npi["A"], npi["B"], npi["C"], npi["D"], None, npi["delta_bias"],
                          dout.numpy(), True)
    names = {"du": "u", "ddelta": "delta", "dA": "A", "dB": "B", "dC": "C", "dD": "D", "ddelta_bias": "delta_bias"}
    for k, src in names.items():
        _check(g[src], want[k], k)


def test_scan_bwd_full_size_adjoint_identity():
    """Size-independent property at the full config-2 size: the scan is linear in u, so
    <dout, scan(v) - scan(0)> == <du, v> where du is the gradient w.r.t. u (adjoint identity)."""
    from mamba_unet_amd import selective_scan_fn
    from oracle.ss2d_ref import make_scan_inputs
    inp = {k: v.cuda() for k, v in make_scan_inputs(24, 768, 3136, seed=0).items()}
    u = inp["u"].clone().requires_grad_(True)
    dout = torch.randn_like(u)
    y = selective_scan_fn(u, inp["delta"], inp["A"], inp["B"], inp["C"], inp["D"], None, inp["delta_bias"], True)
    y.backward(dout)
    v = torch.randn_like(u)
    with torch.no_grad():
        yv = selective_scan_fn(v, inp["delta"], inp["A"], inp["B"], inp["C"], inp["D"], None, inp["delta_bias"], True)
    lhs = (dout.double() * yv.double()).sum().item()
    rhs = (u.grad.double() * v.double()).sum().item()
    assert abs(lhs - rhs) / abs(lhs) < 1e-5


def test_misaligned_parameter_views_are_handled():
    """A / D / delta_bias handed over as views at odd storage offsets: selective_scan_fn copies them to aligned
    storage, and the C ABI refuses the workspace form -- which WRITES dB / dC -- loudly instead of silently accumulating
    into uninitialised gradients."""
    import ctypes
    from mamba_unet_amd import _lib, selective_scan_fn
    from mamba_unet_amd.synthetic import make_scan_inputs
    from oracle.selective_scan_ref import selective_scan_ref
    batch, dim, L, N, G = 2, 32, 256, 16, 4
    inp = make_scan_inputs(batch, dim, L, dstate=N, ngroups=G, seed=21, integer_A=False)
    dout = torch.randn(batch, dim, L, generator=torch.Generator().manual_seed(22))
    ref = {k: v.clone().requires_grad_(True) for k, v in inp.items()}
    selective_scan_ref(ref["u"], ref["delta"], ref["A"], ref["B"], ref["C"], ref["D"], None, ref["delta_bias"], True).backward(dout)
    gpu = {k: v.cuda() for k, v in inp.items()}
    pad = torch.zeros(dim * N + 1, device="cuda")
    pad[1:] = gpu["A"].reshape(-1)
    A_odd = pad[1:].view(dim, N)                       # 4 bytes past a 16-byte boundary
    assert A_odd.data_ptr() % 16 == 4
    leaves = {k: (A_odd if k == "A" else v).detach().requires_grad_(True) for k, v in gpu.items()}
    y = selective_scan_fn(leaves["u"], leaves["delta"], leaves["A"], leaves["B"], leaves["C"], leaves["D"], None, leaves["delta_bias"], True)
    y.backward(dout.cuda())
    for k in inp:
        assert rel_l2(leaves[k].grad.cpu(), ref[k].grad) < 1e-5, k
    # the C ABI with the same misaligned A and a sufficient workspace: an error, not garbage
    lib, P = _lib.load(), _lib.ptr
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ckpt = torch.empty(batch, dim, lib.ss2d_scan_num_chunks(L), N, device="cuda")
    out = torch.empty_like(gpu["u"])
    _lib.check(lib.ss2d_selective_scan_fwd(0, P(gpu["u"]), P(gpu["delta"]), P(A_odd), P(gpu["B"]), P(gpu["C"]), P(gpu["D"]), None,
                                           P(gpu["delta_bias"]), P(out), None, P(ckpt), batch, dim, L, N, G, 1, st), "fwd")
    nb = int(lib.ss2d_scan_bwd_workspace_bytes(batch, dim, L, N, G))
    assert nb > 0
    ws = torch.empty(nb // 4, device="cuda")
    z = lambda *sh: torch.zeros(*sh, device="cuda")           # noqa: E731
    rc = lib.ss2d_selective_scan_dir_bwd_ws(0, P(gpu["u"]), P(gpu["delta"]), P(A_odd), P(gpu["B"]), P(gpu["C"]), P(gpu["D"]), None,
                                            P(gpu["delta_bias"]), P(dout.cuda()), P(ckpt), P(torch.empty_like(out)), P(torch.empty_like(out)),
                                            P(z(dim, N)), P(z(batch, G, N, L)), P(z(batch, G, N, L)), P(z(dim)), None, P(z(dim)),
                                            batch, dim, L, N, G, 1, G, 0, 0, P(ws), nb, st)
    assert rc == -7 and b"aligned" in lib.ss2d_last_error()

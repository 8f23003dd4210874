"""GPU parity of the selective-scan forward against the oracle, through the C ABI (ctypes)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2, rel_max

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-5  # BASELINE.json north_star: rel err <= 1e-5 in fp32


def _run(inp, z=None, softplus=True, return_last_state=False):
    from mamba_unet_amd import selective_scan_fn
    dev = {k: (v.cuda() if v is not None else None) for k, v in inp.items()}
    zz = z.cuda() if z is not None else None
    out = selective_scan_fn(dev["u"], dev["delta"], dev["A"], dev["B"], dev["C"], dev["D"], zz, dev["delta_bias"],
                            softplus, return_last_state)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("batch,dim,L,N,G", [
    (2, 64, 35, 16, 4),      # ragged L, one partial chunk
    (1, 128, 64, 16, 4),
    (3, 32, 300, 16, 4),     # 3 chunks, last partial, L % 4 == 0
    (2, 24, 129, 16, 2),     # L odd -> scalar loads; rows masked (12 channels per group)
    (5, 8, 515, 7, 1),       # odd dstate, single group, batch not multiple of anything
    (2, 768, 196, 16, 4),    # stage-2-like
    (1, 20, 1000, 4, 4),     # 5 channels per group
])
@pytest.mark.parametrize("with_z", [False, True])
def test_scan_fwd_matches_oracle(batch, dim, L, N, G, with_z):
    from oracle.selective_scan_ref import selective_scan_ref
    from oracle.ss2d_ref import make_scan_inputs
    inp = make_scan_inputs(batch, dim, L, dstate=N, ngroups=G, seed=batch * 1000 + L, integer_A=False)
    z = torch.randn(batch, dim, L, generator=torch.Generator().manual_seed(7)) if with_z else None
    want, want_last = selective_scan_ref(inp["u"], inp["delta"], inp["A"], inp["B"], inp["C"], inp["D"], z,
                                         inp["delta_bias"], True, return_last_state=True)
    got, got_last = _run(inp, z, True, return_last_state=True)
    assert got.dtype == torch.float32 and got.shape == want.shape
    assert rel_l2(got.cpu(), want) < FP32_TOL and rel_max(got.cpu(), want) < FP32_TOL
    assert rel_l2(got_last.cpu(), want_last) < FP32_TOL


def test_scan_fwd_3d_BC_no_bias_no_D_no_softplus():
    from oracle.selective_scan_ref import selective_scan_ref
    from oracle.ss2d_ref import make_scan_inputs
    inp = make_scan_inputs(2, 16, 77, dstate=16, ngroups=1, seed=11, integer_A=False)
    inp["B"], inp["C"] = inp["B"][:, 0].contiguous(), inp["C"][:, 0].contiguous()
    inp["delta"] = inp["delta"].abs() * 0.2
    inp["D"], inp["delta_bias"] = None, None
    want = selective_scan_ref(inp["u"], inp["delta"], inp["A"], inp["B"], inp["C"], None, None, None, False)
    got = _run(inp, None, False)
    assert rel_l2(got.cpu(), want) < FP32_TOL and rel_max(got.cpu(), want) < FP32_TOL


@pytest.mark.parametrize("name", ["ss2d_2x8x5x7_s0.npz", "ss2d_1x16x8x8_s1.npz"])
def test_scan_fwd_matches_reference_call_site_golden(name):
    """Arguments captured at the reference's own call site (mamba_sys.py:436-442) and the out_y it got."""
    g = np.load(os.path.join(GOLDEN, name))
    inp = {k: torch.from_numpy(g["scan." + k]) for k in ("u", "delta", "A", "B", "C", "D", "delta_bias")}
    got = _run(inp, None, True)
    assert rel_l2(got.cpu(), g["scan.out_y"]) < FP32_TOL and rel_max(got.cpu(), g["scan.out_y"]) < FP32_TOL


def test_scan_fwd_config2_against_c_oracle():
    """BASELINE.json configs[1] shape, checked against the plain-C oracle on a batch slice (full size
    goes through the size-independent property below)."""
    from oracle import c_ref
    from oracle.ss2d_ref import make_scan_inputs
    inp = make_scan_inputs(4, 768, 3136, dstate=16, ngroups=4, seed=0, integer_A=True)
    got = _run(inp, None, True).cpu().numpy()
    npi = {k: v.numpy() for k, v in inp.items()}
    want = c_ref.scan_fwd(npi["u"], npi["delta"], npi["A"], npi["B"], npi["C"], npi["D"], None, npi["delta_bias"], True)
    assert rel_l2(got, want) < FP32_TOL and rel_max(got, want) < FP32_TOL


def test_scan_fwd_linearity_in_u_full_size():
    """Size-independent property at the full config-2 size: for fixed delta/A/B/C the scan is linear in u
    (D included):  scan(2u + v) == 2 scan(u) + scan(v)."""
    from mamba_unet_amd import selective_scan_fn
    from oracle.ss2d_ref import make_scan_inputs
    inp = {k: v.cuda() for k, v in make_scan_inputs(24, 768, 3136, seed=0).items()}
    v = torch.randn_like(inp["u"])
    f = lambda uu: selective_scan_fn(uu, inp["delta"], inp["A"], inp["B"], inp["C"], inp["D"], None, inp["delta_bias"], True)  # noqa: E731
    lhs = f(2 * inp["u"] + v)
    rhs = 2 * f(inp["u"]) + f(v)
    err = ((lhs - rhs).norm() / rhs.norm()).item()
    assert err < 5e-6


def test_cpu_tensor_is_rejected():
    from mamba_unet_amd import selective_scan_fn
    from oracle.ss2d_ref import make_scan_inputs
    inp = make_scan_inputs(1, 8, 16)
    with pytest.raises(RuntimeError):
        selective_scan_fn(inp["u"], inp["delta"], inp["A"], inp["B"], inp["C"], inp["D"], None, inp["delta_bias"], True)

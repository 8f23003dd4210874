"""CPU tests that pin the oracle (no GPU needed).

1. restated recurrence == independent closed-form fp64 evaluation (not self-referential);
2. oracle SS2D glue == outputs of the reference's own SS2D stored in tests/golden/*.npz;
3. closed-form index maps == tables measured from the reference's forward_corev0;
4. plain-C restatement (fwd + hand-derived bwd) == torch restatement and its autograd.
"""
import hashlib
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2, rel_max
from oracle import c_ref
from oracle.selective_scan_ref import selective_scan_closed_form, selective_scan_ref
from oracle.ss2d_ref import (cross_merge_index, cross_merge_ref, cross_scan_index, cross_scan_ref,
                             make_scan_inputs, ss2d_core_ref, ss2d_forward_ref)


def _load(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


@pytest.mark.parametrize("grouped,with_z,softplus", [(True, False, True), (False, True, False), (True, True, True)])
def test_recurrence_matches_closed_form(grouped, with_z, softplus):
    inp = make_scan_inputs(2, 8, 23, dstate=5, ngroups=4 if grouped else 1, seed=3, integer_A=False)
    if not grouped:
        inp["B"], inp["C"] = inp["B"][:, 0], inp["C"][:, 0]
    if not softplus:  # raw delta must stay positive or exp(delta*A) blows up
        inp["delta"], inp["delta_bias"] = inp["delta"].abs() * 0.2, inp["delta_bias"].abs() * 0.01
    z = torch.randn(2, 8, 23, generator=torch.Generator().manual_seed(9)) if with_z else None
    y64, last64 = selective_scan_closed_form(inp["u"], inp["delta"], inp["A"], inp["B"], inp["C"], inp["D"], z,
                                             inp["delta_bias"], softplus)
    y, last = selective_scan_ref(inp["u"], inp["delta"], inp["A"], inp["B"], inp["C"], inp["D"], z,
                                 inp["delta_bias"], softplus, return_last_state=True)
    assert rel_l2(y, y64) < 2e-6 and rel_max(y, y64) < 2e-6
    assert rel_l2(last, last64) < 2e-6
    y_hi = selective_scan_ref(inp["u"], inp["delta"], inp["A"], inp["B"], inp["C"], inp["D"], z,
                              inp["delta_bias"], softplus, compute_dtype=torch.float64)
    assert rel_l2(y_hi.double(), y64) < 1e-6  # cast back to fp32 only


@pytest.mark.parametrize("name", ["ss2d_2x8x5x7_s0.npz", "ss2d_1x16x8x8_s1.npz", "ss2d_2x96x14x14_s0.npz"])
def test_ss2d_oracle_matches_reference_golden(name):
    g = _load(name)
    sd = {k[3:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("sd.")}
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    cap = {}
    y = ss2d_forward_ref(x, params, capture=cap)
    assert rel_max(y.detach(), g["y"]) < 1e-5
    core = ss2d_core_ref(torch.from_numpy(g["core_in"]), sd["x_proj_weight"], sd["dt_projs_weight"],
                         sd["dt_projs_bias"], sd["A_logs"], sd["Ds"], sd["out_norm.weight"], sd["out_norm.bias"])
    assert rel_max(core, g["core_out"]) < 1e-5
    (y * torch.from_numpy(g["g"])).sum().backward()
    assert rel_max(x.grad, g["dx"]) < 1e-4
    for k, p in params.items():
        assert rel_max(p.grad, g["grad." + k]) < 1e-4, k
    if "scan.u" in g:
        for k in ("u", "delta", "A", "B", "C", "D", "delta_bias", "out_y"):
            assert rel_max(cap[k].detach().reshape(g["scan." + k].shape), g["scan." + k]) < 1e-5, k


def test_index_tables_match_reference():
    g = _load("index_tables.npz")
    for (H, W) in [(3, 5), (5, 3), (7, 7), (4, 6)]:
        assert np.array_equal(cross_scan_index(H, W), g[f"src_{H}x{W}"])
        assert np.array_equal(cross_merge_index(H, W), g[f"dst_{H}x{W}"])
    for (H, W) in [(56, 56), (128, 128)]:
        for nm, fn in (("src", cross_scan_index), ("dst", cross_merge_index)):
            digest = np.frombuffer(hashlib.sha256(fn(H, W).tobytes()).digest(), dtype=np.uint8)
            assert np.array_equal(digest, g[f"{nm}_{H}x{W}_sha256"])


@pytest.mark.parametrize("H,W", [(3, 5), (6, 4), (7, 7)])
def test_cross_scan_merge_tensor_ops_match_index_maps(H, W):
    x = torch.randn(2, 3, H, W)
    xs = cross_scan_ref(x)
    src = torch.from_numpy(cross_scan_index(H, W))
    flat = x.reshape(2, 3, H * W)
    for k in range(4):
        assert torch.equal(xs[:, k], flat[:, :, src[k]])
    oy = torch.randn(2, 4, 3, H * W)
    dst = torch.from_numpy(cross_merge_index(H, W))
    want = sum(oy[:, k][:, :, dst[k]] for k in range(4))
    assert torch.allclose(cross_merge_ref(oy, H, W), want, atol=1e-6)
    # merge is the adjoint of scan: <scan(x), oy> == <x, merge(oy)>
    assert torch.allclose((xs * oy).sum(), (flat * cross_merge_ref(oy, H, W)).sum(), rtol=1e-4)


@pytest.mark.parametrize("with_z,softplus,G", [(False, True, 4), (True, True, 2), (True, False, 1)])
def test_c_oracle_matches_torch_oracle(with_z, softplus, G):
    inp = make_scan_inputs(2, 8, 37, dstate=16, ngroups=G, seed=5, integer_A=False)
    z = torch.randn(2, 8, 37, generator=torch.Generator().manual_seed(2)) if with_z else None
    if not softplus:
        inp["delta"], inp["delta_bias"] = inp["delta"].abs() * 0.2, inp["delta_bias"].abs() * 0.01
    leaves = {k: v.clone().requires_grad_(True) for k, v in inp.items()}
    zl = z.clone().requires_grad_(True) if with_z else None
    y, last = selective_scan_ref(leaves["u"], leaves["delta"], leaves["A"], leaves["B"], leaves["C"], leaves["D"],
                                 zl, leaves["delta_bias"], softplus, return_last_state=True)
    dout = torch.randn(y.shape, generator=torch.Generator().manual_seed(1))
    (y * dout).sum().backward()
    npi = {k: v.numpy() for k, v in inp.items()}
    zn = z.numpy() if with_z else None
    yc, lastc = c_ref.scan_fwd(npi["u"], npi["delta"], npi["A"], npi["B"], npi["C"], npi["D"], zn,
                               npi["delta_bias"], softplus, return_last_state=True)
    assert rel_max(yc, y.detach()) < 2e-6 and rel_max(lastc, last.detach()) < 2e-6
    gr = c_ref.scan_bwd(npi["u"], npi["delta"], npi["A"], npi["B"], npi["C"], npi["D"], zn, npi["delta_bias"],
                        dout.numpy(), softplus)
    names = {"du": "u", "ddelta": "delta", "dA": "A", "dB": "B", "dC": "C", "dD": "D", "ddelta_bias": "delta_bias"}
    for k, src in names.items():
        assert rel_max(gr[k], leaves[src].grad) < 1e-5, k
    if with_z:
        assert rel_max(gr["dz"], zl.grad) < 1e-5

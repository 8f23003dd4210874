"""Edge cases of the selective scan through the C ABI: minimal sizes, chunk boundaries (checkpoint spacing 128),
ragged channel groups, and every scan shape MambaUNet issues (224x224 and 512x512), forward and all gradients
against the plain-C oracle (oracle/scan_ref.c, fp64 accumulation of the reduced gradients)."""
import numpy as np
import pytest
import torch

from conftest import rel_l2, rel_max

pytestmark = pytest.mark.gpu

L2_TOL = 1e-5      # north_star fp32 gate
MAX_TOL = 5e-5


def _gpu_fwd_bwd(inp, dout):
    from mamba_unet_amd import selective_scan_fn
    leaves = {k: v.cuda().requires_grad_(True) for k, v in inp.items()}
    y, last = selective_scan_fn(leaves["u"], leaves["delta"], leaves["A"], leaves["B"], leaves["C"], leaves["D"], None,
                                leaves["delta_bias"], True, return_last_state=True)
    y.backward(dout.cuda())
    torch.cuda.synchronize()
    return y.detach().cpu(), last.cpu(), {k: v.grad.cpu() for k, v in leaves.items()}


def _c_fwd_bwd(inp, dout):
    from oracle import c_ref
    n = {k: v.numpy() for k, v in inp.items()}
    y, last = c_ref.scan_fwd(n["u"], n["delta"], n["A"], n["B"], n["C"], n["D"], None, n["delta_bias"], True,
                             return_last_state=True)
    g = c_ref.scan_bwd(n["u"], n["delta"], n["A"], n["B"], n["C"], n["D"], None, n["delta_bias"], dout.numpy(), True)
    return y, last, g


@pytest.mark.parametrize("batch,dim,L,N,G", [
    (1, 1, 1, 1, 1),          # the smallest legal call
    (1, 4, 1, 16, 4),         # L = 1
    (2, 8, 127, 16, 2),       # one position short of a checkpoint chunk
    (2, 8, 128, 16, 2),       # exactly one chunk
    (2, 8, 129, 16, 2),       # one position into the second chunk
    (1, 12, 256, 3, 3),       # exactly two chunks, odd dstate, 4 channels per group (one DPP quad, rows full)
    (1, 15, 260, 16, 3),      # 5 channels per group: a quad with masked rows
    (3, 100, 77, 2, 1),       # 100 channels in one group: 4 rounds worth of quads in the backward
    # MambaUNet scan shapes, batch 2 (SURVEY.md 8 stage table): 224x224 stages 1-3 and the 512x512 stage 0 length
    (2, 1536, 784, 16, 4),
    (2, 3072, 196, 16, 4),
    (2, 6144, 49, 16, 4),
    (1, 64, 16384, 16, 4),
])
def test_scan_fwd_bwd_edge_shapes_against_c_oracle(batch, dim, L, N, G):
    from oracle.ss2d_ref import make_scan_inputs
    inp = make_scan_inputs(batch, dim, L, dstate=N, ngroups=G, seed=dim + L, integer_A=False)
    dout = torch.randn(batch, dim, L, generator=torch.Generator().manual_seed(1))
    y, last, g = _gpu_fwd_bwd(inp, dout)
    yr, lastr, gr = _c_fwd_bwd(inp, dout)
    assert rel_l2(y, yr) < L2_TOL and rel_max(y, yr) < MAX_TOL
    assert rel_l2(last, lastr) < L2_TOL
    names = {"du": "u", "ddelta": "delta", "dA": "A", "dB": "B", "dC": "C", "dD": "D", "ddelta_bias": "delta_bias"}
    for k, src in names.items():
        assert rel_l2(g[src], gr[k]) < L2_TOL, (k, rel_l2(g[src], gr[k]))
        assert rel_max(g[src], gr[k]) < MAX_TOL, (k, rel_max(g[src], gr[k]))


@pytest.mark.parametrize("batch,dim,L,N,G", [
    (2, 6144, 56, 16, 4),     # the 7 x 7 stage with its rows padded to 56 (SS2D._scan_core_dirs)
    (3, 160, 64, 16, 4),      # 40 channels per group: a full workgroup of 32 and a ragged one; exactly half a chunk
    (1, 128, 8, 16, 4),       # one lane per channel
    (2, 256, 28, 16, 2),
    (2, 132, 60, 16, 4),      # 33 channels per group: a workgroup with a single channel
    (1, 64, 68, 16, 2),       # just past the packed form's limit: one channel per row again
])
def test_short_sequences_two_channels_per_row_against_c_oracle(batch, dim, L, N, G):
    """Sequences of at most 64 positions run two channels per DPP row in the forward (scan_fwd_kernel<..., PACK>: the row scan is
    cut between lanes 7 and 8); the call without last_state is the one SS2D issues.  Forward and all gradients against the
    plain-C oracle."""
    from mamba_unet_amd import selective_scan_fn
    from oracle import c_ref
    from oracle.ss2d_ref import make_scan_inputs
    inp = make_scan_inputs(batch, dim, L, dstate=N, ngroups=G, seed=dim + L, integer_A=False)
    dout = torch.randn(batch, dim, L, generator=torch.Generator().manual_seed(2))
    leaves = {k: v.cuda().requires_grad_(True) for k, v in inp.items()}
    y = selective_scan_fn(leaves["u"], leaves["delta"], leaves["A"], leaves["B"], leaves["C"], leaves["D"], None, leaves["delta_bias"], True)
    y.backward(dout.cuda())
    n = {k: v.numpy() for k, v in inp.items()}
    yr = c_ref.scan_fwd(n["u"], n["delta"], n["A"], n["B"], n["C"], n["D"], None, n["delta_bias"], True)
    gr = c_ref.scan_bwd(n["u"], n["delta"], n["A"], n["B"], n["C"], n["D"], None, n["delta_bias"], dout.numpy(), True)
    assert rel_l2(y.detach().cpu(), yr) < L2_TOL and rel_max(y.detach().cpu(), yr) < MAX_TOL
    names = {"du": "u", "ddelta": "delta", "dA": "A", "dB": "B", "dC": "C", "dD": "D", "ddelta_bias": "delta_bias"}
    for k, src in names.items():
        assert rel_l2(leaves[src].grad.cpu(), gr[k]) < L2_TOL, (k, rel_l2(leaves[src].grad.cpu(), gr[k]))


def test_scan_rejects_bad_arguments():
    from mamba_unet_amd import selective_scan_fn
    u = torch.randn(1, 8, 16, device="cuda")
    A = -torch.rand(8, 16, device="cuda")
    Bm = torch.randn(1, 3, 16, 16, device="cuda")           # 8 channels not divisible by 3 groups
    with pytest.raises(ValueError):
        selective_scan_fn(u, u, A, Bm, Bm)
    with pytest.raises(ValueError):
        selective_scan_fn(u, u[:, :, :8], A, Bm[:, :1], Bm[:, :1])     # delta shape mismatch
    A32 = -torch.rand(8, 32, device="cuda")                   # dstate 32 > 16: rejected by the library, not truncated
    B32 = torch.randn(1, 1, 32, 16, device="cuda")
    with pytest.raises(ValueError):
        selective_scan_fn(u, u, A32, B32, B32)


def test_empty_batch_is_rejected_not_launched():
    from mamba_unet_amd import selective_scan_fn
    u = torch.empty(0, 8, 16, device="cuda")
    A = -torch.rand(8, 16, device="cuda")
    Bm = torch.empty(0, 1, 16, 16, device="cuda")
    with pytest.raises(ValueError):
        selective_scan_fn(u, u, A, Bm, Bm)

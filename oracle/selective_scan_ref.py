"""ORACLE (test infrastructure, not product code) -- sequential S6 selective scan on the CPU.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Provenance / pin status
-----------------------
The algorithm lives in the third-party CUDA package `mamba_ssm` (version unpinned by the
reference: README.md:101-111 says "install mamba", README.md:277 recommends
causal-conv1d 1.2.2.post1), which is NOT vendored in /root/reference and not installable here.
The reference imports it at code/networks/mamba_sys.py:17-20 and calls it at :436-442.
The body of upstream `selective_scan_ref` is quoted, in four fragments, inside docstrings at
code/networks/mamba_sys.py:62-83, :92-105, :115-131 and :139-144; this file restates those four
fragments in order.  The reference ships no golden vector for the scan itself, so for the scan
kernel alone the pin is: (1) this textual restatement, (2) an independent closed-form float64
evaluation (`selective_scan_closed_form` below, checked in tests/test_oracle.py) and (3) the
reference's own SS2D glue run on top of it (tests/golden/*.npz made by tests/golden/make_golden.py,
which imports the reference's mamba_sys.py and injects this function as `selective_scan_fn`).
=> scan kernel: "parity unpinned by the reference's own tests" (it has none); SS2D glue: pinned by
outputs of the reference code run in this container.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def selective_scan_ref(u, delta, A, B, C, D=None, z=None, delta_bias=None,
                       delta_softplus=False, return_last_state=False, *, compute_dtype=torch.float32):
    """Sequential reference.  Shapes (mamba_sys.py:31-39):
        u, delta (b, d, l); A (d, n) real; B, C (b, n, l) | (b, g, n, l) | (d, n);
        D (d); z (b, d, l); delta_bias (d).
    Returns out (b, d, l) in u.dtype [, last_state (b, d, n)].

    `compute_dtype=torch.float64` gives the high-precision variant used for error budgeting.
    """
    in_dtype = u.dtype
    cd = compute_dtype
    # fragment 1 -- mamba_sys.py:63-82: promote, add bias, softplus, zero initial state
    u_ = u.to(cd)
    dt = delta.to(cd)
    if delta_bias is not None:
        dt = dt + delta_bias.to(cd)[:, None]
    if delta_softplus:
        dt = F.softplus(dt)
    bsz, dim, seqlen = u_.shape
    nstate = A.shape[1]
    if A.is_complex():
        raise NotImplementedError("complex A is not on the Mamba-UNet path (mamba_sys.py:57)")
    A_ = A.to(cd)
    B_ = B.to(cd)
    C_ = C.to(cd)
    var_B = B_.dim() >= 3
    var_C = C_.dim() >= 3

    # fragment 2 -- mamba_sys.py:93-104: discretise; grouped B/C are repeated over the group
    dA = torch.exp(dt[:, :, :, None] * A_[None, :, None, :])            # (b d l n)
    if not var_B:
        dBu = (dt * u_)[:, :, :, None] * B_[None, :, None, :]            # (b d l n)
    else:
        if B_.dim() == 4:
            B_ = B_.repeat_interleave(dim // B_.shape[1], dim=1)         # (b d n l)
            dBu = (dt * u_)[:, :, :, None] * B_.permute(0, 1, 3, 2)
        else:
            dBu = (dt * u_)[:, :, :, None] * B_.permute(0, 2, 1)[:, None]
    if var_C and C_.dim() == 4:
        C_ = C_.repeat_interleave(dim // C_.shape[1], dim=1)             # (b d n l)

    # fragment 3 -- mamba_sys.py:116-130: the recurrence, one time step after another.
    # `unbind` instead of `x[:, :, i]` keeps autograd linear in l; the values are identical.
    state = A_.new_zeros((bsz, dim, nstate))
    dA_steps = dA.unbind(2)
    dBu_steps = dBu.unbind(2)
    if var_C:
        C_steps = C_.unbind(-1)                                         # each (b n) or (b d n)
    ys = []
    for i in range(seqlen):
        state = dA_steps[i] * state + dBu_steps[i]
        if not var_C:
            y = (state * C_[None]).sum(-1)
        elif C_.dim() == 3:
            y = (state * C_steps[i][:, None, :]).sum(-1)
        else:
            y = (state * C_steps[i]).sum(-1)
        ys.append(y)
    last_state = state
    y = torch.stack(ys, dim=2)                                          # (b d l)

    # fragment 4 -- mamba_sys.py:140-143: skip connection, gate, cast back
    out = y if D is None else y + u_ * D.to(cd)[:, None]
    if z is not None:
        out = out * F.silu(z.to(cd))
    out = out.to(in_dtype)
    return (out, last_state) if return_last_state else out


def selective_scan_closed_form(u, delta, A, B, C, D=None, z=None, delta_bias=None,
                               delta_softplus=False):
    """Independent float64 evaluation WITHOUT a recurrence, for tiny l only:
        h_l = sum_{j<=l} exp(A * (S_l - S_j)) * delta_j * B_j * u_j,   S = cumsum(delta)
    (the product of exp(delta_i A) over j<i<=l is exp(A * sum delta_i)).  Used to make sure the
    restated recurrence above is not self-referential."""
    f8 = torch.float64
    u_, dt = u.to(f8), delta.to(f8)
    if delta_bias is not None:
        dt = dt + delta_bias.to(f8)[:, None]
    if delta_softplus:
        dt = torch.where(dt > 20, dt, torch.log1p(torch.exp(dt)))
    b, d, l = u_.shape
    A_ = A.to(f8)
    Bm, Cm = B.to(f8), C.to(f8)
    if Bm.dim() == 3:
        Bm = Bm[:, None]
    if Cm.dim() == 3:
        Cm = Cm[:, None]
    Bm = Bm.repeat_interleave(d // Bm.shape[1], dim=1)                  # (b d n l)
    Cm = Cm.repeat_interleave(d // Cm.shape[1], dim=1)
    S = dt.cumsum(-1)                                                   # (b d l)
    diff = S[:, :, :, None] - S[:, :, None, :]                          # (b d l j) = S_l - S_j
    mask = torch.tril(torch.ones(l, l, dtype=torch.bool))
    w = torch.exp(A_[None, :, :, None, None] * diff[:, :, None])        # (b d n l j)
    w = torch.where(mask, w, torch.zeros((), dtype=f8))
    src = (dt * u_)[:, :, None, :] * Bm                                 # (b d n j)
    h = (w * src[:, :, :, None, :]).sum(-1)                             # (b d n l)
    y = (h * Cm).sum(2)
    if D is not None:
        y = y + u_ * D.to(f8)[:, None]
    if z is not None:
        zz = z.to(f8)
        y = y * zz * torch.sigmoid(zz)
    return y, h[..., -1]

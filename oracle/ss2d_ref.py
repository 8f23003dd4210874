"""ORACLE (test infrastructure, not product code) -- CPU restatement of the SS2D glue around the scan.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Follows code/networks/mamba_sys.py of the reference:
  cross-scan   :403-404      x_proj/dt_proj :407-420     casts/params :423-431
  scan call    :436-442      cross-merge    :445-449     out_norm     :452
  SS2D.forward :552-565      depthwise conv + SiLU :299-308, :558-559
Pinned by tests/golden/ss2d_*.npz, which hold outputs of the reference's own SS2D (imported from
/root/reference in the authoring container by tests/golden/make_golden.py) with
oracle.selective_scan_ref injected as `selective_scan_fn`.
"""
from __future__ import annotations


import numpy as np
import torch
import torch.nn.functional as F

from .selective_scan_ref import selective_scan_ref


# ----------------------------------------------------------------------------- index maps
def cross_scan_index(H: int, W: int) -> np.ndarray:
    """int64 table src[k, l]: xs[b, k, d, l] = x[b, d].reshape(-1)[src[k, l]]  (mamba_sys.py:403-404).

    k0 row-major, k1 column-major (transpose(2,3) then flatten), k2/k3 = flips of k0/k1."""
    L = H * W
    l = np.arange(L, dtype=np.int64)
    src = np.empty((4, L), dtype=np.int64)
    src[0] = l
    src[1] = (l % H) * W + l // H          # position l of the (W,H)-flattened transpose
    src[2] = src[0][::-1]
    src[3] = src[1][::-1]
    return src


def cross_merge_index(H: int, W: int) -> np.ndarray:
    """int64 table dst[k, p]: y[b, d, p] = sum_k out_y[b, k, d, dst[k, p]]  (mamba_sys.py:445-448)."""
    L = H * W
    p = np.arange(L, dtype=np.int64)
    h, w = p // W, p % W
    dst = np.empty((4, L), dtype=np.int64)
    dst[0] = p
    dst[1] = w * H + h
    dst[2] = L - 1 - p
    dst[3] = L - 1 - (w * H + h)
    return dst


def cross_scan_ref(x: torch.Tensor) -> torch.Tensor:
    """(B, D, H, W) -> (B, 4, D, L), by the tensor ops of mamba_sys.py:403-404."""
    B, D, H, W = x.shape
    L = H * W
    hw = x.reshape(B, D, L)
    wh = x.transpose(2, 3).contiguous().reshape(B, D, L)
    both = torch.stack([hw, wh], dim=1)
    return torch.cat([both, both.flip(-1)], dim=1)


def cross_merge_ref(out_y: torch.Tensor, H: int, W: int) -> torch.Tensor:
    """(B, 4, D, L) -> (B, D, L) summed over directions in row-major order (mamba_sys.py:445-448)."""
    B, K, D, L = out_y.shape
    inv = out_y[:, 2:4].flip(-1)
    wh = out_y[:, 1].reshape(B, D, W, H).transpose(2, 3).contiguous().reshape(B, D, L)
    invwh = inv[:, 1].reshape(B, D, W, H).transpose(2, 3).contiguous().reshape(B, D, L)
    return out_y[:, 0] + inv[:, 0] + wh + invwh


# ----------------------------------------------------------------------------- conv + SiLU
def dwconv3x3_silu_ref(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None) -> torch.Tensor:
    """x (B, D, H, W); weight (D, 1, 3, 3); SiLU(conv2d(groups=D, padding=1)) (mamba_sys.py:299-308, 559)."""
    return F.silu(F.conv2d(x, weight, bias, padding=1, groups=x.shape[1]))


# ----------------------------------------------------------------------------- SS2D core / forward
def ss2d_core_ref(x, x_proj_weight, dt_projs_weight, dt_projs_bias, A_logs, Ds,
                  out_norm_weight, out_norm_bias, eps=1e-5, scan_fn=selective_scan_ref,
                  capture: dict | None = None):
    """forward_corev0 (mamba_sys.py:396-461): x (B, D, H, W) -> (B, H, W, D)."""
    B, D, H, W = x.shape
    L, K = H * W, 4
    N = A_logs.shape[1]
    R = dt_projs_weight.shape[2]
    xs = cross_scan_ref(x)                                                   # :403-404
    x_dbl = torch.einsum("bkdl,kcd->bkcl", xs, x_proj_weight)                # :407
    dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)                       # :417
    dts = torch.einsum("bkrl,kdr->bkdl", dts, dt_projs_weight)               # :420
    u = xs.float().reshape(B, K * D, L)                                      # :423-426
    delta = dts.contiguous().float().reshape(B, K * D, L)
    Bs = Bs.float().reshape(B, K, N, L)
    Cs = Cs.float().reshape(B, K, N, L)
    As = -torch.exp(A_logs.float()).reshape(K * D, N)                        # :429-431
    Dv = Ds.float().reshape(-1)
    bias = dt_projs_bias.float().reshape(-1)
    if capture is not None:
        capture.update(u=u, delta=delta, A=As, B=Bs, C=Cs, D=Dv, delta_bias=bias)
    out_y = scan_fn(u, delta, As, Bs, Cs, Dv, z=None, delta_bias=bias,
                    delta_softplus=True, return_last_state=False).reshape(B, K, D, L)   # :436-442
    if capture is not None:
        capture.update(out_y=out_y)
    y = cross_merge_ref(out_y, H, W)                                         # :445-448
    y = y.transpose(1, 2).contiguous().reshape(B, H, W, D)                   # :449
    y = F.layer_norm(y, (D,), out_norm_weight, out_norm_bias, eps)           # :452
    return y.to(x.dtype)


def ss2d_forward_ref(x, sd: dict, scan_fn=selective_scan_ref, capture: dict | None = None):
    """SS2D.forward (mamba_sys.py:552-565) from a state dict with the reference's parameter names."""
    xz = F.linear(x, sd["in_proj.weight"])                                   # :555
    xi, z = xz.chunk(2, dim=-1)                                              # :556
    xi = xi.permute(0, 3, 1, 2).contiguous()                                 # :558
    xi = dwconv3x3_silu_ref(xi, sd["conv2d.weight"], sd.get("conv2d.bias"))  # :559
    y = ss2d_core_ref(xi, sd["x_proj_weight"], sd["dt_projs_weight"], sd["dt_projs_bias"],
                      sd["A_logs"], sd["Ds"], sd["out_norm.weight"], sd["out_norm.bias"],
                      scan_fn=scan_fn, capture=capture)                      # :560
    y = y * F.silu(z)                                                        # :561
    return F.linear(y, sd["out_proj.weight"])                                # :562


# ----------------------------------------------------------------------------- synthetic inputs
# The generator is not checker logic: it lives in the neutral mamba_unet_amd.synthetic (bench.py and tools/ use it
# without importing this package) and is re-exported here for the tests that already import it from the oracle.
from mamba_unet_amd.synthetic import make_scan_inputs  # noqa: E402,F401

"""`selective_scan_fn` -- drop-in for mamba_ssm.ops.selective_scan_interface.selective_scan_fn as the
reference imports it (code/networks/mamba_sys.py:17-20) and calls it (:436-442).

Same signature, argument meaning and return convention as upstream; every FLOP runs in the HIP
kernels of libss2d_hip.so through the C ABI of include/ss2d.h.  CPU tensors are rejected: there is
no fallback path."""
from __future__ import annotations

import os as _os

import torch

from . import _lib
from .ss2d_ops import chan_gemm, chan_wgrad

_F32, _BF16 = 0, 1          # include/ss2d.h: SS2D_DTYPE_F32 / SS2D_DTYPE_BF16


_stream = _lib.stream


def _bwd_buffers(lib, dev, batch, dim, L, dstate, G, has_z, shA, shD, shbias, shB, shC):
    """Gradient accumulators + workspace of one scan backward.  With a workspace (16 states, no z, L % 4 == 0) the
    library WRITES dB / dC (partial tiles + one reduction pass): they need no zero-fill; dA / dD / dbias (and dB / dC
    on the atomic path) come zero-filled out of one buffer.  Returns dA, dD, dbias, dB, dC, ws, ws_bytes."""
    nbytes = 0 if has_z else int(lib.ss2d_scan_bwd_workspace_bytes(batch, dim, L, dstate, G))
    if nbytes:
        dA, dD, dbias = _lib.zeros_group(dev, shA, shD, shbias)
        dB = torch.empty(shB, device=dev, dtype=torch.float32)
        dC = torch.empty(shC, device=dev, dtype=torch.float32)
        ws = torch.empty(nbytes // 4, device=dev, dtype=torch.float32)
        return dA, dD, dbias, dB, dC, ws, nbytes
    dA, dD, dbias, dB, dC = _lib.zeros_group(dev, shA, shD, shbias, shB, shC)
    return dA, dD, dbias, dB, dC, None, 0


def _prep(t, name, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"selective_scan_fn: `{name}` must be a GPU tensor (no CPU path in mamba_unet_amd)")
    if t.dtype != dtype:
        t = t.to(dtype)
    t = t.contiguous()
    # the kernels' vector paths (and the workspace form of the backward, which writes dB / dC instead of accumulating them)
    # want 16-byte aligned bases: a view at an odd storage offset is copied rather than sent down a slower path
    return t.clone() if t.data_ptr() % 16 else t


class SelectiveScanFn(torch.autograd.Function):
    """autograd wrapper; keeps upstream's class name so `prim::PythonOp.SelectiveScanFn`
    (mamba_sys.py:864) still resolves."""

    @staticmethod
    def forward(ctx, u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                return_last_state=False):
        if A.is_complex():
            raise NotImplementedError("complex A is not supported (not on the Mamba-UNet path, mamba_sys.py:57)")
        if u.dim() != 3 or delta.shape != u.shape:
            raise ValueError(f"u/delta must be (batch, dim, L) and equal: {tuple(u.shape)} vs {tuple(delta.shape)}")
        in_dtype = u.dtype
        # storage type of u, delta, B, C, z, out and their gradients inside the kernels: bf16 stays bf16
        # (half the HBM bytes; arithmetic, A, D, delta_bias, the state checkpoints and dA/dB/dC/dD are fp32)
        io = torch.bfloat16 if in_dtype == torch.bfloat16 else torch.float32
        batch, dim, L = u.shape
        dstate = A.shape[1]
        if A.shape[0] != dim:
            raise ValueError(f"A must be (dim, dstate) = ({dim}, n), got {tuple(A.shape)}")
        if B.dim() not in (3, 4) or C.dim() not in (3, 4):
            raise NotImplementedError("only input-dependent B/C of shape (b, n, l) or (b, g, n, l) are supported")
        ctx.squeeze_B, ctx.squeeze_C = B.dim() == 3, C.dim() == 3
        B4 = B.unsqueeze(1) if B.dim() == 3 else B
        C4 = C.unsqueeze(1) if C.dim() == 3 else C
        G = B4.shape[1]
        if C4.shape != B4.shape or B4.shape != (batch, G, dstate, L):
            raise ValueError(f"B/C must be (batch, [groups,] dstate, L): {tuple(B.shape)} / {tuple(C.shape)}")
        if dim % G:
            raise ValueError(f"dim {dim} not divisible by groups {G}")
        for nm, v in (("D", D), ("delta_bias", delta_bias)):
            if v is not None and v.shape != (dim,):
                raise ValueError(f"{nm} must be ({dim},), got {tuple(v.shape)}")
        if z is not None and z.shape != u.shape:
            raise ValueError("z must have u's shape")

        u_, delta_ = _prep(u, "u", io), _prep(delta, "delta", io)
        A_, B_, C_ = _prep(A, "A"), _prep(B4, "B", io), _prep(C4, "C", io)
        D_, z_, bias_ = _prep(D, "D"), _prep(z, "z", io), _prep(delta_bias, "delta_bias")
        ctx.code = code = _BF16 if io == torch.bfloat16 else _F32
        needs_grad = any(t is not None and t.requires_grad for t in (u, delta, A, B, C, D, z, delta_bias))
        lib = _lib.load()
        out = torch.empty_like(u_)
        last = torch.empty((batch, dim, dstate), device=u_.device, dtype=torch.float32) if return_last_state else None
        ckpt = None
        if needs_grad:
            ckpt = torch.empty((batch, dim, lib.ss2d_scan_num_chunks(L), dstate), device=u_.device, dtype=torch.float32)
        with _lib.on_device(u_.device):
            rc = lib.ss2d_selective_scan_fwd(code, _lib.ptr(u_), _lib.ptr(delta_), _lib.ptr(A_), _lib.ptr(B_),
                                             _lib.ptr(C_), _lib.ptr(D_), _lib.ptr(z_), _lib.ptr(bias_),
                                             _lib.ptr(out), _lib.ptr(last), _lib.ptr(ckpt), batch, dim, L, dstate, G,
                                             int(bool(delta_softplus)), _stream())
        _lib.check(rc, "ss2d_selective_scan_fwd")
        ctx.delta_softplus = bool(delta_softplus)
        ctx.has = (D is not None, z is not None, delta_bias is not None)
        ctx.in_dtypes = (u.dtype, delta.dtype, A.dtype, B.dtype, C.dtype,
                         None if D is None else D.dtype, None if z is None else z.dtype,
                         None if delta_bias is None else delta_bias.dtype)
        ctx.save_for_backward(u_, delta_, A_, B_, C_, D_, z_, bias_, ckpt)
        out = out.to(in_dtype)
        if return_last_state:
            ctx.mark_non_differentiable(last)
            return out, last
        return out

    @staticmethod
    def backward(ctx, dout, *unused):
        u, delta, A, B, C, D, z, bias, ckpt = ctx.saved_tensors
        batch, dim, L = u.shape
        dstate, G = A.shape[1], B.shape[1]
        dout = _prep(dout, "dout", u.dtype)
        lib = _lib.load()
        du, ddelta = torch.empty_like(u), torch.empty_like(u)
        sh = lambda t: None if t is None else tuple(t.shape)  # noqa: E731
        dA, dD, dbias, dB, dC, ws, wsb = _bwd_buffers(lib, u.device, batch, dim, L, dstate, G, z is not None,
                                                       sh(A), sh(D), sh(bias), sh(B), sh(C))
        dz = torch.empty_like(z) if z is not None else None
        with _lib.on_device(u.device):
            rc = lib.ss2d_selective_scan_dir_bwd_ws(ctx.code, _lib.ptr(u), _lib.ptr(delta), _lib.ptr(A), _lib.ptr(B), _lib.ptr(C),
                                                    _lib.ptr(D), _lib.ptr(z), _lib.ptr(bias), _lib.ptr(dout), _lib.ptr(ckpt),
                                                    _lib.ptr(du), _lib.ptr(ddelta), _lib.ptr(dA), _lib.ptr(dB), _lib.ptr(dC),
                                                    _lib.ptr(dD), _lib.ptr(dz), _lib.ptr(dbias), batch, dim, L, dstate, G,
                                                    int(ctx.delta_softplus), G, 0, 0, _lib.ptr(ws), wsb, _stream())
        _lib.check(rc, "ss2d_selective_scan_dir_bwd_ws")
        if ctx.squeeze_B:
            dB = dB.squeeze(1)
        if ctx.squeeze_C:
            dC = dC.squeeze(1)
        dt = ctx.in_dtypes
        cast = lambda g, d: None if g is None else g.to(d)  # noqa: E731
        return (cast(du, dt[0]), cast(ddelta, dt[1]), cast(dA, dt[2]), cast(dB, dt[3]), cast(dC, dt[4]),
                cast(dD, dt[5]), cast(dz, dt[6]), cast(dbias, dt[7]), None, None)


def selective_scan_fn(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False):
    """u, delta (b, d, l); A (d, n); B, C (b, n, l) or (b, g, n, l); D, delta_bias (d); z (b, d, l).
    Returns out (b, d, l) in u's dtype [, last_state (b, d, n) fp32]."""
    return SelectiveScanFn.apply(u, delta, A, B, C, D, z, delta_bias, delta_softplus, return_last_state)


class ScanMergeFn(torch.autograd.Function):
    """Direction-aware selective scan + CrossMerge in one autograd node (mamba_sys.py:436-449 without the flips):
    the scan groups are ordered (row-major, row-major reversed, column-major, column-major reversed); reversed groups
    walk the un-flipped tensors backwards in place (rev_mask), a direction and its reversal share one copy of u and,
    in the backward, one copy of dout (u_groups = 2).  Returns the merged y (B, H, W, D).
    With `x` (B, D, H, W), the image u2 was scattered from, the gradient w.r.t. u goes straight to x as one 4-slot
    gather of the scan's du (instead of pair sum -> add to the x_proj gradient -> 2-slot gather), and u2 gets none."""

    REV_MASK, U_GROUPS = 0b1010, 2

    @staticmethod
    def forward(ctx, u2, delta, A, Bm, Cm, D, delta_bias, H, W, x=None):
        if not u2.is_cuda:
            raise RuntimeError("scan_merge_fn: GPU tensors required (no CPU path in mamba_unet_amd)")
        batch, dim, L = delta.shape
        G, N = Bm.shape[1], A.shape[1]
        if G != 4 or L != H * W or u2.shape != (batch, dim // 2, L) or Bm.shape != (batch, 4, N, L) or Cm.shape != Bm.shape:
            raise ValueError(f"scan_merge_fn: inconsistent shapes u {tuple(u2.shape)} delta {tuple(delta.shape)} B {tuple(Bm.shape)}")
        io = torch.bfloat16 if u2.dtype == torch.bfloat16 else torch.float32
        code = _BF16 if io == torch.bfloat16 else _F32
        u_, delta_ = _prep(u2, "u", io), _prep(delta, "delta", io)
        # B and C may be two row ranges of every block of one x_proj output (torch.split views): read in place
        bcs = Bm.stride(1)
        in_place = (Bm.dtype == io and Cm.dtype == io and Bm.stride() == Cm.stride() and Bm.stride(3) == 1
                    and Bm.stride(2) == L and Bm.stride(0) == 4 * bcs and bcs >= N * L and bcs % 4 == 0)
        if in_place:
            B_, C_ = Bm, Cm
        else:
            B_, C_, bcs = _prep(Bm, "B", io), _prep(Cm, "C", io), N * L
        A_, D_, bias_ = _prep(A, "A"), _prep(D, "D"), _prep(delta_bias, "delta_bias")
        needs_grad = any(t is not None and t.requires_grad for t in (u2, delta, A, Bm, Cm, D, delta_bias))
        lib = _lib.load()
        out = torch.empty((batch, dim, L), device=u_.device, dtype=io)
        ckpt = torch.empty((batch, dim, lib.ss2d_scan_num_chunks(L), N), device=u_.device, dtype=torch.float32) if needs_grad else None
        Dch = dim // 4
        y = torch.empty((batch, H, W, Dch), device=u_.device, dtype=io)
        with _lib.on_device(u_.device):
            rc = lib.ss2d_selective_scan_dir_fwd(code, _lib.ptr(u_), _lib.ptr(delta_), _lib.ptr(A_), _lib.ptr(B_), _lib.ptr(C_),
                                                 _lib.ptr(D_), None, _lib.ptr(bias_), _lib.ptr(out), None, _lib.ptr(ckpt),
                                                 batch, dim, L, N, 4, 1, ScanMergeFn.U_GROUPS, ScanMergeFn.REV_MASK, bcs, _stream())
            _lib.check(rc, "ss2d_selective_scan_dir_fwd")
            rc = lib.ss2d_dirs_gather(code, 1, _lib.ptr(out), _lib.ptr(y), batch, Dch, H, W, 1, _stream())
            _lib.check(rc, "ss2d_dirs_gather")
        if x is not None and tuple(x.shape) != (batch, Dch, H, W):
            raise ValueError(f"scan_merge_fn: x {tuple(x.shape)} is not (batch, d, H, W) = {(batch, Dch, H, W)}")
        ctx.save_for_backward(u_, delta_, A_, B_, C_, D_, bias_, ckpt)
        ctx.meta = (code, H, W, u2.dtype, delta.dtype, A.dtype, Bm.dtype, Cm.dtype, None if D is None else D.dtype,
                    None if delta_bias is None else delta_bias.dtype)
        ctx.x_dtype = None if x is None else x.dtype
        ctx.bcs = bcs
        return y

    @staticmethod
    def backward(ctx, dy):
        u, delta, A, Bm, Cm, D, bias, ckpt = ctx.saved_tensors
        code, H, W, *dts = ctx.meta
        batch, dim, L = delta.shape
        N, Dch = A.shape[1], dim // 4
        lib = _lib.load()
        g = dy.to(u.dtype).contiguous()
        dys2 = torch.empty((batch, 2, Dch, L), device=g.device, dtype=u.dtype)
        du, ddelta = torch.empty_like(delta), torch.empty_like(delta)
        sh = lambda t: None if t is None else tuple(t.shape)  # noqa: E731
        dA, dD, dbias, dB, dC, ws, wsb = _bwd_buffers(lib, g.device, batch, dim, L, N, 4, False, sh(A), sh(D), sh(bias), sh(Bm), sh(Cm))
        with _lib.on_device(g.device):
            rc = lib.ss2d_dirs_scatter(code, 2, _lib.ptr(g), _lib.ptr(dys2), batch, Dch, H, W, 1, _stream())
            _lib.check(rc, "ss2d_dirs_scatter")
            rc = lib.ss2d_selective_scan_dir_bwd_ws(code, _lib.ptr(u), _lib.ptr(delta), _lib.ptr(A), _lib.ptr(Bm), _lib.ptr(Cm),
                                                    _lib.ptr(D), None, _lib.ptr(bias), _lib.ptr(dys2), _lib.ptr(ckpt), _lib.ptr(du),
                                                    _lib.ptr(ddelta), _lib.ptr(dA), _lib.ptr(dB), _lib.ptr(dC), _lib.ptr(dD), None,
                                                    _lib.ptr(dbias), batch, dim, L, N, 4, 1, ScanMergeFn.U_GROUPS,
                                                    ScanMergeFn.REV_MASK, ctx.bcs, _lib.ptr(ws), wsb, _stream())
            _lib.check(rc, "ss2d_selective_scan_dir_bwd_ws")
            dx = None
            if ctx.x_dtype is not None:   # all four directions' du -> dx in one pass (the adjoint of the scatter)
                dx = torch.empty((batch, Dch, H, W), device=g.device, dtype=u.dtype)
                rc = lib.ss2d_dirs_gather(code, 1, _lib.ptr(du), _lib.ptr(dx), batch, Dch, H, W, 0, _stream())
                _lib.check(rc, "ss2d_dirs_gather")
        cast = lambda t, d: None if t is None else t.to(d)  # noqa: E731
        du2 = None
        if dx is None:
            du2 = du.view(batch, 2, 2, Dch, L).sum(2).view(batch, 2 * Dch, L)   # a direction and its reversal share u
        return (cast(du2, dts[0]), cast(ddelta, dts[1]), cast(dA, dts[2]), cast(dB, dts[3]), cast(dC, dts[4]),
                cast(dD, dts[5]), cast(dbias, dts[6]), None, None, cast(dx, ctx.x_dtype))


def scan_merge_fn(u2, delta, A, B, C, D, delta_bias, H, W, x=None):
    """u2 (b, 2*d, l): row-major and column-major copies of x; delta (b, 4*d, l), A (4*d, n), B, C (b, 4, n, l), D,
    delta_bias (4*d) in scan-group order (row-major, row-major reversed, column-major, column-major reversed), all in
    un-reversed memory order.  softplus(delta + bias) as in SS2D.  Returns y (b, H, W, d) = CrossMerge of the scans."""
    return ScanMergeFn.apply(u2, delta, A, B, C, D, delta_bias, H, W, x)


class ScanCoreFn(torch.autograd.Function):
    """dt_proj + direction-aware selective scan + CrossMerge on the x_proj output, one autograd node
    (mamba_sys.py:417-449 without the split copies, casts and flips): x_dbl (b, 4, R+2N, l) in scan-group order is read
    in place three times -- rows [0, R) by the dt_proj kernel, rows [R, R+N) / [R+N, R+2N) as B / C by the scan
    (bc_bg_stride) -- and its gradient is assembled in one pass from d_dts and the scan backward's fp32 dB / dC
    (ss2d_xdbl_grad_pack) instead of two cast passes and a concatenation.  Arguments as `scan_merge_fn`, with
    dt_w (4, d, R) in place of delta."""

    # class-wide switch (A/B runs; env SS2D_NO_FUSE_DELTA=1): compute delta inside the scan kernels for dt_rank 6 / 12
    FUSE_DELTA = not _os.environ.get("SS2D_NO_FUSE_DELTA")
    # (A/B; env SS2D_NO_PACKED_DXDBL=1): the backward's reduction pass writes dB | dC into the x_proj gradient's rows
    PACKED_DXDBL = not _os.environ.get("SS2D_NO_PACKED_DXDBL")

    @staticmethod
    def forward(ctx, u2, x_dbl, dt_w, A, D, delta_bias, H, W, x=None, holder=None, dtwT=None, dtw_io=None, x_channel_last=False):
        ctx.holder = holder
        ctx.x_cl = bool(x_channel_last)
        if not x_dbl.is_cuda:
            raise RuntimeError("scan_core_fn: GPU tensors required (no CPU path in mamba_unet_amd)")
        batch, K, Cx, L = x_dbl.shape
        Dch, R, N = dt_w.shape[1], dt_w.shape[2], A.shape[1]
        dim = 4 * Dch
        if K != 4 or Cx != R + 2 * N or L < H * W or u2.shape != (batch, 2 * Dch, L) or dt_w.shape[0] != 4 or A.shape[0] != dim:
            raise ValueError(f"scan_core_fn: inconsistent shapes u {tuple(u2.shape)} x_dbl {tuple(x_dbl.shape)} dt_w {tuple(dt_w.shape)}")
        if x is not None and tuple(x.shape) != ((batch, H, W, Dch) if x_channel_last else (batch, Dch, H, W)):
            raise ValueError(f"scan_core_fn: x {tuple(x.shape)} is not (batch, d, H, W) = {(batch, Dch, H, W)} (or channel-last with x_channel_last)")
        io = torch.bfloat16 if x_dbl.dtype == torch.bfloat16 else torch.float32
        code = _BF16 if io == torch.bfloat16 else _F32
        u_, xd, w32 = _prep(u2, "u", io), _prep(x_dbl, "x_dbl", io), _prep(dt_w, "dt_w")
        A_, D_, bias_ = _prep(A, "A"), _prep(D, "D"), _prep(delta_bias, "delta_bias")
        needs_grad = any(t is not None and t.requires_grad for t in (u2, x_dbl, dt_w, A, D, delta_bias, x))
        lib = _lib.load()
        dev = xd.device
        out = torch.empty((batch, dim, L), device=dev, dtype=io)
        ckpt = torch.empty((batch, dim, lib.ss2d_scan_num_chunks(L), N), device=dev, dtype=torch.float32) if needs_grad else None
        y = torch.empty((batch, H, W, Dch), device=dev, dtype=io)
        fused_delta = ScanCoreFn.FUSE_DELTA and R in (6, 12) and L % 4 == 0 and N == 16
        with _lib.on_device(dev):
            if fused_delta:
                # delta = dt_w . dts is computed inside the scan kernels (forward and backward): never materialised
                delta = None
                rc = lib.ss2d_selective_scan_core_fwd(code, _lib.ptr(u_), _lib.ptr(xd), _lib.ptr(w32), _lib.ptr(A_), _lib.ptr(D_),
                                                      _lib.ptr(bias_), _lib.ptr(out), _lib.ptr(ckpt), batch, dim, L, N, 4, R,
                                                      ScanMergeFn.U_GROUPS, ScanMergeFn.REV_MASK, _stream())
                _lib.check(rc, "ss2d_selective_scan_core_fwd")
            elif R in (6, 12) and L % 4 == 0:
                delta = torch.empty((batch, dim, L), device=dev, dtype=io)
                rc = lib.ss2d_dt_proj_fwd(code, _lib.ptr(xd), _lib.ptr(w32), _lib.ptr(delta), batch, 4, Dch, R, L, Cx * L, _stream())
                _lib.check(rc, "ss2d_dt_proj_fwd")
            else:   # other ranks (24 / 48 at stages 2 / 3): bf16 MFMA kernel over the dts rows in place, else the library GEMM
                delta = chan_gemm(dtw_io, xd[:, :, :R], padded=True) if (dtw_io is not None and dtw_io.dtype == io) else chan_gemm(w32, xd[:, :, :R])
                if delta is None:
                    with torch.autocast("cuda", enabled=False):
                        delta = torch.matmul(w32.to(io).unsqueeze(0), xd[:, :, :R])
                delta = delta.reshape(batch, dim, L)
            if not fused_delta:
                if (Cx * L) % 4 == 0:
                    B_, C_, bcs = xd[:, :, R:R + N], xd[:, :, R + N:], Cx * L
                else:
                    B_, C_, bcs = xd[:, :, R:R + N].contiguous(), xd[:, :, R + N:].contiguous(), N * L
                rc = lib.ss2d_selective_scan_dir_fwd(code, _lib.ptr(u_), _lib.ptr(delta), _lib.ptr(A_), _lib.ptr(B_), _lib.ptr(C_),
                                                     _lib.ptr(D_), None, _lib.ptr(bias_), _lib.ptr(out), None, _lib.ptr(ckpt),
                                                     batch, dim, L, N, 4, 1, ScanMergeFn.U_GROUPS, ScanMergeFn.REV_MASK, bcs, _stream())
                _lib.check(rc, "ss2d_selective_scan_dir_fwd")
            rc = lib.ss2d_dirs_gather_ld(code, 1, _lib.ptr(out), _lib.ptr(y), batch, Dch, H, W, 1, L, _stream())
            _lib.check(rc, "ss2d_dirs_gather_ld")
        ctx.fused_delta = fused_delta
        if dtwT is not None and dtwT.dtype != io:
            dtwT = None
        ctx.save_for_backward(u_, delta, A_, xd, w32, D_, bias_, ckpt, dtwT)
        ctx.meta = (code, H, W, R, u2.dtype, x_dbl.dtype, dt_w.dtype, A.dtype, None if D is None else D.dtype,
                    None if delta_bias is None else delta_bias.dtype, None if x is None else x.dtype)
        return y

    @staticmethod
    def backward(ctx, dy):
        u, delta, A, xd, w32, D, bias, ckpt, dtwT = ctx.saved_tensors
        code, H, W, R, u_dt, xdbl_dt, w_dt, A_dt, D_dt, b_dt, x_dt = ctx.meta
        batch, L = xd.shape[0], xd.shape[3]
        dim = A.shape[0]
        N, Dch, Cx = A.shape[1], dim // 4, xd.shape[2]
        io = xd.dtype
        lib = _lib.load()
        g = dy.to(io).contiguous()
        dev = g.device
        dys2 = (torch.empty if L == H * W else torch.zeros)((batch, 2, Dch, L), device=dev, dtype=io)   # padded rows: dy = 0 in the pad
        du = torch.empty((batch, dim, L), device=dev, dtype=io)
        ddelta = torch.empty_like(du)
        sh = lambda t: None if t is None else tuple(t.shape)  # noqa: E731
        packed = ctx.fused_delta and ScanCoreFn.PACKED_DXDBL     # dB | dC written straight into d_xdbl's rows by the reduction pass
        dA, dD, dbias, dB, dC, ws, wsb = _bwd_buffers(lib, dev, batch, dim, L, N, 4, False, sh(A), sh(D), sh(bias),
                                                       (0,) if packed else (batch, 4, N, L), (0,) if packed else (batch, 4, N, L))
        if ctx.fused_delta:
            B_ = C_ = None
        elif (Cx * L) % 4 == 0:
            B_, C_, bcs = xd[:, :, R:R + N], xd[:, :, R + N:], Cx * L
        else:
            B_, C_, bcs = xd[:, :, R:R + N].contiguous(), xd[:, :, R + N:].contiguous(), N * L
        d_xdbl = torch.empty_like(xd)
        with _lib.on_device(dev):
            rc = lib.ss2d_dirs_scatter_ld(code, 2, _lib.ptr(g), _lib.ptr(dys2), batch, Dch, H, W, 1, L, _stream())
            _lib.check(rc, "ss2d_dirs_scatter_ld")
            if packed:
                rc = lib.ss2d_selective_scan_core_bwd_packed(code, _lib.ptr(u), _lib.ptr(xd), _lib.ptr(w32), _lib.ptr(A), _lib.ptr(D),
                                                             _lib.ptr(bias), _lib.ptr(dys2), _lib.ptr(ckpt), _lib.ptr(du), _lib.ptr(ddelta),
                                                             _lib.ptr(dA), _lib.ptr(d_xdbl), _lib.ptr(dD), _lib.ptr(dbias), batch, dim,
                                                             L, N, 4, R, ScanMergeFn.U_GROUPS, ScanMergeFn.REV_MASK, _lib.ptr(ws), wsb, _stream())
                _lib.check(rc, "ss2d_selective_scan_core_bwd_packed")
            elif ctx.fused_delta:
                rc = lib.ss2d_selective_scan_core_bwd(code, _lib.ptr(u), _lib.ptr(xd), _lib.ptr(w32), _lib.ptr(A), _lib.ptr(D),
                                                      _lib.ptr(bias), _lib.ptr(dys2), _lib.ptr(ckpt), _lib.ptr(du), _lib.ptr(ddelta),
                                                      _lib.ptr(dA), _lib.ptr(dB), _lib.ptr(dC), _lib.ptr(dD), _lib.ptr(dbias), batch, dim,
                                                      L, N, 4, R, ScanMergeFn.U_GROUPS, ScanMergeFn.REV_MASK, _lib.ptr(ws), wsb, _stream())
                _lib.check(rc, "ss2d_selective_scan_core_bwd")
            else:
                rc = lib.ss2d_selective_scan_dir_bwd_ws(code, _lib.ptr(u), _lib.ptr(delta), _lib.ptr(A), _lib.ptr(B_), _lib.ptr(C_),
                                                        _lib.ptr(D), None, _lib.ptr(bias), _lib.ptr(dys2), _lib.ptr(ckpt), _lib.ptr(du),
                                                        _lib.ptr(ddelta), _lib.ptr(dA), _lib.ptr(dB), _lib.ptr(dC), _lib.ptr(dD), None,
                                                        _lib.ptr(dbias), batch, dim, L, N, 4, 1, ScanMergeFn.U_GROUPS,
                                                        ScanMergeFn.REV_MASK, bcs, _lib.ptr(ws), wsb, _stream())
                _lib.check(rc, "ss2d_selective_scan_dir_bwd_ws")
            dx = du2 = None
            if x_dt is not None and ctx.holder is not None:
                # left for the scatter node's backward, which adds it to the x_proj input gradient in one 6-slot gather
                # (ss2d_ops.GradSlots); x gets no gradient from this node then
                ctx.holder.du = du
            elif x_dt is not None:      # all four directions' du -> dx in one pass (the adjoint of the scatter)
                dx = torch.empty((batch, H, W, Dch) if ctx.x_cl else (batch, Dch, H, W), device=dev, dtype=io)
                rc = lib.ss2d_dirs_gather_ld(code, 1, _lib.ptr(du), _lib.ptr(dx), batch, Dch, H, W, int(ctx.x_cl), L, _stream())
                _lib.check(rc, "ss2d_dirs_gather_ld")
            else:
                du2 = du.view(batch, 2, 2, Dch, L).sum(2).view(batch, 2 * Dch, L)
            # dt_proj backward on library GEMMs (K = d contraction for d_dts, K = l for the weight)
            with torch.autocast("cuda", enabled=False):
                dd4 = ddelta.view(batch, 4, Dch, L)
                # d_dts (b, 4, R, l) = dt_w^T . ddelta.  bf16: the own channel-major MFMA kernel; the library's batched GEMM first
                # materialises the weight once per batch entry (a (b, 4, R, d) copy: 5-26 us per call on top of the product)
                dts_rows = d_xdbl[:, :, :R] if packed else None          # the d_dts rows of the x_proj output's gradient, written in place
                d_dts = chan_gemm(dtwT, dd4, padded=True, out=dts_rows) if dtwT is not None else chan_gemm(w32.transpose(1, 2), dd4, out=dts_rows)
                if d_dts is None:
                    d_dts = torch.matmul(w32.to(io).transpose(1, 2).unsqueeze(0), dd4).contiguous()
                # (4, d, R): own MFMA kernel (bf16 25 / 19 / 13 us at stages 0-2 against the library's 61 / 42 / 28, fp32 -- as
                # three-way bf16 splits -- 65 / 37 / 23 against 66 / 66 / 57: the library runs B*4 small products and a sum
                # over the batch)
                d_w = chan_wgrad(dd4, xd[:, :, :R])
                if d_w is None:
                    d_w = torch.matmul(dd4, xd[:, :, :R].transpose(2, 3)).sum(0, dtype=torch.float32)
            if not packed:
                rc = lib.ss2d_xdbl_grad_pack(code, _lib.ptr(d_dts), _lib.ptr(dB), _lib.ptr(dC), _lib.ptr(d_xdbl), batch * 4, R, N, L, _stream())
                _lib.check(rc, "ss2d_xdbl_grad_pack")
            elif d_dts is not dts_rows:        # the GEMM did not take the in-place path
                dts_rows.copy_(d_dts)
        cast = lambda t, d: None if t is None else t.to(d)  # noqa: E731
        return (cast(du2, u_dt), cast(d_xdbl, xdbl_dt), cast(d_w, w_dt), cast(dA, A_dt), cast(dD, D_dt), cast(dbias, b_dt),
                None, None, cast(dx, x_dt), None, None, None, None)


def scan_core_fn(u2, x_dbl, dt_w, A, D, delta_bias, H, W, x=None, holder=None, dtwT=None, dtw_io=None, x_channel_last=False):
    """u2 (b, 2*d, l): row-major and column-major copies of x; x_dbl (b, 4, R+2N, l): the x_proj output with rows
    [dts | B | C] per scan group; dt_w (4, d, R); A (4*d, n), D, delta_bias (4*d) -- all in scan-group order
    (row-major, row-major reversed, column-major, column-major reversed).  delta = dt_w @ dts, softplus(delta + bias)
    as in SS2D.  l may exceed H*W: rows padded by the caller, u2 zero in the pad (SS2D._scan_core_dirs).  Returns y
    (b, H, W, d) = CrossMerge of the four scans.  dtwT (4, R, d) / dtw_io (4, d, R): optional prepared copies of dt_w in the
    activations' dtype with 16-byte padded rows (ss2d_ops.block_params), used in place by the dt_proj GEMMs."""
    return ScanCoreFn.apply(u2, x_dbl, dt_w, A, D, delta_bias, H, W, x, holder, dtwT, dtw_io, x_channel_last)

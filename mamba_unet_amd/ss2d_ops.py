"""autograd wrappers of the CrossScan / CrossMerge / depthwise-conv+SiLU kernels (C ABI: include/ss2d.h).

Reference lines they replace: cross-scan mamba_sys.py:403-404, cross-merge :445-449,
`act(conv2d(x.permute(0,3,1,2).contiguous()))` :558-559.  GPU tensors only (no CPU path)."""
from __future__ import annotations

import os as _os

import os

import torch

from . import _lib

_F32, _BF16 = 0, 1          # include/ss2d.h: SS2D_DTYPE_F32 / SS2D_DTYPE_BF16


def _io(t):
    """storage type inside the kernels: bf16 tensors stay bf16, everything else goes through fp32"""
    return (torch.bfloat16, _BF16) if t.dtype == torch.bfloat16 else (torch.float32, _F32)


_stream = _lib.stream


def _need_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: GPU tensor required (mamba_unet_amd has no CPU path)")


class _CrossScan(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need_gpu(x, "cross_scan")
        B, D, H, W = x.shape
        in_dtype = x.dtype
        io, code = _io(x)
        xf = x.to(io).contiguous()
        xs = torch.empty((B, 4, D, H * W), device=x.device, dtype=io)
        with _lib.on_device(x.device):
            rc = _lib.load().ss2d_cross_scan_fwd(code, _lib.ptr(xf), _lib.ptr(xs), B, D, H, W, _stream())
        _lib.check(rc, "ss2d_cross_scan_fwd")
        ctx.shape, ctx.in_dtype = (B, D, H, W), in_dtype
        return xs.to(in_dtype)

    @staticmethod
    def backward(ctx, dxs):
        B, D, H, W = ctx.shape
        io, code = _io(dxs)
        g = dxs.to(io).contiguous()
        dx = torch.empty((B, D, H, W), device=g.device, dtype=io)
        with _lib.on_device(g.device):
            rc = _lib.load().ss2d_cross_scan_bwd(code, _lib.ptr(g), _lib.ptr(dx), B, D, H, W, _stream())
        _lib.check(rc, "ss2d_cross_scan_bwd")
        return dx.to(ctx.in_dtype)


class _CrossMerge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ys, H, W, channel_last):
        _need_gpu(ys, "cross_merge")
        B, K, D, L = ys.shape
        if K != 4 or L != H * W:
            raise ValueError(f"cross_merge: expected (B, 4, D, {H * W}), got {tuple(ys.shape)}")
        in_dtype = ys.dtype
        io, code = _io(ys)
        yf = ys.to(io).contiguous()
        shape = (B, H, W, D) if channel_last else (B, D, L)
        y = torch.empty(shape, device=ys.device, dtype=io)
        with _lib.on_device(ys.device):
            rc = _lib.load().ss2d_cross_merge_fwd(code, _lib.ptr(yf), _lib.ptr(y), B, D, H, W, int(channel_last), _stream())
        _lib.check(rc, "ss2d_cross_merge_fwd")
        ctx.dims, ctx.in_dtype = (B, D, H, W, int(channel_last)), in_dtype
        return y.to(in_dtype)

    @staticmethod
    def backward(ctx, dy):
        B, D, H, W, cl = ctx.dims
        io, code = _io(dy)
        g = dy.to(io).contiguous()
        dys = torch.empty((B, 4, D, H * W), device=g.device, dtype=io)
        with _lib.on_device(g.device):
            rc = _lib.load().ss2d_cross_merge_bwd(code, _lib.ptr(g), _lib.ptr(dys), B, D, H, W, cl, _stream())
        _lib.check(rc, "ss2d_cross_merge_bwd")
        return dys.to(ctx.in_dtype), None, None, None


class GradHalves:
    """Shared by the three autograd nodes around one `xz = in_proj(x)` (split, dwconv, out_norm gate): the backward of the
    gate allocates ONE (.., 2D) gradient buffer and writes dz into its second half, the backward of the convolution writes
    dx into the first half, and the split's backward hands the buffer on as d(xz) -- instead of two tensors and the
    58 MB `cat` (stage 0, bf16) that autograd's split backward would make of them."""
    __slots__ = ("buf",)

    def __init__(self):
        self.buf = None


class _SplitHalves(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xz, holder):
        ctx.holder, ctx.D = holder, xz.shape[-1] // 2
        return xz[..., :ctx.D], xz[..., ctx.D:]

    @staticmethod
    def backward(ctx, gx, gz):
        D, buf = ctx.D, ctx.holder.buf
        ctx.holder.buf = None
        if (buf is not None and gx is not None and gz is not None and gx.dtype == buf.dtype and gz.dtype == buf.dtype
                and gx.shape[-1] == D and gz.shape[-1] == D and gx.stride() == buf.stride() and gz.stride() == buf.stride()
                and gx.data_ptr() == buf.data_ptr() and gz.data_ptr() == buf.data_ptr() + D * buf.element_size()):
            return buf, None                      # both halves were written in place
        if gx is None:
            gx = torch.zeros_like(gz)
        if gz is None:
            gz = torch.zeros_like(gx)
        return torch.cat([gx, gz], dim=-1), None


def split_halves(xz: torch.Tensor, holder: GradHalves):
    """xz[..., :D], xz[..., D:] (views), whose gradients can meet in one buffer (see GradHalves)."""
    return _SplitHalves.apply(xz, holder)


class _DwConvSiLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, channel_last_in, holder=None):
        ctx.holder = holder
        _need_gpu(x, "dwconv3x3_silu")
        in_dtype = x.dtype
        io, code = _io(x)
        if channel_last_in:
            B, H, W, D = x.shape
            xf = x.to(io)
            # accept a channel slice of a wider tensor (the x half of xz): unit channel stride, dense (b,h,w)
            if not (xf.stride(3) == 1 and xf.stride(1) == W * xf.stride(2) and xf.stride(0) == H * W * xf.stride(2)):
                xf = xf.contiguous()
            ldx = xf.stride(2)
        else:
            B, D, H, W = x.shape
            xf = x.to(io).contiguous()
            ldx = D
        if tuple(weight.shape) != (D, 1, 3, 3):
            raise ValueError(f"dwconv3x3_silu: weight must be ({D}, 1, 3, 3), got {tuple(weight.shape)}")
        wf = weight.float().contiguous()
        bf = None if bias is None else bias.float().contiguous()
        y = torch.empty((B, D, H, W), device=x.device, dtype=io)
        with _lib.on_device(x.device):
            rc = _lib.load().ss2d_dwconv3x3_silu_fwd(code, _lib.ptr(xf), _lib.ptr(wf), _lib.ptr(bf), _lib.ptr(y),
                                                     B, D, H, W, int(channel_last_in), ldx, _stream())
        _lib.check(rc, "ss2d_dwconv3x3_silu_fwd")
        ctx.save_for_backward(xf, wf, bf)
        ctx.meta = (B, D, H, W, int(channel_last_in), ldx, in_dtype, weight.dtype, None if bias is None else bias.dtype, code)
        return y.to(in_dtype)

    @staticmethod
    def backward(ctx, dy):
        xf, wf, bf = ctx.saved_tensors
        B, D, H, W, cl, ldx, in_dtype, w_dtype, b_dtype, code = ctx.meta
        g = dy.to(xf.dtype).contiguous()
        buf = ctx.holder.buf if ctx.holder is not None else None
        if cl and buf is not None and buf.dtype == xf.dtype and tuple(buf.shape) == (B, H, W, 2 * D) and D % 4 == 0:
            dx, lddx = buf[..., :D], 2 * D          # first half of the shared d(xz) buffer (GradHalves)
        else:
            dx, lddx = torch.empty((B, H, W, D) if cl else (B, D, H, W), device=g.device, dtype=xf.dtype), D
        dw, db = _lib.zeros_group(g.device, tuple(wf.shape), None if bf is None else tuple(bf.shape))
        with _lib.on_device(g.device):
            rc = _lib.load().ss2d_dwconv3x3_silu_bwd(code, _lib.ptr(xf), _lib.ptr(wf), _lib.ptr(bf), _lib.ptr(g),
                                                     _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db), B, D, H, W, cl, ldx, lddx,
                                                     _stream())
        _lib.check(rc, "ss2d_dwconv3x3_silu_bwd")
        return dx.to(in_dtype), dw.to(w_dtype), None if db is None else db.to(b_dtype), None, None


def _pixel_pitch(t, D):
    """Pitch (elements between pixels) of a (B, H, W, D) tensor whose pixels lie at a constant distance with the D channels
    contiguous -- a dense tensor or a channel slice of a wider one (the x half of xz) -- else None."""
    B, H, W, _ = t.shape
    ld = t.stride(2)
    if t.stride(3) == 1 and t.stride(1) == W * ld and t.stride(0) == H * W * ld and ld >= D and ld % 4 == 0 \
            and t.data_ptr() % (4 * t.element_size()) == 0:
        return ld
    return None


class _DwConvSiLUCL(torch.autograd.Function):
    """SiLU(depthwise conv3x3(x) + bias) with channel-last tensors on both sides (csrc/ss2d_dwconv_cl.hip): x (B, H, W, D),
    possibly a channel slice of a wider tensor, -> y (B, H, W, D); the backward takes a channel-last dy."""

    @staticmethod
    def forward(ctx, x, weight, bias, holder=None):
        ctx.holder = holder
        _need_gpu(x, "dwconv3x3_silu")
        in_dtype = x.dtype
        io, code = _io(x)
        B, H, W, D = x.shape
        xf = x.to(io)
        ldx = _pixel_pitch(xf, D)
        if ldx is None:
            xf = xf.contiguous()
            ldx = D
        if tuple(weight.shape) != (D, 1, 3, 3):
            raise ValueError(f"dwconv3x3_silu: weight must be ({D}, 1, 3, 3), got {tuple(weight.shape)}")
        wf = weight.float().contiguous()
        bf = None if bias is None else bias.float().contiguous()
        if wf.data_ptr() % 16:           # the kernels fetch a lane's 36 taps / 4 biases as 16-byte pieces
            wf = wf.clone()
        if bf is not None and bf.data_ptr() % 16:
            bf = bf.clone()
        y = torch.empty((B, H, W, D), device=x.device, dtype=io)
        with _lib.on_device(x.device):
            rc = _lib.load().ss2d_dwconv3x3_silu_cl_fwd(code, _lib.ptr(xf), _lib.ptr(wf), _lib.ptr(bf), _lib.ptr(y),
                                                        B, D, H, W, ldx, D, _stream())
        _lib.check(rc, "ss2d_dwconv3x3_silu_cl_fwd")
        ctx.save_for_backward(xf, wf, bf)
        ctx.meta = (B, D, H, W, ldx, in_dtype, weight.dtype, None if bias is None else bias.dtype, code)
        return y.to(in_dtype)

    @staticmethod
    def backward(ctx, dy):
        xf, wf, bf = ctx.saved_tensors
        B, D, H, W, ldx, in_dtype, w_dtype, b_dtype, code = ctx.meta
        g = dy.to(xf.dtype)
        lddy = _pixel_pitch(g, D)
        if lddy is None:
            g = g.contiguous()
            lddy = D
        buf = ctx.holder.buf if ctx.holder is not None else None
        if buf is not None and buf.dtype == xf.dtype and tuple(buf.shape) == (B, H, W, 2 * D) and buf.is_contiguous():
            dx, lddx = buf[..., :D], 2 * D          # first half of the shared d(xz) buffer (GradHalves)
        else:
            dx, lddx = torch.empty((B, H, W, D), device=g.device, dtype=xf.dtype), D
        dw, db = _lib.zeros_group(g.device, tuple(wf.shape), None if bf is None else tuple(bf.shape))
        with _lib.on_device(g.device):
            rc = _lib.load().ss2d_dwconv3x3_silu_cl_bwd(code, _lib.ptr(xf), _lib.ptr(wf), _lib.ptr(bf), _lib.ptr(g),
                                                        _lib.ptr(dx), _lib.ptr(dw), _lib.ptr(db), B, D, H, W, ldx, lddy, lddx,
                                                        _stream())
        _lib.check(rc, "ss2d_dwconv3x3_silu_cl_bwd")
        return dx.to(in_dtype), dw.to(w_dtype), None if db is None else db.to(b_dtype), None


def cross_scan(x: torch.Tensor) -> torch.Tensor:
    """(B, D, H, W) -> (B, 4, D, H*W): row-major, column-major and their reversals (mamba_sys.py:403-404)."""
    return _CrossScan.apply(x)


def cross_merge(ys: torch.Tensor, H: int, W: int, channel_last: bool = True) -> torch.Tensor:
    """(B, 4, D, H*W) -> (B, H, W, D) (channel_last) or (B, D, H*W): mamba_sys.py:445-449."""
    return _CrossMerge.apply(ys, H, W, channel_last)


def dwconv3x3_silu(x: torch.Tensor, weight: torch.Tensor, bias, channel_last_in: bool = True, holder=None,
                   channel_last_out: bool = False) -> torch.Tensor:
    """SiLU(depthwise conv3x3(x) + bias) -> (B, D, H, W), or (B, H, W, D) with `channel_last_out`.  x is (B, H, W, D)
    (channel_last_in; may be a channel slice of a wider tensor) or (B, D, H, W).  mamba_sys.py:299-308, :558-559.
    `holder`: see GradHalves.  Channel-last on both sides with D % 4 == 0 runs the kernels of csrc/ss2d_dwconv_cl.hip
    (nothing is transposed); every other combination the tile kernels of csrc/ss2d_dwconv.hip."""
    if channel_last_out:
        if channel_last_in and x.shape[-1] % 4 == 0 and x.is_cuda:
            return _DwConvSiLUCL.apply(x, weight, bias, holder)
        return _DwConvSiLU.apply(x, weight, bias, channel_last_in, holder).permute(0, 2, 3, 1)
    return _DwConvSiLU.apply(x, weight, bias, channel_last_in, holder)


def linear_wgrad(grad_out: torch.Tensor, inp: torch.Tensor) -> torch.Tensor:
    """dW (M, N) fp32 of a bias-free Linear: sum over all leading axes of grad_out[..., M]^T inp[..., N], on the
    MFMA kernel of csrc/ss2d_wgrad.hip (mamba_sys.py:298/:337 in_proj / out_proj).  bf16 or fp32 operands."""
    _need_gpu(grad_out, "linear_wgrad")
    M, N = grad_out.shape[-1], inp.shape[-1]
    io, code = _io(grad_out)
    g2 = grad_out.reshape(-1, M).to(io)
    x2 = inp.reshape(-1, N).to(io)
    if g2.stride(1) != 1:
        g2 = g2.contiguous()
    if x2.stride(1) != 1:
        x2 = x2.contiguous()
    if g2.shape[0] != x2.shape[0]:
        raise ValueError(f"linear_wgrad: {tuple(grad_out.shape)} and {tuple(inp.shape)} disagree on the token axes")
    dw = _lib.zeros(grad_out.device, (M, N))
    with _lib.on_device(grad_out.device):
        rc = _lib.load().ss2d_linear_wgrad(code, _lib.ptr(g2), _lib.ptr(x2), _lib.ptr(dw), g2.shape[0], M, N,
                                           g2.stride(0), x2.stride(0), N, _stream())
    _lib.check(rc, "ss2d_linear_wgrad")
    return dw


def chan_wgrad(P: torch.Tensor, Q: torch.Tensor) -> torch.Tensor:
    """out (G, M, N) fp32 = sum_b P[b, g] @ Q[b, g]^T for channel-major operands P (B, G, M, L), Q (B, G, N, L) whose rows
    are contiguous and L apart (views into wider blocks are fine): the weight gradient of dt_proj (mamba_sys.py:420; also
    right for x_proj's, :407, where the library's batched GEMM is as fast) on the MFMA kernel of csrc/ss2d_cmwgrad.hip.
    None when the layout does not qualify."""
    _need_gpu(P, "chan_wgrad")
    B, G, M, L = P.shape
    N = Q.shape[2]
    if (Q.shape[0], Q.shape[1], Q.shape[3]) != (B, G, L) or P.dtype != Q.dtype or P.dtype not in (torch.float32, torch.bfloat16):
        return None
    ok = lambda t: t.stride(3) == 1 and t.stride(2) == L and t.stride(0) % 4 == 0 and t.stride(1) % 4 == 0  # noqa: E731
    if L % 4 or not ok(P) or not ok(Q) or (P.data_ptr() | Q.data_ptr()) % (4 * P.element_size()):
        return None
    code = _BF16 if P.dtype == torch.bfloat16 else _F32
    out = _lib.zeros(P.device, (G, M, N))
    wide, narrow, soi, soj, Mb, Ns = (P, Q, N, 1, M, N) if M >= N else (Q, P, 1, N, N, M)
    with _lib.on_device(P.device):
        rc = _lib.load().ss2d_chan_wgrad(code, _lib.ptr(wide), _lib.ptr(narrow), _lib.ptr(out), B, G, Mb, Ns, L, wide.stride(0),
                                         wide.stride(1), narrow.stride(0), narrow.stride(1), M * N, soi, soj, _stream())
    _lib.check(rc, "ss2d_chan_wgrad")
    return out


def chan_gemm(W: torch.Tensor, X: torch.Tensor, padded: bool = False, out=None):
    """Y (B, G, M, L) = W (G, M, K) @ X (B, G, K, L) for a channel-major X (bf16 or fp32) whose rows are L apart (row ranges
    of wider blocks are fine): x_proj, dt_proj and their input gradients (mamba_sys.py:407, :420) on the MFMA kernels of
    csrc/ss2d_cmgemm.hip (fp32: three-way bf16 split, fp32 results).  Y has X's dtype.  None when dtype or layout do not
    qualify (the caller then uses the BLAS library).  `padded`: W is a (G, M, K) view of a buffer in X's dtype whose rows
    are already padded with zeros to a multiple of 16 bytes (block_params): used in place, no cast / pad launches.
    `out`: a (B, G, M, L) view (rows contiguous and L apart, e.g. row ranges of wider blocks) that receives Y."""
    _need_gpu(X, "chan_gemm")
    B, G, K, L = X.shape
    M = W.shape[1]
    if X.dtype not in (torch.bfloat16, torch.float32) or W.shape[0] != G or W.shape[2] != K or L % 4:
        return None
    f32 = X.dtype == torch.float32
    if not (X.stride(3) == 1 and X.stride(2) == L and X.stride(0) % 4 == 0 and X.stride(1) % 4 == 0) or X.data_ptr() % (16 if f32 else 8):
        return None
    q = 4 if f32 else 8                                               # W rows are padded with zeros to 16 bytes
    if padded:
        if W.dtype != X.dtype or W.stride(2) != 1 or W.stride(1) % q or W.stride(1) < K or W.stride(0) % q or W.data_ptr() % 16:
            raise ValueError("chan_gemm: `padded` weight is not a view of a 16-byte padded buffer in the activations' dtype")
        Wp, ldw, swg = W, W.stride(1), W.stride(0)
    else:
        ldw = (K + q - 1) // q * q
        Wp = W.to(X.dtype)
        Wp = torch.nn.functional.pad(Wp, (0, ldw - K)) if ldw != K else Wp.contiguous()
        swg = M * ldw
    if out is not None:
        if (tuple(out.shape) != (B, G, M, L) or out.dtype != X.dtype or out.stride(3) != 1 or out.stride(2) != L or out.stride(0) % 4
                or out.stride(1) % 4 or out.data_ptr() % (16 if f32 else 8)):
            return None
        Y = out
    else:
        Y = torch.empty((B, G, M, L), device=X.device, dtype=X.dtype)
    with _lib.on_device(X.device):
        rc = _lib.load().ss2d_chan_gemm(_F32 if f32 else _BF16, _lib.ptr(Wp), _lib.ptr(X), _lib.ptr(Y), B, G, M, K, ldw, L, swg,
                                        X.stride(0), X.stride(1), Y.stride(0), Y.stride(1), _stream())
    _lib.check(rc, "ss2d_chan_gemm")
    return Y


class _XProj(torch.autograd.Function):
    """x_dbl[b, o] = W[o] @ xs[b, o] (mamba_sys.py:407 with the two directions of an orientation stacked in W).  bf16:
    forward and dX on the MFMA kernel of csrc/ss2d_cmgemm.hip (26 / 14 / 19 us and 27 / 20 / 12 us at stages 0-2 against
    the library's 56 / 23 / 24 and 49 / 41 / 30), dW on csrc/ss2d_cmwgrad.hip where it is the faster one (long rows);
    fp32 and odd layouts: the library's batched GEMM.
    With `wT` and `holder` (block_params): w / wT are the prepared operands (activations' dtype, padded rows) and are used
    in place; the fp32 weight gradient is left in holder.d_xw for block_params' backward, which regroups it into
    x_proj_weight's layout -- w itself gets no gradient from this node."""

    @staticmethod
    def forward(ctx, w, xs, wT=None, holder=None):
        if torch.is_autocast_enabled():
            xs = xs.to(torch.get_autocast_dtype("cuda"))
        ctx.holder = holder
        prepared = wT is not None and w.dtype == xs.dtype
        ctx.prepared = prepared
        w2 = w if prepared else w.to(xs.dtype)
        ctx.save_for_backward(w2, xs, wT if prepared else None)
        ctx.w_dtype = w.dtype
        y = chan_gemm(w2, xs, padded=prepared)
        if y is None:
            with torch.autocast("cuda", enabled=False):
                y = torch.matmul(w2.unsqueeze(0), xs)
        return y

    @staticmethod
    def backward(ctx, g):
        w2, xs, wT = ctx.saved_tensors
        g = g.to(xs.dtype)
        g = g if g.is_contiguous() else g.contiguous()
        dxs = dw = None
        want_dw = ctx.needs_input_grad[0] or (ctx.holder is not None and ctx.holder.want_xw)
        with torch.autocast("cuda", enabled=False):
            if ctx.needs_input_grad[1]:
                dxs = chan_gemm(wT, g, padded=True) if ctx.prepared else chan_gemm(w2.transpose(1, 2), g)
                if dxs is None:
                    dxs = torch.matmul(w2.transpose(1, 2).unsqueeze(0), g)
            if want_dw:
                dw = chan_wgrad(g, xs) if g.dtype == torch.bfloat16 and g.shape[-1] >= 512 else None
                if dw is None:
                    dw = torch.matmul(g, xs.transpose(2, 3)).sum(0, dtype=torch.float32)
        if ctx.holder is not None:
            ctx.holder.d_xw = dw          # fp32 (2, 2Cx, D): picked up by _BlockParams.backward
            return None, dxs, None, None
        return (None if dw is None else dw.to(ctx.w_dtype)), dxs, None, None


def x_proj(w: torch.Tensor, xs: torch.Tensor, wT=None, holder=None) -> torch.Tensor:
    """w (G, C, D), xs (B, G, D, L) -> (B, G, C, L): the x_proj of mamba_sys.py:407 for channel-major activations."""
    return _XProj.apply(w, xs, wT, holder)


def colsum(g: torch.Tensor) -> torch.Tensor:
    """fp32 sum over all leading axes of g[..., M] (a Linear's bias gradient), csrc/ss2d_colsum.hip: deterministic, and
    unlike torch's multi-block reduction it replays correctly inside a captured HIP graph."""
    _need_gpu(g, "colsum")
    io, code = _io(g)
    M = g.shape[-1]
    g2 = g.reshape(-1, M).to(io)
    if g2.stride(1) != 1:
        g2 = g2.contiguous()
    lib = _lib.load()
    out = torch.empty(M, device=g.device, dtype=torch.float32)
    ws = torch.empty(int(lib.ss2d_colsum_workspace_floats(M)), device=g.device, dtype=torch.float32)
    with _lib.on_device(g.device):
        rc = lib.ss2d_colsum(code, _lib.ptr(g2), _lib.ptr(out), _lib.ptr(ws), g2.shape[0], M, g2.stride(0), _stream())
    _lib.check(rc, "ss2d_colsum")
    return out


_OWN_TOKEN_GEMM = not os.environ.get("SS2D_LIBRARY_LINEAR")      # A/B switch: the token-major Linears on the BLAS library


def token_gemm(a: torch.Tensor, w: torch.Tensor, k_major: bool):
    """a (..., K) times a weight: k_major False -> a @ w.T with w (N, K) (F.linear); True -> a @ w with w (K, N) (the
    input gradient of F.linear) -- csrc/ss2d_tokgemm.hip.  bf16, or fp32 (three-way bf16 split on the same MFMA: fp32
    results).  Returns (..., N) in the operands' dtype, or None when dtype / layout do not qualify (the caller then uses the
    BLAS library)."""
    if a.dtype not in (torch.bfloat16, torch.float32) or w.dtype != a.dtype or not a.is_cuda or w.dim() != 2:
        return None
    q = 16 // a.element_size()                                        # elements per 16 bytes
    K = a.shape[-1]
    N = w.shape[1] if k_major else w.shape[0]
    if (w.shape[0] if k_major else w.shape[1]) != K or K % q or N % q or w.stride(1) != 1 or w.stride(0) % q or w.data_ptr() % 16:
        return None                                                   # a column range of a wider weight is fine (_ConcatLinear)
    a2 = a.reshape(-1, K)
    if a2.shape[0] == 0 or a2.stride(1) != 1 or a2.data_ptr() % 16:
        return None
    lda = a2.stride(0) if a2.shape[0] > 1 else K
    if lda % q or lda < K:
        return None
    out = torch.empty((a2.shape[0], N), device=a.device, dtype=a.dtype)
    with _lib.on_device(a.device):
        rc = _lib.load().ss2d_token_gemm(_io(a)[1], _lib.ptr(a2), _lib.ptr(w), _lib.ptr(out), a2.shape[0], N, K, lda,
                                         w.stride(0), N, int(k_major), _stream())
    _lib.check(rc, "ss2d_token_gemm")
    return out.view(*a.shape[:-1], N)


def _own_wgrad_pays(rows: int, m: int, n: int, dtype) -> bool:
    """Split-token MFMA weight gradient or the library's g^T x: the own kernel wins wherever the token axis is long against
    the output (stages 0 / 1) and, with 16-byte aligned rows (the LDS-staged kernels, bf16 and fp32), down to ~2000 tokens
    (stage 2: bf16 21-28 us against 37, fp32 37-58 against 43-66; at the 1176 tokens of stage 3 they tie or the library
    wins) -- tools/wgrad_shapes_bench.py."""
    q = 8 if dtype == torch.bfloat16 else 4
    return rows >= 32 * min(m, n) or (rows >= 2048 and m % q == 0 and n % q == 0)


def _lib_wgrad(g: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    g2 = g.reshape(-1, g.shape[-1])
    return g2.t() @ x.reshape(g2.shape[0], -1)


def _token_gemm_pays(tokens: int, n: int, k: int, dtype=torch.bfloat16) -> bool:
    """The own kernel walks the whole contraction in one workgroup per 128 x 96 tile: with a long contraction and too few
    tiles to fill 256 CUs (stage 3: 1176 tokens) the library's split is faster (tools/token_gemm_bench.py).  fp32 (six MFMAs
    per product): the own kernel wins from one tile per CU on (stages 0-2 except the 148-tile products), the library below."""
    tiles = ((tokens + 127) // 128) * ((n + 95) // 96)
    if dtype == torch.float32:
        return _OWN_TOKEN_GEMM and tiles >= 256
    return _OWN_TOKEN_GEMM and not (k >= 1536 and tiles < 256)


_WORKING_COPIES = {}      # id(parameter) -> (weakref to it, low-precision copy, parameter._version when the copy was made)


def cast_weight(p: torch.Tensor, dt) -> torch.Tensor:
    """p.to(dt) -- or the registered working copy of the parameter (train.WeightCopies: the fp32 master / bf16 working copy
    arrangement of mixed-precision training, refreshed once per optimizer step by ONE multi-tensor kernel instead of a cast
    kernel per Linear and step).  A copy is used only while the parameter's version counter still has the value it had at
    the refresh; any in-place update since (optimizer, load_state_dict, copy_) falls back to the cast."""
    if p.dtype == dt:
        return p
    e = _WORKING_COPIES.get(id(p))
    if e is not None and e[0]() is p and e[1].dtype == dt and e[2] == p._version:
        return e[1]
    return p.to(dt)


def token_gemm_cat(a: torch.Tensor, a2, w: torch.Tensor, bias):
    """[a | a2] @ w.T + bias (fp32 accumulation, fp32 bias added before any rounding) without the concatenation --
    csrc/ss2d_tokgemm.hip.  a (..., K1), a2 (..., K2) or None, w (N, K1 + K2), all bf16 or all fp32; bias (N) fp32 or None.
    None when dtype / layout do not qualify."""
    if a.dtype not in (torch.bfloat16, torch.float32) or w.dtype != a.dtype or not a.is_cuda or w.dim() != 2:
        return None
    f32 = a.dtype == torch.float32
    q = 4 if f32 else 8                                               # elements per 16 bytes
    K1 = a.shape[-1]
    K2 = 0 if a2 is None else a2.shape[-1]
    N, K = w.shape
    if K != K1 + K2 or K % q or N % q or w.stride(1) != 1 or w.stride(0) % q or w.data_ptr() % 16:
        return None
    if a2 is not None and (a2.dtype != a.dtype or a2.shape[:-1] != a.shape[:-1] or K1 % (32 if f32 else 64 if K > 256 else 32)):
        return None
    if bias is not None and (bias.dtype != torch.float32 or not bias.is_contiguous() or bias.data_ptr() % 16 or bias.numel() != N):
        return None
    ops = []
    for t in (a, a2):
        if t is None:
            ops.append((None, 0))
            continue
        t2 = t.reshape(-1, t.shape[-1])
        ld = t2.stride(0) if t2.shape[0] > 1 else t2.shape[1]
        if t2.shape[0] == 0 or t2.stride(1) != 1 or t2.data_ptr() % 16 or ld % q or ld < t2.shape[1]:
            return None
        ops.append((t2, ld))
    (a_, lda), (a2_, lda2) = ops
    out = torch.empty((a_.shape[0], N), device=a.device, dtype=a.dtype)
    with _lib.on_device(a.device):
        rc = _lib.load().ss2d_token_gemm_cat(_io(a)[1], _lib.ptr(a_), _lib.ptr(a2_), _lib.ptr(w), _lib.ptr(bias), _lib.ptr(out), a_.shape[0],
                                             N, K, K1, lda, lda2, w.stride(0), N, _stream())
    _lib.check(rc, "ss2d_token_gemm_cat")
    return out.view(*a.shape[:-1], N)


class _Linear(torch.autograd.Function):
    """y = x W^T (+ b) for the Linears whose token axis dwarfs both feature axes: forward and dX on the token-major MFMA
    kernel (ss2d_token_gemm: bf16, or fp32 as a three-way bf16 split; hipBLASLt with a non-fp32 bias and where
    _token_gemm_pays says no), dW on the split-token MFMA kernel, db a column sum."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        if torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype("cuda")
            x, w = x.to(dt), cast_weight(weight, dt)
            b = None if bias is None else bias.to(dt)
        else:
            w, b = weight, bias
        ctx.save_for_backward(x, w)
        ctx.w_dtype, ctx.b_dtype = weight.dtype, None if bias is None else bias.dtype
        y = None
        # bf16: own MFMA kernel.  Its bias input is fp32 (added before the rounding): a bias of any other dtype (a module
        # converted with .bfloat16(), no autocast) goes to the library below -- never silently without its bias
        if _token_gemm_pays(x.numel() // x.shape[-1], w.shape[0], w.shape[1], x.dtype) and (bias is None or bias.dtype == torch.float32):
            y = token_gemm(x, w, False) if bias is None else token_gemm_cat(x, None, w, bias)
        if y is not None:
            return y
        with torch.autocast("cuda", enabled=False):
            return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        g = g.to(x.dtype)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = token_gemm(g, w, True) if _token_gemm_pays(g.numel() // g.shape[-1], w.shape[1], w.shape[0], g.dtype) else None
            if dx is None:
                dx = torch.matmul(g, w)
        dw = None
        if ctx.needs_input_grad[1]:
            own = _own_wgrad_pays(g.numel() // g.shape[-1], g.shape[-1], x.shape[-1], g.dtype)
            dw = (linear_wgrad(g, x) if own else _lib_wgrad(g, x)).to(ctx.w_dtype)
        db = None
        if ctx.b_dtype is not None and ctx.needs_input_grad[2]:
            db = colsum(g).to(ctx.b_dtype)
        return dx, dw, db


class _LinearRes(torch.autograd.Function):
    """res + x W^T for a bias-free Linear (out_proj and the residual connection of a VSSBlock, mamba_sys.py:562, :585) in
    one pass on the token GEMM (ss2d_token_gemm_res: the product is rounded to the storage type before the sum, bit for bit
    what the Linear followed by `res + y` gives).  Backward: d res = g as it is, dx and dW as _Linear."""

    @staticmethod
    def forward(ctx, x, weight, res):
        if torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype("cuda")
            x, w = x.to(dt), cast_weight(weight, dt)
        else:
            w = weight
        ctx.save_for_backward(x, w)
        ctx.w_dtype = weight.dtype
        N, K = w.shape
        q = 16 // x.element_size()
        x2, r2 = x.reshape(-1, K), res.reshape(-1, N)
        ok = (x.dtype in (torch.bfloat16, torch.float32) and w.dtype == x.dtype and res.dtype == x.dtype and x2.is_contiguous() and r2.is_contiguous()
              and w.is_contiguous() and K % q == 0 and N % q == 0 and (x2.data_ptr() | r2.data_ptr() | w.data_ptr()) % 16 == 0
              and _token_gemm_pays(x2.shape[0], N, K, x.dtype))
        if ok:
            out = torch.empty_like(r2)
            with _lib.on_device(x.device):
                rc = _lib.load().ss2d_token_gemm_res(_io(x)[1], _lib.ptr(x2), _lib.ptr(w), _lib.ptr(r2), _lib.ptr(out), x2.shape[0], N, K,
                                                     K, K, N, N, _stream())
            _lib.check(rc, "ss2d_token_gemm_res")
            return out.view(res.shape)
        with torch.autocast("cuda", enabled=False):
            return res + torch.nn.functional.linear(x, w).to(res.dtype)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = g.to(x.dtype)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = token_gemm(gx, w, True) if _token_gemm_pays(gx.numel() // gx.shape[-1], w.shape[1], w.shape[0], gx.dtype) else None
            if dx is None:
                dx = torch.matmul(gx, w)
        dw = None
        if ctx.needs_input_grad[1]:
            own = _own_wgrad_pays(gx.numel() // gx.shape[-1], gx.shape[-1], x.shape[-1], gx.dtype)
            dw = (linear_wgrad(gx, x) if own else _lib_wgrad(gx, x)).to(ctx.w_dtype)
        return dx, dw, (g if ctx.needs_input_grad[2] else None)


def linear_residual(module: torch.nn.Module, x: torch.Tensor, res: torch.Tensor) -> torch.Tensor:
    """res + module(x) for a bias-free nn.Linear, as one kernel where the token GEMM applies."""
    rows = x.numel() // x.shape[-1]
    if (type(module) is torch.nn.Linear and module.bias is None and x.is_cuda and rows >= 1024 and res.shape[:-1] == x.shape[:-1]
            and res.shape[-1] == module.out_features and not _os.environ.get("SS2D_NO_WGRAD") and not _os.environ.get("SS2D_NO_LINEAR_RES")):
        return _LinearRes.apply(x, module.weight, res)
    return res + apply_linear(module, x)


class _ConcatLinear(torch.autograd.Function):
    """y = cat([x, skip], -1) W^T + b without the concatenated tensor (the decoder's skip connection,
    mamba_sys.py:835): W = [W1 | W2], y = x W1^T + b, then y += skip W2^T accumulated in place by the second GEMM.
    Backward: dx = g W1, dskip = g W2 (no split of a concatenated gradient), dW from the MFMA weight-gradient kernel for
    each half, db from `colsum`."""

    @staticmethod
    def forward(ctx, x, skip, weight, bias):
        if torch.is_autocast_enabled():
            dt = torch.get_autocast_dtype("cuda")
            x, skip, w = x.to(dt), skip.to(dt), cast_weight(weight, dt)
            b = None if bias is None else bias.to(dt)
        else:
            w, b = weight, bias
            skip = skip.to(x.dtype)
        C1 = x.shape[-1]
        ctx.save_for_backward(x, skip, w)
        ctx.w_dtype, ctx.b_dtype = weight.dtype, None if bias is None else bias.dtype
        y = None
        if _token_gemm_pays(x.numel() // C1, w.shape[0], w.shape[1], x.dtype) and (bias is None or bias.dtype == torch.float32):
            y = token_gemm_cat(x, skip, w, bias)                       # both halves and the bias in one pass
        if y is not None:
            return y
        with torch.autocast("cuda", enabled=False):
            y = torch.nn.functional.linear(x, w[:, :C1], b)
            y.view(-1, y.shape[-1]).addmm_(skip.reshape(-1, skip.shape[-1]), w[:, C1:].t())
        return y

    @staticmethod
    def backward(ctx, g):
        x, skip, w = ctx.saved_tensors
        g = g.to(x.dtype)
        C1 = x.shape[-1]
        rows, M = g.numel() // g.shape[-1], g.shape[-1]

        def dinput(wpart):                 # g @ W[:, part]: the k-major token GEMM reads the column range in place
            d = token_gemm(g, wpart, True) if _token_gemm_pays(rows, wpart.shape[1], M, g.dtype) else None
            return torch.matmul(g, wpart) if d is None else d
        dx = dinput(w[:, :C1]) if ctx.needs_input_grad[0] else None
        dskip = dinput(w[:, C1:]) if ctx.needs_input_grad[1] else None
        dw = None
        if ctx.needs_input_grad[2]:
            wg = linear_wgrad if _own_wgrad_pays(rows, M, C1, g.dtype) else _lib_wgrad
            dw = torch.cat([wg(g, x), wg(g, skip)], dim=1).to(ctx.w_dtype)
        db = colsum(g).to(ctx.b_dtype) if (ctx.b_dtype is not None and ctx.needs_input_grad[3]) else None
        return dx, dskip, dw, db


def concat_linear(module: torch.nn.Module, x: torch.Tensor, skip: torch.Tensor) -> torch.Tensor:
    """module(torch.cat([x, skip], -1)) for an nn.Linear, without materialising the concatenation."""
    if type(module) is torch.nn.Linear and x.is_cuda and module.in_features == x.shape[-1] + skip.shape[-1] and x.shape[:-1] == skip.shape[:-1]:
        return _ConcatLinear.apply(x, skip, module.weight, module.bias)
    return apply_linear(module, torch.cat([x, skip], dim=-1))


def linear(x: torch.Tensor, weight: torch.Tensor, bias=None) -> torch.Tensor:
    """F.linear(x, weight, bias) through _Linear on the GPU: every product goes to the own MFMA kernel or to the BLAS library
    by shape (_token_gemm_pays, _own_wgrad_pays); plain F.linear for a handful of rows, on the CPU, and with the A/B switch."""
    rows = x.numel() // x.shape[-1]
    if x.is_cuda and rows >= 1024 and not _os.environ.get("SS2D_NO_WGRAD"):   # env: A/B switch
        return _Linear.apply(x, weight, bias)
    return torch.nn.functional.linear(x, weight, bias)


def linear_nobias(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    return linear(x, weight, None)


def apply_linear(module: torch.nn.Module, x: torch.Tensor) -> torch.Tensor:
    """module(x) for an nn.Linear (or nn.Identity) through `linear` above; the module and its parameters are untouched."""
    if type(module) is torch.nn.Linear:
        return linear(x, module.weight, module.bias)
    return module(x)


class _OutNormGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, z, gamma, beta, eps, holder=None, scale=None):
        _need_gpu(y, "outnorm_gate")
        ctx.holder = holder
        D = y.shape[-1]
        io, code = _io(y)
        y2 = y.reshape(-1, D)
        y2 = y2 if y2.is_contiguous() else y2.contiguous()
        z2 = z.to(io)
        rows = y2.shape[0]
        # z may be the second half of xz: unit channel stride, one constant token stride
        zs = z2.reshape(rows, D) if z2.is_contiguous() else z2
        if not z2.is_contiguous():
            tok = z2.stride(-2)
            ok = z2.stride(-1) == 1 and tok % 4 == 0 and all(z2.stride(i) == z2.stride(i + 1) * z2.shape[i + 1] for i in range(z2.dim() - 2))
            if ok and z2.data_ptr() % 16 == 0:
                ldz = tok
            else:
                zs, ldz = z2.contiguous().reshape(rows, D), D
        if z2.is_contiguous():
            ldz = D
        g32, b32 = gamma.float().contiguous(), beta.float().contiguous()
        out = torch.empty((rows, D), device=y.device, dtype=io)
        mean = torch.empty(rows, device=y.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        rps = 0
        if scale is not None:      # one factor per sample (leading axis of y): fp32, contiguous
            if scale.numel() != y.shape[0] or rows % y.shape[0]:
                raise ValueError("outnorm_gate: `scale` must hold one factor per sample")
            scale = scale.detach().reshape(-1).float().contiguous()
            rps = rows // y.shape[0]
        with _lib.on_device(y.device):
            rc = _lib.load().ss2d_outnorm_gate_scaled_fwd(code, _lib.ptr(y2), _lib.ptr(zs), _lib.ptr(g32), _lib.ptr(b32), _lib.ptr(scale), rps,
                                                          _lib.ptr(out), _lib.ptr(mean), _lib.ptr(rstd), rows, D, ldz, float(eps), _stream())
        _lib.check(rc, "ss2d_outnorm_gate_fwd")
        ctx.save_for_backward(y2, zs, g32, b32, mean, rstd, scale)
        ctx.meta = (code, rows, D, ldz, y.shape, z.shape, y.dtype, z.dtype, gamma.dtype, beta.dtype, rps)
        return out.view(y.shape)

    @staticmethod
    def backward(ctx, dout):
        y2, zs, g32, b32, mean, rstd, scale = ctx.saved_tensors
        code, rows, D, ldz, yshape, zshape, ydt, zdt, gdt, bdt, rps = ctx.meta
        g = dout.reshape(rows, D).to(y2.dtype)
        g = g if g.is_contiguous() else g.contiguous()
        dy = torch.empty_like(y2)
        if ctx.holder is not None and zdt == y2.dtype and D % 4 == 0:
            # second half of the shared d(xz) buffer; the convolution's backward fills the first (GradHalves)
            ctx.holder.buf = torch.empty(tuple(zshape[:-1]) + (2 * D,), device=g.device, dtype=y2.dtype)
            dz, lddz = ctx.holder.buf[..., D:], 2 * D
        else:
            dz, lddz = torch.empty((rows, D), device=g.device, dtype=y2.dtype), D
        dg, db = _lib.zeros_group(g.device, tuple(g32.shape), tuple(b32.shape))
        with _lib.on_device(g.device):
            rc = _lib.load().ss2d_outnorm_gate_scaled_bwd(code, _lib.ptr(g), _lib.ptr(y2), _lib.ptr(zs), _lib.ptr(g32), _lib.ptr(b32),
                                                          _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(scale), rps, _lib.ptr(dy),
                                                          _lib.ptr(dz), _lib.ptr(dg), _lib.ptr(db), rows, D, ldz, lddz, _stream())
        _lib.check(rc, "ss2d_outnorm_gate_bwd")
        return dy.view(yshape).to(ydt), dz.view(zshape).to(zdt), dg.to(gdt), db.to(bdt), None, None, None


def outnorm_gate(y: torch.Tensor, z: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5, holder=None, scale=None) -> torch.Tensor:
    """LayerNorm over the last axis of y (weight, bias, eps) times SiLU(z), fused (mamba_sys.py:452 + :561).
    z may be a channel slice of a wider tensor (the z half of xz).  Output has y's dtype.  `holder`: see GradHalves.
    `scale` (one factor per sample, not differentiated): multiplied into the output -- the block's stochastic-depth factor,
    applied in front of the bias-free out_proj (mamba_sys.py:581-585)."""
    return _OutNormGate.apply(y, z, weight, bias, eps, holder, scale)


class _LayerNorm(torch.autograd.Function):
    """LayerNorm over the last axis; with ps > 0 over the rows of a ps x ps pixel un-shuffle of x (B, H, W, ps*ps*D):
    out (B, H*ps, W*ps, D) = norm(rearrange(x, 'b h w (p1 p2 c) -> b (h p1) (w p2) c')), gathered inside the kernel;
    with ps = -2 over the 2x2 neighbourhoods of x (B, H, W, C): out (B, H/2, W/2, 4C) = norm(x0 | x1 | x2 | x3)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, ps=0, skip=False):
        _need_gpu(x, "layer_norm")
        io, code = _io(x)
        if ps < 0:
            B, H, W, C = x.shape
            D = 4 * C
            x2 = x.to(io)
            x2 = x2 if x2.is_contiguous() else x2.contiguous()
            out_shape, rows = (B, H // 2, W // 2, D), B * (H // 2) * (W // 2)
        elif ps:
            B, H, W, C = x.shape
            D = C // (ps * ps)
            x2 = x.to(io)
            x2 = x2 if x2.is_contiguous() else x2.contiguous()
            out_shape, rows = (B, H * ps, W * ps, D), B * H * ps * W * ps
        else:
            D = x.shape[-1]
            x2 = x.to(io).reshape(-1, D)
            x2 = x2 if x2.is_contiguous() else x2.contiguous()
            out_shape, rows = x.shape, x2.shape[0]
        # under bf16 autocast the consumer is a bf16 GEMM: write bf16 directly (torch would return fp32 and cast)
        out_bf16 = io == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)
        ocode, odt = (_BF16, torch.bfloat16) if out_bf16 else (_F32, torch.float32)
        g32, b32 = weight.float().contiguous(), bias.float().contiguous()
        out = torch.empty((rows, D), device=x.device, dtype=odt)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        lib = _lib.load()
        with _lib.on_device(x.device):
            if ps < 0:
                rc = lib.ss2d_layernorm_merge_fwd(code, ocode, _lib.ptr(x2), _lib.ptr(g32), _lib.ptr(b32), _lib.ptr(out),
                                                  _lib.ptr(mean), _lib.ptr(rstd), B, H, W, C, float(eps), _stream())
            elif ps:
                rc = lib.ss2d_layernorm_unshuffle_fwd(code, ocode, _lib.ptr(x2), _lib.ptr(g32), _lib.ptr(b32), _lib.ptr(out),
                                                      _lib.ptr(mean), _lib.ptr(rstd), B, H, W, ps, D, float(eps), _stream())
            else:
                rc = lib.ss2d_layernorm_fwd(code, ocode, _lib.ptr(x2), _lib.ptr(g32), _lib.ptr(b32), _lib.ptr(out),
                                            _lib.ptr(mean), _lib.ptr(rstd), rows, D, float(eps), _stream())
        _lib.check(rc, "ss2d_layernorm_fwd")
        ctx.save_for_backward(x2, g32, mean, rstd)
        ctx.meta = (code, ocode, odt, rows, D, x.shape, x.dtype, weight.dtype, bias.dtype, ps)
        if skip:
            # (out, x): x leaves as a second output, so that the gradient of everything else that consumes x -- the
            # residual sum around the norm -- arrives HERE and is added inside the backward kernel (ss2d_layernorm_bwd_add)
            # instead of by autograd's accumulation pass
            ctx.set_materialize_grads(False)
            return out.view(out_shape), x.view_as(x)
        return out.view(out_shape)

    @staticmethod
    def backward(ctx, dout, dskip=None):
        x2, g32, mean, rstd = ctx.saved_tensors
        code, ocode, odt, rows, D, xshape, xdt, wdt, bdt, ps = ctx.meta
        if dout is None:       # only the pass-through output was used
            return (None if dskip is None else dskip.to(xdt)), None, None, None, None, None
        g = dout.reshape(rows, D).to(odt)
        g = g if g.is_contiguous() else g.contiguous()
        dx = torch.empty_like(x2)
        add = None
        if dskip is not None and ps == 0:
            add = dskip.reshape(rows, D).to(x2.dtype)
            add = add if add.is_contiguous() else add.contiguous()
        dg, db = _lib.zeros_group(g.device, tuple(g32.shape), tuple(g32.shape))
        lib = _lib.load()
        with _lib.on_device(g.device):
            if ps < 0:
                B, H, W, C = xshape
                rc = lib.ss2d_layernorm_merge_bwd(code, ocode, _lib.ptr(g), _lib.ptr(x2), _lib.ptr(g32), _lib.ptr(mean),
                                                  _lib.ptr(rstd), _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db), B, H, W, C, _stream())
            elif ps:
                B, H, W, _ = xshape
                rc = lib.ss2d_layernorm_unshuffle_bwd(code, ocode, _lib.ptr(g), _lib.ptr(x2), _lib.ptr(g32), _lib.ptr(mean),
                                                      _lib.ptr(rstd), _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db), B, H, W, ps, D, _stream())
            elif add is not None:
                rc = lib.ss2d_layernorm_bwd_add(code, ocode, _lib.ptr(g), _lib.ptr(x2), _lib.ptr(g32), _lib.ptr(mean), _lib.ptr(rstd),
                                                _lib.ptr(add), _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db), rows, D, _stream())
            else:
                rc = lib.ss2d_layernorm_bwd(code, ocode, _lib.ptr(g), _lib.ptr(x2), _lib.ptr(g32), _lib.ptr(mean),
                                            _lib.ptr(rstd), _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db), rows, D, _stream())
        _lib.check(rc, "ss2d_layernorm_bwd")
        dx = dx.view(xshape).to(xdt)
        if dskip is not None and add is None:      # gathered rows (ps != 0): not fused
            dx = dx + dskip.to(xdt)
        return dx, dg.to(wdt), db.to(bdt), None, None, None


def layer_norm(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """F.layer_norm(x, (D,), weight, bias, eps) over the last axis on the HIP kernel (D % 4 == 0, D <= 1536, GPU
    tensors); anything else goes to torch.  Under bf16 autocast the result is bf16."""
    D = x.shape[-1]
    if x.is_cuda and D % 4 == 0 and D <= 1536 and weight is not None and bias is not None:
        return _LayerNorm.apply(x, weight, bias, eps)
    return torch.nn.functional.layer_norm(x, (D,), weight, bias, eps)


class LayerNorm(torch.nn.LayerNorm):
    """nn.LayerNorm (same parameters / state-dict keys) whose forward runs the HIP kernel."""

    def forward(self, x):
        if len(self.normalized_shape) == 1 and self.elementwise_affine and self.bias is not None:
            return layer_norm(x, self.weight, self.bias, self.eps)
        return super().forward(x)


def apply_norm(module: torch.nn.Module, x: torch.Tensor) -> torch.Tensor:
    """module(x), on the HIP LayerNorm kernel when `module` is a plain nn.LayerNorm over the last axis (the modules
    stay nn.LayerNorm so the reference's constructor arguments, init and state-dict keys are untouched)."""
    if type(module) is torch.nn.LayerNorm and len(module.normalized_shape) == 1 and module.elementwise_affine \
            and module.bias is not None and x.is_cuda:
        return layer_norm(x, module.weight, module.bias, module.eps)
    return module(x)


_NO_LN_SKIP = bool(__import__("os").environ.get("SS2D_NO_LN_SKIP"))     # A/B switch


def apply_norm_skip(module: torch.nn.Module, x: torch.Tensor):
    """(module(x), x') with x' == x: for `x + f(norm(x))` (VSSBlock, mamba_sys.py:583-585).  Using x' in the residual sum
    routes its gradient through the norm's backward kernel, which adds it to dx in the same pass."""
    if type(module) is torch.nn.LayerNorm and len(module.normalized_shape) == 1 and module.elementwise_affine \
            and module.bias is not None and x.is_cuda and x.shape[-1] % 4 == 0 and x.shape[-1] <= 1536 and torch.is_grad_enabled() and x.requires_grad \
            and not _NO_LN_SKIP:
        return _LayerNorm.apply(x, module.weight, module.bias, module.eps, 0, True)
    return apply_norm(module, x), x


def pixel_unshuffle_cl(x: torch.Tensor, p: int) -> torch.Tensor:
    """(B, H, W, p*p*c) -> (B, H*p, W*p, c), channel groups ordered (p1, p2, c) -- the einops pattern
    'b h w (p1 p2 c) -> b (h p1) (w p2) c' of mamba_sys.py:244, :262 (a copy)."""
    B, H, W, C = x.shape
    c = C // (p * p)
    return x.view(B, H, W, p, p, c).permute(0, 1, 3, 2, 4, 5).reshape(B, H * p, W * p, c)


def apply_norm_unshuffle(module: torch.nn.Module, x: torch.Tensor, p: int) -> torch.Tensor:
    """module(pixel_unshuffle_cl(x, p)) for an nn.LayerNorm over c = C / p^2 channels, with the rearrangement done by
    the LayerNorm kernel's addressing instead of a copy (PatchExpand / FinalPatchExpand_X4, mamba_sys.py:233-265)."""
    c = x.shape[-1] // (p * p)
    if (isinstance(module, torch.nn.LayerNorm) and x.is_cuda and x.dim() == 4 and x.shape[-1] == p * p * c and c % 4 == 0
            and c <= 1536 and module.weight is not None and module.bias is not None and tuple(module.normalized_shape) == (c,)):
        return _LayerNorm.apply(x, module.weight, module.bias, module.eps, p)
    return apply_norm(module, pixel_unshuffle_cl(x, p))


def apply_norm_merge(module: torch.nn.Module, x: torch.Tensor) -> torch.Tensor:
    """module(cat([x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]], -1)) for an nn.LayerNorm over 4C
    channels (PatchMerging2D, mamba_sys.py:213-221), the 2x2 gather done by the LayerNorm kernel's addressing.  Odd sizes
    (the reference truncates them) and other norms take the literal path."""
    B, H, W, C = x.shape
    if (type(module) is torch.nn.LayerNorm and x.is_cuda and H % 2 == 0 and W % 2 == 0 and C % 4 == 0 and 4 * C <= 1536
            and module.weight is not None and module.bias is not None and tuple(module.normalized_shape) == (4 * C,)):
        return _LayerNorm.apply(x, module.weight, module.bias, module.eps, -2)
    h2, w2 = H // 2, W // 2
    parts = [x[:, i::2, j::2, :][:, :h2, :w2, :] for (i, j) in ((0, 0), (1, 0), (0, 1), (1, 1))]
    return apply_norm(module, torch.cat(parts, dim=-1).view(B, h2, w2, 4 * C))


class GradSlots:
    """Shared by the scatter node and the scan node of one SS2D call: the scan's backward leaves its du (B, 4, D, L) here
    instead of gathering it into a dx of its own, and the scatter's backward adds it to the x_proj input gradient
    (B, 2, D, L) in ONE 6-slot gather -- instead of two gathers and autograd's sum of their results."""
    __slots__ = ("du",)

    def __init__(self):
        self.du = None


class _DirsScatter2(torch.autograd.Function):
    """x (B, D, H, W) -> (B, 2, D, pitch): the row-major and the column-major copy only (layout MIRROR_FREE_2); rows of
    H*W positions, `pitch` apart, the pad zero."""

    @staticmethod
    def forward(ctx, x, holder=None, pitch=None, channel_last=False):
        _need_gpu(x, "dirs_scatter2")
        ctx.holder = holder
        if channel_last:
            B, H, W, D = x.shape
        else:
            B, D, H, W = x.shape
        pitch = H * W if pitch is None else int(pitch)
        io, code = _io(x)
        xf = x.to(io).contiguous()
        xs = (torch.empty if pitch == H * W else torch.zeros)((B, 2, D, pitch), device=x.device, dtype=io)
        with _lib.on_device(x.device):
            rc = _lib.load().ss2d_dirs_scatter_ld(code, 2, _lib.ptr(xf), _lib.ptr(xs), B, D, H, W, int(channel_last), pitch, _stream())
        _lib.check(rc, "ss2d_dirs_scatter_ld")
        ctx.shape, ctx.in_dtype, ctx.pitch, ctx.cl = (B, D, H, W), x.dtype, pitch, bool(channel_last)
        return xs.to(x.dtype)

    @staticmethod
    def backward(ctx, dxs):
        B, D, H, W = ctx.shape
        pitch = ctx.pitch
        io, code = _io(dxs)
        cl = ctx.cl
        g = dxs.to(io).contiguous()
        dx = torch.empty((B, H, W, D) if cl else (B, D, H, W), device=g.device, dtype=io)
        du = None
        if ctx.holder is not None:
            du, ctx.holder.du = ctx.holder.du, None
        with _lib.on_device(g.device):
            if du is not None and du.dtype == io and tuple(du.shape) == (B, 4 * D, pitch) and du.is_contiguous():
                gather6 = _lib.load().ss2d_dirs_gather6_cl_ld if cl else _lib.load().ss2d_dirs_gather6_ld
                rc = gather6(code, _lib.ptr(du), _lib.ptr(g), _lib.ptr(dx), B, D, H, W, pitch, _stream())
                du = None
            else:
                rc = _lib.load().ss2d_dirs_gather_ld(code, 2, _lib.ptr(g), _lib.ptr(dx), B, D, H, W, int(cl), pitch, _stream())
        _lib.check(rc, "ss2d_dirs_gather_ld")
        if du is not None:      # a du this node could not fold in (dtype / shape): gather it the plain way and add
            dx2 = torch.empty_like(dx)
            du = du.to(io).contiguous()
            with _lib.on_device(g.device):
                rc = _lib.load().ss2d_dirs_gather_ld(code, 1, _lib.ptr(du), _lib.ptr(dx2), B, D, H, W, int(cl), du.shape[-1], _stream())
            _lib.check(rc, "ss2d_dirs_gather_ld")
            dx = dx + dx2
        return dx.to(ctx.in_dtype), None, None, None


def dirs_scatter2(x: torch.Tensor, holder=None, pitch=None, channel_last: bool = False) -> torch.Tensor:
    """(B, D, H, W) -- or (B, H, W, D) with `channel_last` -- -> (B, 2, D, H*W): x flattened row-major and column-major -- the
    two orientations of CrossScan (mamba_sys.py:403); the two reversed directions (:404) are read from the same copies by
    the scan itself.  With `pitch` > H*W the rows are that far apart and zero behind their H*W positions (see
    SS2D._scan_core_dirs)."""
    return _DirsScatter2.apply(x, holder, pitch, channel_last)


class _DtProj(torch.autograd.Function):
    """dts = einsum("bgrl,gdr->bgdl") (mamba_sys.py:420): forward on the streaming HIP kernel, gradients on library
    GEMMs (dX = W^T g, dW = sum_b g x^T)."""

    @staticmethod
    def forward(ctx, x, weight):
        B, G, R, L = x.shape
        D = weight.shape[1]
        io, code = _io(x)
        xs = x.to(io)
        if not (xs.stride(3) == 1 and xs.stride(2) == L and xs.stride(0) == G * xs.stride(1)):
            xs = xs.contiguous()
        w32 = weight.float().contiguous()
        out = torch.empty((B, G, D, L), device=x.device, dtype=io)
        with _lib.on_device(x.device):
            rc = _lib.load().ss2d_dt_proj_fwd(code, _lib.ptr(xs), _lib.ptr(w32), _lib.ptr(out), B, G, D, R, L,
                                              xs.stride(1), _stream())
        _lib.check(rc, "ss2d_dt_proj_fwd")
        ctx.save_for_backward(xs, weight)
        ctx.x_dtype = x.dtype
        return out.to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        xs, weight = ctx.saved_tensors
        g = g.to(xs.dtype)
        w = weight.to(xs.dtype)
        dx = torch.matmul(w.transpose(1, 2).unsqueeze(0), g) if ctx.needs_input_grad[0] else None
        dw = torch.matmul(g, xs.transpose(2, 3)).sum(0).to(weight.dtype) if ctx.needs_input_grad[1] else None
        return None if dx is None else dx.to(ctx.x_dtype), dw


def dt_proj(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """x (B, G, R, L) (may be the first R rows of each block of x_dbl), weight (G, D, R) -> (B, G, D, L)."""
    B, G, R, L = x.shape
    if x.is_cuda and R in (6, 12) and L % 4 == 0 and x.stride(3) == 1 and x.stride(1) % 4 == 0 and x.data_ptr() % 16 == 0:
        return _DtProj.apply(x, weight)
    return torch.matmul(weight.unsqueeze(0), x)


class _ScanParams(torch.autograd.Function):
    """The parameter transforms in front of the scan (mamba_sys.py:428-431 and the dt_projs_weight of :420) in one
    launch each way: A = -exp(A_logs), Ds, dt_projs_bias, dt_projs_weight, optionally regrouped from the reference's
    direction order to the scan-group order of `scan_merge_fn` (csrc/ss2d_scanparams.hip)."""

    @staticmethod
    def forward(ctx, A_logs, Ds, dt_bias, dt_w, regroup):
        _need_gpu(A_logs, "scan_params")
        K, Dm, R = dt_w.shape
        N = A_logs.shape[1]
        if A_logs.shape[0] != K * Dm or Ds.numel() != K * Dm or dt_bias.numel() != K * Dm:
            raise ValueError("scan_params: inconsistent parameter shapes")
        src = [t.detach().float().contiguous() for t in (A_logs, Ds, dt_bias, dt_w)]
        dev = A_logs.device
        A = torch.empty((K * Dm, N), device=dev, dtype=torch.float32)
        D = torch.empty((K * Dm,), device=dev, dtype=torch.float32)
        bias = torch.empty((K * Dm,), device=dev, dtype=torch.float32)
        w = torch.empty((K, Dm, R), device=dev, dtype=torch.float32)
        with _lib.on_device(dev):
            rc = _lib.load().ss2d_scan_params_fwd(*[_lib.ptr(t) for t in src], _lib.ptr(A), _lib.ptr(D), _lib.ptr(bias),
                                                  _lib.ptr(w), K, Dm, N, R, int(regroup), _stream())
        _lib.check(rc, "ss2d_scan_params_fwd")
        ctx.save_for_backward(A)
        ctx.meta = (K, Dm, N, R, int(regroup), A_logs.dtype, Ds.dtype, dt_bias.dtype, dt_w.dtype, tuple(dt_bias.shape))
        return A, D, bias, w

    @staticmethod
    def backward(ctx, dA, dD, dbias, dw):
        (A,) = ctx.saved_tensors
        K, Dm, N, R, regroup, *dts, bias_shape = ctx.meta
        g = [None if t is None else t.float().contiguous() for t in (dA, dD, dbias, dw)]
        dev = A.device
        dA_logs = torch.empty((K * Dm, N), device=dev, dtype=torch.float32)
        dDs = torch.empty((K * Dm,), device=dev, dtype=torch.float32)
        ddt_bias = torch.empty(bias_shape, device=dev, dtype=torch.float32)
        ddt_w = torch.empty((K, Dm, R), device=dev, dtype=torch.float32)
        with _lib.on_device(dev):
            rc = _lib.load().ss2d_scan_params_bwd(_lib.ptr(A), *[_lib.ptr(t) for t in g], _lib.ptr(dA_logs), _lib.ptr(dDs),
                                                  _lib.ptr(ddt_bias), _lib.ptr(ddt_w), K, Dm, N, R, regroup, _stream())
        _lib.check(rc, "ss2d_scan_params_bwd")
        return dA_logs.to(dts[0]), dDs.to(dts[1]), ddt_bias.to(dts[2]), ddt_w.to(dts[3]), None


def scan_params(A_logs: torch.Tensor, Ds: torch.Tensor, dt_projs_bias: torch.Tensor, dt_projs_weight: torch.Tensor,
                regroup: bool = False):
    """(A_logs (K*D, N), Ds (K*D), dt_projs_bias (K, D), dt_projs_weight (K, D, R)) ->
    (A = -exp(A_logs) (K*D, N), D (K*D), delta_bias (K*D), dt weight (K, D, R)), fp32; with `regroup` in the
    scan-group order (row-major, row-major reversed, column-major, column-major reversed) of `scan_merge_fn`."""
    return _ScanParams.apply(A_logs, Ds, dt_projs_bias, dt_projs_weight, regroup)


class BlockGrads:
    """The weight gradient of the stacked x_proj on its way from _XProj.backward to _BlockParams.backward (fp32; a plain
    attribute, like GradSlots: autograd would cast a gradient returned for the bf16 operand to bf16 first)."""
    __slots__ = ("d_xw", "want_xw")

    def __init__(self, want_xw: bool):
        self.d_xw, self.want_xw = None, want_xw


class _BlockParams(torch.autograd.Function):
    """Every parameter-derived operand of one fused-direction SS2D call in one launch each way (csrc/ss2d_scanparams.hip,
    ss2d_block_params_*): scan_params(regroup=True) plus the stacked x_proj weight, its transpose, the dt_proj weight
    transposed and (ranks other than 6 / 12) the dt_proj weight, in the activations' dtype with 16-byte padded rows --
    the operands chan_gemm takes in place.  As torch ops: a regrouping copy, a cast, a transposing copy and a pad per
    operand and block (mamba_sys.py:407, :420, :428-431 are views / einsum operands there)."""

    @staticmethod
    def forward(ctx, A_logs, Ds, dt_bias, dt_w, x_w, io, holder, want_dtw_io):
        _need_gpu(A_logs, "block_params")
        K, Dm, R = dt_w.shape
        N, Cx = A_logs.shape[1], x_w.shape[1]
        if K != 4 or A_logs.shape[0] != K * Dm or Ds.numel() != K * Dm or dt_bias.numel() != K * Dm or tuple(x_w.shape) != (4, Cx, Dm) or Cx != R + 2 * N:
            raise ValueError("block_params: inconsistent parameter shapes")
        src = [t.detach().float().contiguous() for t in (A_logs, Ds, dt_bias, dt_w, x_w)]
        dev = A_logs.device
        q = 4 if io == torch.float32 else 8
        up = lambda v: (v + q - 1) // q * q  # noqa: E731
        ldD, ldC, ldR = up(Dm), up(2 * Cx), up(R)
        A = torch.empty((K * Dm, N), device=dev, dtype=torch.float32)
        D = torch.empty((K * Dm,), device=dev, dtype=torch.float32)
        bias = torch.empty((K * Dm,), device=dev, dtype=torch.float32)
        w = torch.empty((K, Dm, R), device=dev, dtype=torch.float32)
        xw = torch.empty((2, 2 * Cx, ldD), device=dev, dtype=io)
        xwT = torch.empty((2, Dm, ldC), device=dev, dtype=io)
        dtwT = torch.empty((4, R, ldD), device=dev, dtype=io)
        dtw_io = torch.empty((4, Dm, ldR), device=dev, dtype=io) if want_dtw_io else None
        with _lib.on_device(dev):
            rc = _lib.load().ss2d_block_params_fwd(_F32 if io == torch.float32 else _BF16, *[_lib.ptr(t) for t in src], _lib.ptr(A),
                                                   _lib.ptr(D), _lib.ptr(bias), _lib.ptr(w), _lib.ptr(xw), _lib.ptr(xwT), _lib.ptr(dtwT),
                                                   _lib.ptr(dtw_io), Dm, N, R, ldD, ldC, ldR, _stream())
        _lib.check(rc, "ss2d_block_params_fwd")
        ctx.save_for_backward(A)
        ctx.holder = holder
        ctx.set_materialize_grads(False)     # (the non-differentiable outputs would each get a zero-filled gradient tensor)
        ctx.meta = (Dm, N, R, Cx, A_logs.dtype, Ds.dtype, dt_bias.dtype, dt_w.dtype, x_w.dtype, tuple(dt_bias.shape))
        outs = (xw[:, :, :Dm], xwT[:, :, :2 * Cx], dtwT[:, :, :Dm]) + ((dtw_io[:, :, :R],) if want_dtw_io else ())
        ctx.mark_non_differentiable(*outs)
        return (A, D, bias, w) + outs + ((None,) if not want_dtw_io else ())

    @staticmethod
    def backward(ctx, dA, dD, dbias, dw, *unused):
        (A,) = ctx.saved_tensors
        Dm, N, R, Cx, *dts, bias_shape = ctx.meta
        d_xw = ctx.holder.d_xw if ctx.holder is not None else None
        if ctx.holder is not None:
            ctx.holder.d_xw = None
        g = [None if t is None else t.float().contiguous() for t in (dA, dD, dbias, dw, d_xw)]
        dev = A.device
        dA_logs = torch.empty((4 * Dm, N), device=dev, dtype=torch.float32)
        dDs = torch.empty((4 * Dm,), device=dev, dtype=torch.float32)
        ddt_bias = torch.empty(bias_shape, device=dev, dtype=torch.float32)
        ddt_w = torch.empty((4, Dm, R), device=dev, dtype=torch.float32)
        dx_w = torch.empty((4, Cx, Dm), device=dev, dtype=torch.float32)
        with _lib.on_device(dev):
            rc = _lib.load().ss2d_block_params_bwd(_lib.ptr(A), *[_lib.ptr(t) for t in g], _lib.ptr(dA_logs), _lib.ptr(dDs),
                                                   _lib.ptr(ddt_bias), _lib.ptr(ddt_w), _lib.ptr(dx_w), Dm, N, R, _stream())
        _lib.check(rc, "ss2d_block_params_bwd")
        return dA_logs.to(dts[0]), dDs.to(dts[1]), ddt_bias.to(dts[2]), ddt_w.to(dts[3]), dx_w.to(dts[4]), None, None, None


def block_params(A_logs, Ds, dt_projs_bias, dt_projs_weight, x_proj_weight, io, want_dtw_io: bool = False):
    """-> (A, D, delta_bias, dt_w) fp32 in scan-group order (as scan_params(regroup=True)), xw (2, 2Cx, D), xwT (2, D, 2Cx),
    dtwT (4, R, D), dtw_io (4, D, R) or None: views of 16-byte padded buffers in dtype `io` (chan_gemm(..., padded=True)),
    and the BlockGrads holder that x_proj(..., holder=) fills in its backward."""
    holder = BlockGrads(x_proj_weight.requires_grad)
    return _BlockParams.apply(A_logs, Ds, dt_projs_bias, dt_projs_weight, x_proj_weight, io, holder, want_dtw_io) + (holder,)


class _LNHead(torch.autograd.Function):
    """logits = Conv1x1(LN(pixel_unshuffle(x))) in one pass each way (csrc/ss2d_layernorm.hip, ln_head_*): the tail of
    VSSM.up_x4 (mamba_sys.py:859-866).  x (B, H, W, ps*ps*c) -> (B, classes, H*ps, W*ps)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, ps, w_head):
        _need_gpu(x, "ln_head")
        io, code = _io(x)
        B, H, W, C = x.shape
        D, nc = C // (ps * ps), w_head.shape[0]
        x2 = x.to(io)
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        out_bf16 = io == torch.bfloat16 or (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16)
        ocode, odt = (_BF16, torch.bfloat16) if out_bf16 else (_F32, torch.float32)
        g32, b32 = gamma.float().contiguous(), beta.float().contiguous()
        w32 = w_head.float().reshape(nc, D).contiguous()
        rows = B * H * ps * W * ps
        logits = torch.empty((B, nc, H * ps, W * ps), device=x.device, dtype=odt)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        with _lib.on_device(x.device):
            rc = _lib.load().ss2d_ln_head_fwd(code, ocode, _lib.ptr(x2), _lib.ptr(g32), _lib.ptr(b32), _lib.ptr(w32),
                                              _lib.ptr(logits), _lib.ptr(mean), _lib.ptr(rstd), B, H, W, ps, D, nc,
                                              float(eps), _stream())
        _lib.check(rc, "ss2d_ln_head_fwd")
        ctx.save_for_backward(x2, g32, b32, w32, mean, rstd)
        ctx.meta = (code, ocode, odt, ps, x.shape, x.dtype, gamma.dtype, beta.dtype, w_head.dtype, tuple(w_head.shape))
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        x2, g32, b32, w32, mean, rstd = ctx.saved_tensors
        code, ocode, odt, ps, xshape, xdt, gdt, bdt, wdt, wshape = ctx.meta
        B, H, W, C = xshape
        nc, D = w32.shape
        g = dlogits.to(odt)
        g = g if g.is_contiguous() else g.contiguous()
        dx = torch.empty_like(x2)
        dg, db, dw = _lib.zeros_group(g.device, (D,), (D,), (nc, D))
        with _lib.on_device(g.device):
            rc = _lib.load().ss2d_ln_head_bwd(code, ocode, _lib.ptr(g), _lib.ptr(x2), _lib.ptr(g32), _lib.ptr(b32), _lib.ptr(w32),
                                              _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(dx), _lib.ptr(dg), _lib.ptr(db),
                                              _lib.ptr(dw), B, H, W, ps, D, nc, _stream())
        _lib.check(rc, "ss2d_ln_head_bwd")
        return dx.view(xshape).to(xdt), dg.to(gdt), db.to(bdt), None, None, dw.view(wshape).to(wdt)


def ln_unshuffle_head(norm: torch.nn.Module, x: torch.Tensor, p: int, conv: torch.nn.Module) -> torch.Tensor:
    """conv(norm(pixel_unshuffle_cl(x, p)).permute(0, 3, 1, 2)) for an nn.LayerNorm and a bias-free 1x1 nn.Conv2d, fused
    when the shapes allow (c <= 128, classes <= 16); the separate kernels otherwise."""
    c = x.shape[-1] // (p * p)
    if (isinstance(norm, torch.nn.LayerNorm) and isinstance(conv, torch.nn.Conv2d) and x.is_cuda and x.dim() == 4
            and x.shape[-1] == p * p * c and c % 4 == 0 and c <= 128 and tuple(norm.normalized_shape) == (c,)
            and norm.weight is not None and norm.bias is not None and conv.bias is None and conv.kernel_size == (1, 1)
            and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.groups == 1 and conv.in_channels == c
            and conv.out_channels <= 16):
        return _LNHead.apply(x, norm.weight, norm.bias, norm.eps, p, conv.weight)
    return conv(apply_norm_unshuffle(norm, x, p).permute(0, 3, 1, 2))


class _SegLossStats(torch.autograd.Function):
    """stats (3C+1) = [sum p_c*[y==c] | sum [y==c] | sum p_c^2 | sum -log p_y], p = softmax(logits, 1), in one pass
    over the logits (csrc/ss2d_segloss.hip); the backward is one pass from dL/dstats.  Everything DiceLoss
    (utils/losses.py:179-215) and nn.CrossEntropyLoss need of the (B, C, H, W) tensors."""

    @staticmethod
    def forward(ctx, logits, labels, n_classes):
        _need_gpu(logits, "seg_loss_stats")
        _need_gpu(labels, "seg_loss_stats")
        io, code = _io(logits)
        if logits.dim() < 3 or logits.shape[1] != n_classes:
            raise ValueError(f"seg_loss_stats: logits must be (B, {n_classes}, ...), got {tuple(logits.shape)}")
        B, C = logits.shape[0], n_classes
        HW = logits[0, 0].numel()
        if labels.numel() != B * HW:
            raise ValueError("predict & target shape do not match")          # the reference's assertion text (losses.py:205)
        if labels.dtype not in (torch.uint8, torch.int32, torch.int64):
            labels = labels.long()
        lg = logits.to(io)
        lg = lg if lg.is_contiguous() else lg.contiguous()
        lb = labels if labels.is_contiguous() else labels.contiguous()
        lib = _lib.load()
        stats = torch.empty(3 * C + 1, device=lg.device, dtype=torch.float32)
        ws = torch.empty(int(lib.ss2d_seg_loss_workspace_floats(C)), device=lg.device, dtype=torch.float32)
        with _lib.on_device(lg.device):
            rc = lib.ss2d_seg_loss_stats(code, _lib.ptr(lg), _lib.ptr(lb), lb.element_size(), _lib.ptr(stats), _lib.ptr(ws),
                                         B, C, HW, _stream())
        _lib.check(rc, "ss2d_seg_loss_stats")
        ctx.save_for_backward(lg, lb)
        ctx.meta = (code, B, C, HW, logits.dtype)
        return stats

    @staticmethod
    def backward(ctx, g):
        lg, lb = ctx.saved_tensors
        code, B, C, HW, in_dtype = ctx.meta
        g32 = g.float().contiguous()
        dl = torch.empty_like(lg)
        with _lib.on_device(lg.device):
            rc = _lib.load().ss2d_seg_loss_bwd(code, _lib.ptr(lg), _lib.ptr(lb), lb.element_size(), _lib.ptr(g32), _lib.ptr(dl),
                                               B, C, HW, _stream())
        _lib.check(rc, "ss2d_seg_loss_bwd")
        return dl.to(in_dtype), None, None


def seg_loss_stats(logits: torch.Tensor, labels: torch.Tensor, n_classes: int) -> torch.Tensor:
    """logits (B, C, H, W) fp32 / bf16, labels (B, H, W) or (B, 1, H, W) integers -> fp32 (3C+1):
    [0, C) sum_{b,i} p_c [y == c], [C, 2C) sum [y == c], [2C, 3C) sum p_c^2, [3C] sum -log p_y.  Differentiable in logits."""
    return _SegLossStats.apply(logits, labels, n_classes)

"""SS2D / VSSBlock / VSSLayer / VSSLayer_up on the MI355X kernels.

Host-side mirror of the reference's operator interface (code/networks/mamba_sys.py:267-716): same
constructor arguments, same parameter names and shapes (checkpoint contract), same
random-number consumption order at construction (so a given seed yields the reference's init), but
`forward` runs (default, `SS2D.fused_directions = True`):  in_proj (hipBLASLt) -> ss2d_dwconv3x3_silu (reads the
channel-last x half of xz in place) -> ss2d_dirs_scatter (row-major + column-major copy of x only) -> x_proj
as one batched GEMM per orientation -> scan_core_fn: dt_proj, direction-aware selective scan
(ss2d_selective_scan_dir_fwd / _bwd_ws: reversed directions read in place, B / C read in place from the x_proj
output) and ss2d_dirs_gather (merge, channel-last) as one autograd node -> fused out_norm + SiLU(z) gate ->
out_proj.  `fused_directions = False` keeps the literal pipeline of the reference
(ss2d_cross_scan -> two batched GEMMs -> selective_scan_fn -> ss2d_cross_merge).
"""
from __future__ import annotations

import math
from functools import partial
from typing import Callable

import os

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.utils.checkpoint as checkpoint

from .selective_scan_interface import scan_core_fn, scan_merge_fn, selective_scan_fn
from .ss2d_ops import (GradHalves, GradSlots, split_halves, apply_linear, apply_norm, apply_norm_skip, block_params, cross_merge, cross_scan, dirs_scatter2, dt_proj, dwconv3x3_silu,
                       linear_residual, outnorm_gate, scan_params, x_proj)


class DropPath(nn.Module):
    """Stochastic depth per sample (the reference takes it from timm, mamba_sys.py:12, :581)."""

    def __init__(self, drop_prob: float = 0.0):
        super().__init__()
        self.drop_prob = float(drop_prob)

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.dim() - 1)).bernoulli_(keep)
        return x * mask / keep

    def extra_repr(self):
        return f"drop_prob={self.drop_prob}"


class SS2D(nn.Module):
    """mamba_sys.py:267-565.  Parameters: in_proj.weight (2D,C), conv2d.{weight (D,1,3,3), bias (D)},
    x_proj_weight (4,R+2N,D), dt_projs_weight (4,D,R), dt_projs_bias (4,D), A_logs (4D,N), Ds (4D),
    out_norm.{weight,bias} (D), out_proj.weight (C,D)."""

    def __init__(self, d_model, d_state=16, d_conv=3, expand=2, dt_rank="auto", dt_min=0.001, dt_max=0.1,
                 dt_init="random", dt_scale=1.0, dt_init_floor=1e-4, dropout=0.0, conv_bias=True, bias=False,
                 device=None, dtype=None, **kwargs):
        super().__init__()
        fk = {"device": device, "dtype": dtype}
        self.d_model, self.d_state, self.d_conv, self.expand = d_model, d_state, d_conv, expand
        self.d_inner = int(expand * d_model)
        self.dt_rank = math.ceil(d_model / 16) if dt_rank == "auto" else dt_rank
        if d_conv != 3:
            raise NotImplementedError("the HIP depthwise-conv kernel is 3x3 (the only size the reference uses)")
        K, D, R, N = 4, self.d_inner, self.dt_rank, d_state

        # creation order == random-number order of the reference (mamba_sys.py:298-337)
        self.in_proj = nn.Linear(d_model, 2 * D, bias=bias, **fk)
        self.conv2d = nn.Conv2d(D, D, kernel_size=d_conv, padding=(d_conv - 1) // 2, groups=D, bias=conv_bias, **fk)
        self.act = nn.SiLU()
        xw = [nn.Linear(D, R + 2 * N, bias=False, **fk).weight for _ in range(K)]
        self.x_proj_weight = nn.Parameter(torch.stack(list(xw), dim=0))
        dts = [self.dt_init(R, D, dt_scale, dt_init, dt_min, dt_max, dt_init_floor, **fk) for _ in range(K)]
        self.dt_projs_weight = nn.Parameter(torch.stack([t.weight for t in dts], dim=0))
        self.dt_projs_bias = nn.Parameter(torch.stack([t.bias for t in dts], dim=0))
        self.A_logs = self.A_log_init(N, D, copies=K, merge=True)
        self.Ds = self.D_init(D, copies=K, merge=True)
        self.forward_core = self.forward_corev0
        self.out_norm = nn.LayerNorm(D)
        self.out_proj = nn.Linear(D, d_model, bias=bias, **fk)
        self.dropout = nn.Dropout(dropout) if dropout > 0.0 else None

    # ---- init statics, mamba_sys.py:340-394
    @staticmethod
    def dt_init(dt_rank, d_inner, dt_scale=1.0, dt_init="random", dt_min=0.001, dt_max=0.1, dt_init_floor=1e-4,
                **factory_kwargs):
        proj = nn.Linear(dt_rank, d_inner, bias=True, **factory_kwargs)
        std = dt_rank ** -0.5 * dt_scale
        if dt_init == "constant":
            nn.init.constant_(proj.weight, std)
        elif dt_init == "random":
            nn.init.uniform_(proj.weight, -std, std)
        else:
            raise NotImplementedError(dt_init)
        # bias = softplus^-1(dt), dt log-uniform in [dt_min, dt_max]
        dt = torch.exp(torch.rand(d_inner, **factory_kwargs) * (math.log(dt_max) - math.log(dt_min))
                       + math.log(dt_min)).clamp(min=dt_init_floor)
        with torch.no_grad():
            proj.bias.copy_(dt + torch.log(-torch.expm1(-dt)))
        proj.bias._no_reinit = True
        return proj

    @staticmethod
    def A_log_init(d_state, d_inner, copies=1, device=None, merge=True):
        A_log = torch.log(torch.arange(1, d_state + 1, dtype=torch.float32, device=device)).repeat(d_inner, 1)
        if copies > 1:
            A_log = A_log.unsqueeze(0).repeat(copies, 1, 1)
            if merge:
                A_log = A_log.flatten(0, 1)
        A_log = nn.Parameter(A_log.contiguous())
        A_log._no_weight_decay = True
        return A_log

    @staticmethod
    def D_init(d_inner, copies=1, device=None, merge=True):
        Dv = torch.ones(d_inner, device=device)
        if copies > 1:
            Dv = Dv.unsqueeze(0).repeat(copies, 1)
            if merge:
                Dv = Dv.flatten(0, 1)
        Dv = nn.Parameter(Dv.contiguous())
        Dv._no_weight_decay = True
        return Dv

    bf16_scan_io = False   # class-wide switch (set SS2D.bf16_scan_io = True, e.g. bench_model.py --dtype bf16)

    # ---- mamba_sys.py:396-461
    def forward_corev0(self, x: torch.Tensor):
        """x (B, D, H, W) -> (B, H, W, D), normalised (the reference's swappable `forward_core`)."""
        return self.out_norm(self._scan_core(x)).to(x.dtype)

    fused_directions = True   # class-wide switch: _scan_core_dirs instead of the literal cross_scan / scan / cross_merge

    _GROUP_ORDER = (0, 2, 1, 3)   # scan group -> reference direction k (row-major, its reversal, column-major, its reversal)

    def _scan_core_dirs(self, x: torch.Tensor, channel_last: bool = False):
        """Same function as _scan_core with the direction permutations folded into the scan's loads and stores:
        only the row-major and the column-major copy of x are materialised (2 of the 4 `xs` slices of :403-404), the
        reversed directions read them backwards in place, x_proj runs once per orientation with the two directions'
        weights stacked (M = 2*(R+2N)), and the scan output is merged without un-flipping.  `channel_last`: x is (B, H, W, D),
        as the channel-last convolution leaves it (SS2D.forward)."""
        if channel_last:
            B, H, W, D = x.shape
        else:
            B, D, H, W = x.shape
        L, N, R = H * W, self.d_state, self.dt_rank
        Cx = R + 2 * N
        # reference direction k = 2*rev + orientation  ->  scan group 2*orientation + rev: swap the two index bits
        # (plain views / one fused kernel; no index tensor, so the step stays HIP-graph capturable)
        slots = GradSlots() if (x.is_cuda and x.requires_grad and not os.environ.get("SS2D_NO_GATHER6")) else None     # du and d(xs2) meet in one gather (backward)
        core = x.dtype == torch.float32 or (self.bf16_scan_io and x.dtype == torch.bfloat16)
        if core and L % 4 and x.is_cuda and not os.environ.get("SS2D_NO_ROW_PAD"):
            # 7 x 7 (L = 49; any L that is not a multiple of 4): rows padded to a multiple of 8 and zero in the pad, so that x_proj, dt_proj and the scan keep
            # their aligned kernels.  u = 0 there: a reversed scan (which meets the pad first) stays at state 0 until the
            # first real position, a forward scan reaches the pad behind the last one; x_dbl and dy are 0 in the pad, so no
            # parameter gradient sees it, and the gathers read the H*W real positions only.
            L = (L + 7) // 8 * 8
        if channel_last and not (core and x.is_cuda):
            x, channel_last = x.permute(0, 3, 1, 2), False      # the literal composition below wants (B, D, H, W)
        xs2 = dirs_scatter2(x, slots, L, channel_last)                                          # (B, 2, D, L)
        if core and x.is_cuda and self.x_proj_weight.dtype == torch.float32 and not os.environ.get("SS2D_NO_BLOCK_PARAMS"):
            # every parameter-derived operand of the call from one launch (forward) / into one launch (backward): the
            # stacked x_proj weight [orientation][rev*Cx + c], its transpose, dt_w and its transpose in the activations'
            # dtype, A = -exp(A_logs), D, bias regrouped
            A, Dv, bias, dt_w, xw, xwT, dtwT, dtw_io, pg = block_params(self.A_logs, self.Ds, self.dt_projs_bias, self.dt_projs_weight,
                                                                         self.x_proj_weight, x.dtype, want_dtw_io=R not in (6, 12))
            x_dbl = x_proj(xw, xs2, xwT, pg).view(B, 4, Cx, L)                          # group = 2*orientation + rev
            return scan_core_fn(xs2.reshape(B, 2 * D, L), x_dbl, dt_w, A, Dv, bias, H, W, x=x, holder=slots, dtwT=dtwT, dtw_io=dtw_io,
                                x_channel_last=channel_last)
        w_cat = self.x_proj_weight.view(2, 2, Cx, D).transpose(0, 1).reshape(2, 2 * Cx, D)     # [orientation][rev*Cx + c]
        x_dbl = x_proj(w_cat, xs2).view(B, 4, Cx, L)                                    # group = 2*orientation + rev
        A, Dv, bias, dt_w = scan_params(self.A_logs, self.Ds, self.dt_projs_bias, self.dt_projs_weight, regroup=True)
        if core:
            # dt_proj + scan + merge as one node reading x_dbl in place (rows [dts | B | C] of every block)
            return scan_core_fn(xs2.reshape(B, 2 * D, L), x_dbl, dt_w, A, Dv, bias, H, W, x=x, holder=slots, x_channel_last=channel_last)
        # the reference's fp32 casts around the scan (:423-426) under autocast
        dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
        dts = dt_proj(dts, dt_w)                                                                # (B, 4, D, L)
        return scan_merge_fn(xs2.float().reshape(B, 2 * D, L), dts.float().reshape(B, 4 * D, L), A,
                             Bs.float(), Cs.float(), Dv, bias, H, W, x=x)

    def _scan_core(self, x: torch.Tensor, channel_last: bool = False):
        """forward_corev0 up to (not including) out_norm: x (B, D, H, W) -> merged y (B, H, W, D)."""
        if self.fused_directions:
            return self._scan_core_dirs(x, channel_last)
        if channel_last:
            x = x.permute(0, 3, 1, 2)
        B, D, H, W = x.shape
        L, K, N, R = H * W, 4, self.d_state, self.dt_rank
        xs = cross_scan(x)                                                        # (B, 4, D, L)
        # the two einsums of :407/:420 as broadcast batched GEMMs on the tensors where they lie: einsum would
        # first copy xs (the 231 MB tensor) into a (k, b, l, d) layout and permute the results back
        x_dbl = torch.matmul(self.x_proj_weight.unsqueeze(0), xs)                 # (B, K, R+2N, L)
        dts, Bs, Cs = torch.split(x_dbl, [R, N, N], dim=2)
        dts = torch.matmul(self.dt_projs_weight.unsqueeze(0), dts)               # (B, K, D, L)
        # The reference casts the four scan inputs to fp32 here (:423-426), also under autocast.  With
        # `bf16_scan_io` (build's addition, off by default) bf16 activations go to the kernels as they are:
        # half the HBM bytes for u/delta/B/C/y and their gradients, fp32 arithmetic and state inside.
        cast = (lambda t: t) if (self.bf16_scan_io and xs.dtype == torch.bfloat16) else (lambda t: t.float())
        out_y = selective_scan_fn(
            cast(xs).reshape(B, K * D, L), cast(dts).reshape(B, K * D, L),
            -torch.exp(self.A_logs.float()).view(K * D, N),
            cast(Bs).reshape(B, K, N, L), cast(Cs).reshape(B, K, N, L),
            self.Ds.float().view(-1), z=None, delta_bias=self.dt_projs_bias.float().view(-1),
            delta_softplus=True, return_last_state=False)
        assert out_y.dtype == (torch.bfloat16 if (self.bf16_scan_io and xs.dtype == torch.bfloat16) else torch.float32)
        return cross_merge(out_y.view(B, K, D, L), H, W, channel_last=True)      # (B, H, W, D)

    # ---- mamba_sys.py:552-565
    def forward(self, x: torch.Tensor, residual=None, drop_scale=None, **kwargs):
        """`residual` / `drop_scale` (this build's additions, used by VSSBlock): returns residual + drop_scale[b] * SS2D(x)
        -- the block's residual connection with its stochastic-depth factor (mamba_sys.py:581-585) -- with the factor
        applied inside the out_norm + gate kernel (out_proj has no bias: scaling its input equals scaling its output) and
        the sum inside out_proj's GEMM."""
        D = self.d_inner
        xz = apply_linear(self.in_proj, x)                                         # (B, H, W, 2D)
        fused_gate = self.forward_core == self.forward_corev0 and isinstance(self.out_norm, nn.LayerNorm)
        if fused_gate and xz.is_cuda:
            # views whose gradients are written into the halves of ONE d(xz) buffer by the two consumers below
            halves = GradHalves()
            xh, z = split_halves(xz, halves)
        else:
            halves = None
            xh, z = torch.split(xz, D, dim=-1)     # views; split's backward is one cat instead of two zero-fills + add
        # the convolution output stays channel-last (no transposing tile in the convolution; the scatter reads either layout)
        cl = bool(fused_gate and xz.is_cuda and self.fused_directions and D % 4 == 0 and not os.environ.get("SS2D_NO_CL_CONV"))
        xc = dwconv3x3_silu(xh, self.conv2d.weight, self.conv2d.bias, channel_last_in=True, holder=halves, channel_last_out=cl)
        fold = (residual is not None and fused_gate and xz.is_cuda and self.dropout is None and self.out_proj.bias is None
                and not os.environ.get("SS2D_NO_LINEAR_RES"))
        if fused_gate:
            # out_norm, SiLU(z) and the product in one kernel (:452 + :561)
            y = outnorm_gate(self._scan_core(xc, cl), z, self.out_norm.weight, self.out_norm.bias, self.out_norm.eps, holder=halves,
                             scale=drop_scale if fold else None)
        else:
            y = self.forward_core(xc)
            y = y * F.silu(z)
        if fold:
            return linear_residual(self.out_proj, y, residual)
        out = apply_linear(self.out_proj, y)
        if self.dropout is not None:
            out = self.dropout(out)
        if residual is None:
            return out
        if drop_scale is None:
            return residual + out
        return _ResidualScaled.apply(residual, out, drop_scale.to(residual.dtype).view((out.shape[0],) + (1,) * (out.dim() - 1)))


class _ResidualScaled(torch.autograd.Function):
    """input + y * scale (scale per sample): the residual connection with stochastic depth as one pass each way.  Autograd's
    backward of addcmul is a multiply in the residual stream's dtype followed by a cast to y's; here the gradient of y is
    written in y's dtype by the multiply itself, and the residual gradient passes through untouched."""

    @staticmethod
    def forward(ctx, input, y, scale):
        ctx.save_for_backward(scale)
        ctx.y_dtype = y.dtype
        return torch.addcmul(input, y, scale)

    @staticmethod
    def backward(ctx, g):
        (scale,) = ctx.saved_tensors
        gy = None
        if ctx.needs_input_grad[1]:
            gy = torch.empty(g.shape, device=g.device, dtype=ctx.y_dtype)
            torch.mul(g, scale, out=gy)
        return (g if ctx.needs_input_grad[0] else None), gy, None


class VSSBlock(nn.Module):
    """x + DropPath(SS2D(LayerNorm(x)))  (mamba_sys.py:568-585)."""

    def __init__(self, hidden_dim: int = 0, drop_path: float = 0,
                 norm_layer: Callable[..., nn.Module] = partial(nn.LayerNorm, eps=1e-6),
                 attn_drop_rate: float = 0, d_state: int = 16, **kwargs):
        super().__init__()
        self.ln_1 = norm_layer(hidden_dim)
        self.self_attention = SS2D(d_model=hidden_dim, dropout=attn_drop_rate, d_state=d_state, **kwargs)
        self.drop_path = DropPath(drop_path)

    def forward(self, input: torch.Tensor):
        ln, input = apply_norm_skip(self.ln_1, input)     # `input` again, with its residual gradient routed through ln_1's backward kernel
        dp = self.drop_path
        if type(dp) is DropPath and type(self.self_attention) is SS2D:
            # residual + stochastic depth inside SS2D's last two kernels: input + SS2D(ln) * mask / keep (the same Bernoulli
            # draw per sample as timm's DropPath; the factor in fp32)
            scale = None
            if dp.training and dp.drop_prob > 0.0:
                scale = self._dp_scale
                if scale is not None and not self._dp_keep:
                    self._dp_scale = None                  # one forward only (kept for a checkpointed stage's recomputation)
                if scale is None or scale.shape[0] != input.shape[0] or scale.device != input.device:
                    keep = 1.0 - dp.drop_prob
                    scale = torch.empty(input.shape[0], device=input.device, dtype=torch.float32).bernoulli_(keep).div_(keep)
            return self.self_attention(ln, residual=input, drop_scale=scale)
        return input + dp(self.self_attention(ln))

    # per-sample scale mask / keep of this forward, drawn for all blocks of the network at once (VSSM.forward: one Bernoulli
    # launch per step instead of three small launches per block); None: the block draws its own
    _dp_scale = None
    _dp_keep = False


def _burn_out_proj_rng(module: nn.Module):
    """The reference re-initialises a detached CLONE of every SS2D.out_proj.weight right after building a
    stage ("fake init, just to keep the seed", mamba_sys.py:627-633): no parameter changes, but the
    generator advances.  Reproduced so that equal seeds give equal models."""
    for m in module.modules():
        if isinstance(m, SS2D):
            nn.init.kaiming_uniform_(m.out_proj.weight.detach().clone(), a=math.sqrt(5))


class _Stage(nn.Module):
    def __init__(self, dim, depth, attn_drop, drop_path, norm_layer, use_checkpoint, d_state):
        super().__init__()
        self.dim, self.use_checkpoint = dim, use_checkpoint
        self.blocks = nn.ModuleList([
            VSSBlock(hidden_dim=dim, drop_path=drop_path[i] if isinstance(drop_path, list) else drop_path,
                     norm_layer=norm_layer, attn_drop_rate=attn_drop, d_state=d_state)
            for i in range(depth)])
        _burn_out_proj_rng(self)

    def _run_blocks(self, x):
        for blk in self.blocks:
            x = checkpoint.checkpoint(blk, x, use_reentrant=False) if self.use_checkpoint else blk(x)
        return x


class VSSLayer(_Stage):
    """Encoder stage: `depth` VSSBlocks then an optional PatchMerging2D (mamba_sys.py:588-651)."""

    def __init__(self, dim, depth, attn_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm, downsample=None,
                 use_checkpoint=False, d_state=16, **kwargs):
        super().__init__(dim, depth, attn_drop, drop_path, norm_layer, use_checkpoint, d_state)
        self.downsample = downsample(dim=dim, norm_layer=norm_layer) if downsample is not None else None

    def forward(self, x):
        x = self._run_blocks(x)
        return self.downsample(x) if self.downsample is not None else x


class VSSLayer_up(_Stage):
    """Decoder stage: `depth` VSSBlocks then an optional PatchExpand (mamba_sys.py:653-716)."""

    def __init__(self, dim, depth, attn_drop=0.0, drop_path=0.0, norm_layer=nn.LayerNorm, upsample=None,
                 use_checkpoint=False, d_state=16, **kwargs):
        super().__init__(dim, depth, attn_drop, drop_path, norm_layer, use_checkpoint, d_state)
        from .vssm import PatchExpand  # the reference always builds PatchExpand here, whatever `upsample` is
        self.upsample = PatchExpand(dim, dim_scale=2, norm_layer=nn.LayerNorm) if upsample is not None else None

    def forward(self, x):
        x = self._run_blocks(x)
        return self.upsample(x) if self.upsample is not None else x

"""VSSM (the Mamba-UNet U-shaped network) and the MambaUnet wrapper on top of the HIP SS2D path.

Mirrors the reference's module tree, state-dict keys and construction-time RNG order
(code/networks/mamba_sys.py:166-265 patch ops, :719-876 VSSM; code/networks/vision_mamba.py:23-83
MambaUnet) so that checkpoints and seeds carry over.  Patch embedding, merging and expansion run on the package's
GEMM / LayerNorm kernels (ss2d_ops); the 1x1 convolutions of the heads are fused into their LayerNorm.
"""
from __future__ import annotations

import copy
import math
import os
from dataclasses import dataclass, field
from typing import List, Optional

import torch
import torch.nn as nn

from .ss2d import VSSLayer, VSSLayer_up
from .ss2d_ops import linear, apply_linear, apply_norm, apply_norm_merge, apply_norm_unshuffle, concat_linear, ln_unshuffle_head


class PatchEmbed2D(nn.Module):
    """conv(patch, stride patch) -> channel-last -> LayerNorm (mamba_sys.py:166-188)."""

    def __init__(self, patch_size=4, in_chans=3, embed_dim=96, norm_layer=None, **kwargs):
        super().__init__()
        if isinstance(patch_size, int):
            patch_size = (patch_size, patch_size)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = norm_layer(embed_dim) if norm_layer is not None else None

    def forward(self, x):
        pj = self.proj
        (ph, pw), (B, C, H, W) = pj.kernel_size, x.shape
        if (x.is_cuda and pj.stride == (ph, pw) and pj.padding == (0, 0) and pj.dilation == (1, 1) and pj.groups == 1 and H % ph == 0
                and W % pw == 0 and (C * ph * pw) % 8 == 0 and not os.environ.get("SS2D_PATCH_CONV")):
            # non-overlapping patches: the convolution IS a Linear over (c, kh, kw).  One copy gathers (and, under autocast,
            # casts) the patches, the product runs on the token GEMM with the bias, the weight gradient on the split-token
            # kernel (MIOpen at batch 24: 54 us forward + 104 us weight gradient; this: ~25 + ~25)
            dt = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else x.dtype
            Hp, Wp = H // ph, W // pw
            patches = torch.empty((B, Hp, Wp, C * ph * pw), device=x.device, dtype=dt)
            patches.view(B, Hp, Wp, C, ph, pw).copy_(x.view(B, C, Hp, ph, Wp, pw).permute(0, 2, 4, 1, 3, 5))
            x = linear(patches, pj.weight.view(pj.out_channels, -1), pj.bias)
        else:
            x = pj(x).permute(0, 2, 3, 1)
        return apply_norm(self.norm, x) if self.norm is not None else x


class PatchMerging2D(nn.Module):
    """2x2 neighbourhood -> 4C -> LayerNorm -> Linear 4C->2C (mamba_sys.py:191-231)."""

    def __init__(self, dim, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.reduction = nn.Linear(4 * dim, 2 * dim, bias=False)
        self.norm = norm_layer(4 * dim)

    def forward(self, x):
        H, W = x.shape[1], x.shape[2]
        if (H % 2) or (W % 2):
            print(f"Warning, x.shape {x.shape} is not match even ===========", flush=True)
        return apply_linear(self.reduction, apply_norm_merge(self.norm, x))


class PatchExpand(nn.Module):
    """Linear C->2C, 2x2 pixel un-shuffle, LayerNorm(C/2) (mamba_sys.py:233-247)."""

    def __init__(self, dim, dim_scale=2, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim = dim
        self.expand = nn.Linear(dim, 2 * dim, bias=False) if dim_scale == 2 else nn.Identity()
        self.norm = norm_layer(dim // dim_scale)

    def forward(self, x):
        return apply_norm_unshuffle(self.norm, apply_linear(self.expand, x), 2)


class FinalPatchExpand_X4(nn.Module):
    """Linear C->16C, 4x4 pixel un-shuffle, LayerNorm(C) (mamba_sys.py:249-265)."""

    def __init__(self, dim, dim_scale=4, norm_layer=nn.LayerNorm):
        super().__init__()
        self.dim, self.dim_scale = dim, dim_scale
        self.expand = nn.Linear(dim, 16 * dim, bias=False)
        self.output_dim = dim
        self.norm = norm_layer(self.output_dim)

    def forward(self, x):
        return apply_norm_unshuffle(self.norm, apply_linear(self.expand, x), self.dim_scale)


class VSSM(nn.Module):
    """mamba_sys.py:719-876.  Note (as in the reference) that only dims[0] is used: stage i has width
    dims[0] * 2**i."""

    def __init__(self, patch_size=4, in_chans=1, num_classes=4, depths=[2, 2, 9, 2], dims=[96, 192, 384, 768],
                 d_state=16, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.1, norm_layer=nn.LayerNorm,
                 patch_norm=True, use_checkpoint=False, final_upsample="expand_first", **kwargs):
        super().__init__()
        self.num_classes = num_classes
        self.num_layers = n = len(depths)
        if isinstance(dims, int):
            dims = [int(dims * 2 ** i) for i in range(n)]
        self.embed_dim, self.num_features = dims[0], dims[-1]
        self.num_features_up = int(dims[0] * 2)
        self.dims, self.final_upsample = dims, final_upsample
        width = lambda i: int(dims[0] * 2 ** i)  # noqa: E731
        state = math.ceil(dims[0] / 6) if d_state is None else d_state

        self.patch_embed = PatchEmbed2D(patch_size=patch_size, in_chans=in_chans, embed_dim=self.embed_dim,
                                        norm_layer=norm_layer if patch_norm else None)
        dpr = [v.item() for v in torch.linspace(0, drop_path_rate, sum(depths))]
        span = lambda i: dpr[sum(depths[:i]):sum(depths[:i + 1])]  # noqa: E731

        self.layers = nn.ModuleList()
        for i in range(n):
            self.layers.append(VSSLayer(dim=width(i), depth=depths[i], d_state=state, drop=drop_rate,
                                        attn_drop=attn_drop_rate, drop_path=span(i), norm_layer=norm_layer,
                                        downsample=PatchMerging2D if i < n - 1 else None,
                                        use_checkpoint=use_checkpoint))

        self.layers_up = nn.ModuleList()
        self.concat_back_dim = nn.ModuleList()
        for i in range(n):
            j = n - 1 - i  # mirrored encoder stage
            concat = nn.Linear(2 * width(j), width(j)) if i > 0 else nn.Identity()
            if i == 0:
                up = PatchExpand(dim=width(j), dim_scale=2, norm_layer=norm_layer)
            else:
                up = VSSLayer_up(dim=width(j), depth=depths[j], d_state=state, drop=drop_rate,
                                 attn_drop=attn_drop_rate, drop_path=span(j), norm_layer=norm_layer,
                                 upsample=PatchExpand if i < n - 1 else None, use_checkpoint=use_checkpoint)
            self.layers_up.append(up)
            self.concat_back_dim.append(concat)

        self.norm = norm_layer(self.num_features)
        self.norm_up = norm_layer(self.embed_dim)
        if self.final_upsample == "expand_first":
            self.up = FinalPatchExpand_X4(dim_scale=4, dim=self.embed_dim)
            self.output = nn.Conv2d(self.embed_dim, self.num_classes, kernel_size=1, bias=False)
        self.apply(self._init_weights)

    def _init_weights(self, m: nn.Module):
        # mamba_sys.py:792-808: every nn.Linear -> trunc_normal(.02), zero bias; LayerNorm -> (1, 0);
        # convs and the bare SS2D parameters keep their constructor init
        if isinstance(m, nn.Linear):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    def _draw_stochastic_depth(self, x):
        """The per-sample keep masks of every VSSBlock of this forward (timm's DropPath, mamba_sys.py:581) from ONE Bernoulli
        draw: (blocks, batch) with each block's own keep probability, scaled by 1 / keep in the residual stream's dtype."""
        from .ss2d import DropPath
        live = [(blk, stage.use_checkpoint) for stage in (*self.layers, *self.layers_up) if hasattr(stage, "blocks") for blk in stage.blocks
                if type(blk.drop_path) is DropPath and blk.drop_path.training and blk.drop_path.drop_prob > 0.0]
        if not live or not x.is_cuda or os.environ.get("SS2D_NO_DP_PREDRAW"):     # (A/B switch: every block draws its own)
            return
        key = (x.device, tuple(blk.drop_path.drop_prob for blk, _ in live))
        if getattr(self, "_dp_key", None) != key:
            self._dp_key, self._dp_keep_prob = key, torch.tensor([1.0 - q for q in key[1]], device=x.device).view(-1, 1)
        keep = self._dp_keep_prob
        scale = torch.bernoulli(keep.expand(len(live), x.shape[0])).div_(keep)       # fp32: applied inside the out_norm + gate kernel
        for i, (blk, ckpt) in enumerate(live):
            blk._dp_scale, blk._dp_keep = scale[i], bool(ckpt)

    def forward_features(self, x):
        x = self.patch_embed(x)
        if self.training:
            self._draw_stochastic_depth(x)
        skips = []
        for layer in self.layers:
            skips.append(x)
            x = layer(x)
        return apply_norm(self.norm, x), skips

    def forward_up_features(self, x, x_downsample):
        for i, up in enumerate(self.layers_up):
            if i > 0:
                x = concat_linear(self.concat_back_dim[i], x, x_downsample[3 - i])       # mamba_sys.py:835 without the cat
            x = up(x)
        return apply_norm(self.norm_up, x)

    def up_x4(self, x):
        if self.final_upsample == "expand_first":
            # FinalPatchExpand_X4 (expand, 4x4 un-shuffle, norm) + permute + 1x1 `output` conv; norm and conv run as
            # one kernel that never writes the (B, 4H, 4W, C) tensor between them
            x = ln_unshuffle_head(self.up.norm, apply_linear(self.up.expand, x), self.up.dim_scale, self.output)
        return x

    def forward(self, x):
        x, skips = self.forward_features(x)
        return self.up_x4(self.forward_up_features(x, skips))


@dataclass
class MambaUnetConfig:
    """The four fields of the reference's yacs config that reach VSSM (vision_mamba.py:30-40;
    defaults of configs/vmamba_tiny.yaml)."""
    patch_size: int = 4
    in_chans: int = 3
    depths: List[int] = field(default_factory=lambda: [2, 2, 2, 2])
    drop_path_rate: float = 0.2
    pretrain_ckpt: Optional[str] = None

    @classmethod
    def from_reference(cls, config):
        """Accepts the reference's yacs-style object (config.MODEL.VSSM.PATCH_SIZE ...)."""
        m = config.MODEL
        return cls(m.VSSM.PATCH_SIZE, m.VSSM.IN_CHANS, list(m.VSSM.DEPTHS), m.DROP_PATH_RATE,
                   getattr(m, "PRETRAIN_CKPT", None))


class MambaUnet(nn.Module):
    """vision_mamba.py:23-83: 1->3 channel repeat, VSSM, checkpoint loading with the encoder->decoder remap."""

    def __init__(self, config=None, img_size=224, num_classes=21843, zero_head=False, vis=False):
        super().__init__()
        if config is None:
            config = MambaUnetConfig()
        elif not isinstance(config, MambaUnetConfig):
            config = MambaUnetConfig.from_reference(config)
        self.num_classes, self.zero_head, self.config = num_classes, zero_head, config
        self.mamba_unet = VSSM(patch_size=config.patch_size, in_chans=config.in_chans, num_classes=num_classes,
                               depths=config.depths, drop_path_rate=config.drop_path_rate)

    def forward(self, x):
        if x.size(1) == 1:
            x = x.repeat(1, 3, 1, 1)
        return self.mamba_unet(x)

    def load_from(self, config=None):
        """Loads `pretrain_ckpt` (tensors only); encoder stage `layers.N` also initialises decoder stage
        `layers_up.(3-N)`; shape-mismatched entries are dropped (vision_mamba.py:48-83)."""
        path = (config or self.config).pretrain_ckpt if isinstance(config or self.config, MambaUnetConfig) \
            else config.MODEL.PRETRAIN_CKPT
        if path is None:
            print("none pretrain")
            return None
        dev = next(self.parameters()).device
        ckpt = torch.load(path, map_location=dev, weights_only=True)
        if "model" not in ckpt:
            sd = {k[17:]: v for k, v in ckpt.items()}  # vision_mamba.py:57
            sd = {k: v for k, v in sd.items() if "output" not in k}
            return self.mamba_unet.load_state_dict(sd, strict=False)
        src = ckpt["model"]
        full = copy.deepcopy(src)
        for k, v in src.items():
            if k.startswith("layers."):
                full["layers_up." + str(3 - int(k[7:8])) + k[8:]] = v
        own = self.mamba_unet.state_dict()
        for k in list(full):
            if k in own and full[k].shape != own[k].shape:
                del full[k]
        return self.mamba_unet.load_state_dict(full, strict=False)

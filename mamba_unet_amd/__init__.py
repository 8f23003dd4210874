"""MI355X-native SS2D hot path of Mamba-UNet (selective scan, cross-scan/merge, dwconv+SiLU).

Host-side mirror of the reference's operator interface (code/networks/mamba_sys.py,
code/networks/vision_mamba.py): `selective_scan_fn`, `SS2D`, `VSSBlock`, `VSSLayer`, `VSSLayer_up`,
`VSSM`, `MambaUnet`.  All hot-path compute goes through libss2d_hip.so (include/ss2d.h); there is no
CPU path."""
from .selective_scan_interface import SelectiveScanFn, scan_core_fn, scan_merge_fn, selective_scan_fn  # noqa: F401
from .ss2d import SS2D, DropPath, VSSBlock, VSSLayer, VSSLayer_up  # noqa: F401
from .ss2d_ops import (cross_merge, cross_scan, dwconv3x3_silu, layer_norm, linear_nobias, linear_wgrad, outnorm_gate,  # noqa: F401
                       scan_params)
from .vssm import (VSSM, FinalPatchExpand_X4, MambaUnet, MambaUnetConfig, PatchEmbed2D,  # noqa: F401
                   PatchExpand, PatchMerging2D)

__all__ = ["selective_scan_fn", "SelectiveScanFn", "SS2D", "VSSBlock", "VSSLayer", "VSSLayer_up", "VSSM",
           "MambaUnet", "MambaUnetConfig", "cross_scan", "cross_merge", "dwconv3x3_silu", "DropPath",
           "PatchEmbed2D", "PatchMerging2D", "PatchExpand", "FinalPatchExpand_X4"]

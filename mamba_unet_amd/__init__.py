"""MI355X-native SS2D hot path of Mamba-UNet (selective scan, cross-scan/merge, dwconv+SiLU).

Host-side mirror of the reference's operator interface (code/networks/mamba_sys.py):
`selective_scan_fn`, `SS2D`, `VSSBlock`, `VSSLayer`, `VSSLayer_up`, `VSSM`, `MambaUnet`.
All compute goes through libss2d_hip.so (include/ss2d.h); there is no CPU path."""
from .selective_scan_interface import selective_scan_fn  # noqa: F401

__all__ = ["selective_scan_fn"]

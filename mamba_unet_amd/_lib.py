"""ctypes binding of libss2d_hip.so (C ABI declared in include/ss2d.h).

There is no CPU fallback: if the shared library is missing or a call fails, this raises."""
from __future__ import annotations

import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SS2D_HIP_LIB", os.path.join(_HERE, "libss2d_hip.so"))  # env override: A/B builds
ABI_VERSION = 1

_vp, _fp, _i, _sz = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t

_SIGNATURES = {
    "ss2d_abi_version": (_i, []),
    "ss2d_last_error": (ctypes.c_char_p, []),
    "ss2d_scan_num_chunks": (_i, [_i]),
    "ss2d_scan_ckpt_bytes": (_sz, [_i, _i, _i, _i]),
    "ss2d_selective_scan_fwd": (_i, [_i] + [_vp] * 11 + [_i] * 6 + [_vp]),
    "ss2d_selective_scan_bwd": (_i, [_i] + [_vp] * 18 + [_i] * 6 + [_vp]),
    "ss2d_selective_scan_dir_fwd": (_i, [_i] + [_vp] * 11 + [_i] * 7 + [ctypes.c_uint, ctypes.c_longlong, _vp]),
    "ss2d_selective_scan_dir_bwd": (_i, [_i] + [_vp] * 18 + [_i] * 7 + [ctypes.c_uint, ctypes.c_longlong, _vp]),
    "ss2d_scan_bwd_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "ss2d_selective_scan_dir_bwd_ws": (_i, [_i] + [_vp] * 18 + [_i] * 7 + [ctypes.c_uint, ctypes.c_longlong, _vp, _sz, _vp]),
    "ss2d_selective_scan_dir_bwd_static": (_i, [_i] + [_vp] * 18 + [_i] * 7 + [ctypes.c_uint, ctypes.c_longlong, _vp]),
    "ss2d_selective_scan_core_fwd": (_i, [_i] + [_vp] * 8 + [_i] * 7 + [ctypes.c_uint, _vp]),
    "ss2d_selective_scan_core_bwd": (_i, [_i] + [_vp] * 15 + [_i] * 7 + [ctypes.c_uint, _vp, _sz, _vp]),
    "ss2d_selective_scan_core_bwd_packed": (_i, [_i] + [_vp] * 14 + [_i] * 7 + [ctypes.c_uint, _vp, _sz, _vp]),
    "ss2d_cross_scan_fwd": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ss2d_cross_scan_bwd": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ss2d_cross_merge_fwd": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ss2d_cross_merge_bwd": (_i, [_i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ss2d_dirs_scatter": (_i, [_i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ss2d_dirs_gather": (_i, [_i, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "ss2d_dirs_gather6": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "ss2d_dirs_scatter_ld": (_i, [_i, _i, _vp, _vp, _i, _i, _i, _i, _i, ctypes.c_longlong, _vp]),
    "ss2d_dirs_gather_ld": (_i, [_i, _i, _vp, _vp, _i, _i, _i, _i, _i, ctypes.c_longlong, _vp]),
    "ss2d_dirs_gather6_ld": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, ctypes.c_longlong, _vp]),
    "ss2d_dirs_gather6_cl_ld": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, ctypes.c_longlong, _vp]),
    "ss2d_dwconv3x3_silu_cl_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ss2d_dwconv3x3_silu_cl_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ss2d_dwconv3x3_silu_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "ss2d_dwconv3x3_silu_bwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ss2d_dt_proj_fwd": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, ctypes.c_longlong, _vp]),
    "ss2d_scan_params_fwd": (_i, [_vp] * 8 + [_i, _i, _i, _i, _i, _vp]),
    "ss2d_scan_params_bwd": (_i, [_vp] * 9 + [_i, _i, _i, _i, _i, _vp]),
    "ss2d_block_params_fwd": (_i, [_i] + [_vp] * 13 + [_i] * 6 + [_vp]),
    "ss2d_block_params_bwd": (_i, [_vp] * 11 + [_i, _i, _i, _vp]),
    "ss2d_xdbl_grad_pack": (_i, [_i, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _i, _vp]),
    "ss2d_linear_wgrad": (_i, [_i, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _i, _i, _i, _vp]),
    "ss2d_outnorm_gate_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, ctypes.c_float, _vp]),
    "ss2d_outnorm_gate_bwd": (_i, [_i] + [_vp] * 11 + [ctypes.c_longlong, _i, _i, _i, _vp]),
    "ss2d_outnorm_gate_scaled_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, ctypes.c_float, _vp]),
    "ss2d_outnorm_gate_scaled_bwd": (_i, [_i] + [_vp] * 8 + [ctypes.c_longlong] + [_vp] * 4 + [ctypes.c_longlong, _i, _i, _i, _vp]),
    "ss2d_layernorm_fwd": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, ctypes.c_float, _vp]),
    "ss2d_layernorm_bwd": (_i, [_i, _i] + [_vp] * 8 + [ctypes.c_longlong, _i, _vp]),
    "ss2d_layernorm_bwd_add": (_i, [_i, _i] + [_vp] * 9 + [ctypes.c_longlong, _i, _vp]),
    "ss2d_layernorm_unshuffle_fwd": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _i, _i, ctypes.c_float, _vp]),
    "ss2d_ln_head_fwd": (_i, [_i, _i] + [_vp] * 7 + [ctypes.c_longlong, _i, _i, _i, _i, _i, ctypes.c_float, _vp]),
    "ss2d_ln_head_bwd": (_i, [_i, _i] + [_vp] * 11 + [ctypes.c_longlong, _i, _i, _i, _i, _i, _vp]),
    "ss2d_chan_wgrad": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i] + [ctypes.c_longlong] * 7 + [_vp]),
    "ss2d_chan_gemm": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i] + [ctypes.c_longlong] * 5 + [_vp]),
    "ss2d_token_gemm": (_i, [_i, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _i, _i, _i, _i, _vp]),
    "ss2d_token_gemm_cat": (_i, [_i, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "ss2d_token_gemm_res": (_i, [_i, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _i, _i, _i, _i, _vp]),
    "ss2d_colsum_workspace_floats": (_sz, [_i]),
    "ss2d_colsum": (_i, [_i, _vp, _vp, _vp, ctypes.c_longlong, _i, ctypes.c_longlong, _vp]),
    "ss2d_seg_loss_workspace_floats": (_sz, [_i]),
    "ss2d_seg_loss_stats": (_i, [_i, _vp, _vp, _i, _vp, _vp, _i, _i, ctypes.c_longlong, _vp]),
    "ss2d_seg_loss_bwd": (_i, [_i, _vp, _vp, _i, _vp, _vp, _i, _i, ctypes.c_longlong, _vp]),
    "ss2d_layernorm_unshuffle_bwd": (_i, [_i, _i] + [_vp] * 8 + [ctypes.c_longlong, _i, _i, _i, _i, _vp]),
    "ss2d_layernorm_merge_fwd": (_i, [_i, _i, _vp, _vp, _vp, _vp, _vp, _vp, ctypes.c_longlong, _i, _i, _i, ctypes.c_float, _vp]),
    "ss2d_layernorm_merge_bwd": (_i, [_i, _i] + [_vp] * 8 + [ctypes.c_longlong, _i, _i, _i, _vp]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


class SS2DLibraryError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the HIP library once; raise loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SS2DLibraryError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C mamba_unet_amd/csrc`")
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        if lib.ss2d_abi_version() != ABI_VERSION:
            raise SS2DLibraryError(f"ABI mismatch: library {lib.ss2d_abi_version()} != binding {ABI_VERSION}")
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().ss2d_last_error().decode(errors="replace")
        if rc < 0:
            raise ValueError(f"{what}: {msg} (rc={rc})")
        raise RuntimeError(f"{what}: {msg} (hipError {rc})")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class _Here:
    """The no-op context of `on_device` (one shared instance)."""

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_CTX = _Here()


def on_device(dev):
    """`with on_device(t.device):` around a launch -- `torch.cuda.device(dev)` only when dev is not already the current
    device.  The context manager costs ~8 us of host time per use and a training step launches ~600 kernels of this
    library: the host, not the GPU, set the pace of the eager step on a box with slow cores."""
    import torch
    idx = dev.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_CTX
    return torch.cuda.device(dev)


def stream():
    """Raw handle of torch's current stream on the current device (the same value as
    torch.cuda.current_stream().cuda_stream without building a Stream object: 0.3 us instead of 9)."""
    import torch
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


_ARENA = {}                 # (device, raw stream) -> [zero-filled fp32 chunk, floats handed out, capture epoch]
_ARENA_EPOCH = [0, False]   # bumped whenever the stream's capture state changes between two requests
_ARENA_FLOATS = 1 << 23     # 32 MB per chunk: a training step of MambaUNet zero-fills 19 M gradient floats -- three chunks
_ARENA_MAX_REQ = 1 << 21    # requests above 8 MB get a buffer (and a fill) of their own
_ARENA_LOCK = None


def zeros_flat(device, n):
    """n fp32 zeros (1-D).  Small requests are carved out of a zero-filled 32 MB chunk per device, so the ~100 accumulated
    gradient buffers of a training step (dgamma, dbeta, dA, dD, bias and small weight gradients: float-atomic targets)
    cost a handful of fill launches instead of one each.  A chunk is never reused -- its pieces must be zero -- and
    lives as long as any piece does; pieces start on 256-byte boundaries.  A chunk filled outside a HIP-graph capture is
    not used inside one (its fill would not be replayed) and vice versa.  Chunks are kept per (device, stream): the fill
    runs on the stream that is current when the chunk is made, so a piece handed to work on ANOTHER stream would have no
    ordering against that fill (and the caching allocator could recycle the chunk under it)."""
    import threading
    import torch
    global _ARENA_LOCK
    if n > _ARENA_MAX_REQ or os.environ.get("SS2D_NO_ZERO_ARENA"):      # (env: A/B switch)
        return torch.zeros((n,), device=device, dtype=torch.float32)
    if _ARENA_LOCK is None:
        _ARENA_LOCK = threading.Lock()
    need = (n + 63) // 64 * 64
    idx = device.index if getattr(device, "index", None) is not None else torch.cuda.current_device()
    key = (idx, torch._C._cuda_getCurrentRawStream(idx))
    with _ARENA_LOCK:          # forward and autograd threads both come here
        cap = torch.cuda.is_current_stream_capturing()
        if cap != _ARENA_EPOCH[1]:
            _ARENA_EPOCH[0] += 1
            _ARENA_EPOCH[1] = cap
        slot = _ARENA.get(key)
        if slot is None or slot[1] + need > _ARENA_FLOATS or slot[2] != _ARENA_EPOCH[0]:
            slot = _ARENA[key] = [torch.zeros((_ARENA_FLOATS,), device=device, dtype=torch.float32), 0, _ARENA_EPOCH[0]]
        o = slot[1]
        slot[1] = o + need
        return slot[0][o:o + n]


def zeros(device, shape):
    """fp32 zeros of `shape` (see zeros_flat)."""
    import math
    return zeros_flat(device, math.prod(shape)).view(shape)


def zeros_group(device, *shapes):
    """fp32 zero tensors of the given shapes (None -> None), one request to zeros_flat for all of them: the accumulated
    gradients of a backward (dA, dD, dbias ...) share a fill launch.  Pieces start on 256-byte boundaries."""
    import math
    offs, total = [], 0
    for sh in shapes:
        offs.append(total)
        if sh is not None:
            total += (math.prod(sh) + 63) // 64 * 64
    buf = zeros_flat(device, total)
    return [None if sh is None else buf[o:o + math.prod(sh)].view(sh) for sh, o in zip(shapes, offs)]

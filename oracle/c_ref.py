"""ORACLE loader (test infrastructure): ctypes binding of oracle/libscan_ref.so (oracle/scan_ref.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_fp = ctypes.POINTER(ctypes.c_float)


def build() -> str:
    subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return os.path.join(_HERE, "libscan_ref.so")


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libscan_ref.so")
        if not os.path.exists(path):
            build()
        _LIB = ctypes.CDLL(path)
        _LIB.scan_ref_fwd_f32.restype = ctypes.c_int
        _LIB.scan_ref_bwd_f32.restype = ctypes.c_int
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(_fp)


def _c(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def scan_fwd(u, delta, A, B, C, D=None, z=None, delta_bias=None, delta_softplus=False,
             return_last_state=False, nthreads=0):
    u, delta, A, B, C, D, z, delta_bias = map(_c, (u, delta, A, B, C, D, z, delta_bias))
    if B.ndim == 3:
        B, C = B[:, None], C[:, None]
    batch, dim, L = u.shape
    N, G = A.shape[1], B.shape[1]
    out = np.empty_like(u)
    last = np.empty((batch, dim, N), np.float32) if return_last_state else None
    rc = lib().scan_ref_fwd_f32(_p(u), _p(delta), _p(A), _p(B), _p(C), _p(D), _p(z), _p(delta_bias),
                                _p(out), _p(last), batch, dim, L, N, G, int(delta_softplus), nthreads)
    assert rc == 0
    return (out, last) if return_last_state else out


def scan_bwd(u, delta, A, B, C, D, z, delta_bias, dout, delta_softplus=False, nthreads=0):
    u, delta, A, B, C, D, z, delta_bias, dout = map(_c, (u, delta, A, B, C, D, z, delta_bias, dout))
    squeeze = B.ndim == 3
    if squeeze:
        B, C = B[:, None], C[:, None]
    batch, dim, L = u.shape
    N, G = A.shape[1], B.shape[1]
    du, ddelta = np.empty_like(u), np.empty_like(u)
    dA = np.empty_like(A)
    dB, dC = np.empty_like(B), np.empty_like(C)
    dD = np.empty(dim, np.float32) if D is not None else None
    dz = np.empty_like(u) if z is not None else None
    dbias = np.empty(dim, np.float32) if delta_bias is not None else None
    rc = lib().scan_ref_bwd_f32(_p(u), _p(delta), _p(A), _p(B), _p(C), _p(D), _p(z), _p(delta_bias), _p(dout),
                                _p(du), _p(ddelta), _p(dA), _p(dB), _p(dC), _p(dD), _p(dz), _p(dbias),
                                batch, dim, L, N, G, int(delta_softplus), nthreads)
    assert rc == 0
    if squeeze:
        dB, dC = dB[:, 0], dC[:, 0]
    return dict(du=du, ddelta=ddelta, dA=dA, dB=dB, dC=dC, dD=dD, dz=dz, ddelta_bias=dbias)

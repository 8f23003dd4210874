"""Synthetic workloads of the hot path: plain tensor generators, no arithmetic of the path itself.
bench.py, tools/ and the tests all draw their scan inputs here; the oracle re-exports the function."""
from __future__ import annotations

import math

import torch


def make_scan_inputs(batch, dim, L, dstate=16, ngroups=4, seed=0, integer_A=True, dtype=torch.float32):
    """CPU tensors distributed like the arguments of the SS2D call site `mamba_sys.py:436-442` (init statics
    `:340-394`): u ~ N(0,1), delta ~ 0.5 N(0,1), grouped B / C ~ N(0,1); `integer_A` gives A_log_init's
    A = -(1..N) and D = 1, otherwise A = -exp(N(0,1)) and D ~ N(0,1); delta_bias = softplus^-1(dt) of dt_init."""
    g = torch.Generator().manual_seed(seed)
    u = torch.randn(batch, dim, L, generator=g, dtype=dtype)
    delta = 0.5 * torch.randn(batch, dim, L, generator=g, dtype=dtype)
    Bm = torch.randn(batch, ngroups, dstate, L, generator=g, dtype=dtype)
    Cm = torch.randn(batch, ngroups, dstate, L, generator=g, dtype=dtype)
    if integer_A:
        A = -torch.arange(1, dstate + 1, dtype=torch.float32).repeat(dim, 1)        # A_log_init
    else:
        A = -torch.exp(torch.randn(dim, dstate, generator=g))
    D = torch.ones(dim) if integer_A else torch.randn(dim, generator=g)
    dt = torch.exp(torch.rand(dim, generator=g) * (math.log(0.1) - math.log(0.001)) + math.log(0.001)).clamp(min=1e-4)
    delta_bias = dt + torch.log(-torch.expm1(-dt))                                   # dt_init :354-361
    return dict(u=u, delta=delta, A=A, B=Bm, C=Cm, D=D, delta_bias=delta_bias)

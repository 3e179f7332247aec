"""Layer-level ops used by the removal networks.

Every function here dispatches to a hand-written HIP kernel (csrc/*.hip) through the C
ABI.  Ops whose kernel is not written yet run through torch-ROCm on the GPU and are
recorded in ``INTERIM`` -- they never fall back to the CPU or to the oracle, and
``RFX_STRICT_NATIVE=1`` turns any interim op into an error (DESIGN.md keeps the
coverage table).
"""
import os

import torch
import torch.nn.functional as F

from . import _lib, ops
from ._lib import check
from .ops import _ptr, _stream

INTERIM = set()
_STRICT = os.environ.get("RFX_STRICT_NATIVE", "0") == "1"


def _interim(name):
    if _STRICT:
        raise RuntimeError(f"op '{name}' has no HIP kernel yet (RFX_STRICT_NATIVE=1)")
    INTERIM.add(name)


def gelu(x):
    return ops.activation(x, "gelu")


GN_MODES = {"none": 0, "gelu": 1, "glu": 2, "glu_scale_res": 3}


class _GroupNormFn(torch.autograd.Function):
    """GroupNorm fused with GELU / GLU / (res + scale * GLU); backward re-materialises gn(x)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, mode, res, scale):
        ops._req(x, "x")
        x = x.contiguous()
        N, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (N * Cc)
        oshape = list(x.shape)
        if mode >= 2:
            oshape[1] = Cc // 2
        y = torch.empty(oshape, device=x.device, dtype=torch.float32)
        mean = torch.empty(N * groups, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        sums = torch.empty(N * groups * 2, device=x.device, dtype=torch.float64)
        if res is not None:
            res = res.contiguous()
        check(_lib.lib().rfx_groupnorm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), N, Cc, S, groups, eps, mode,
                                           _ptr(res), _ptr(scale), _ptr(sums), _ptr(mean), _ptr(rstd), _ptr(y),
                                           _stream()),
              "rfx_groupnorm_fwd")
        ctx.save_for_backward(x, gamma, beta, mean, rstd, scale)
        ctx.cfg = (N, Cc, S, groups, mode)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, beta, mean, rstd, scale = ctx.saved_tensors
        N, Cc, S, groups, mode = ctx.cfg
        gy = gy.contiguous()
        dx = torch.empty_like(x)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(beta)
        dscale = torch.empty_like(scale) if mode == 3 else None
        gsum = torch.empty(N * Cc * 2 + N * (Cc // 2) + N * groups * 2, device=x.device, dtype=torch.float32)
        check(_lib.lib().rfx_groupnorm_bwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd), _ptr(gy),
                                           N, Cc, S, groups, mode, _ptr(scale), _ptr(gsum), _ptr(dx),
                                           _ptr(dgamma), _ptr(dbeta), _ptr(dscale), _stream()),
              "rfx_groupnorm_bwd")
        return dx, dgamma, dbeta, None, None, None, (gy if mode == 3 else None), dscale


def group_norm(x, groups, weight, bias, eps=1e-5, mode="none", res=None, scale=None):
    """mode: none | gelu | glu | glu_scale_res (out = res + scale[c] * glu(gn(x)))."""
    return _GroupNormFn.apply(x, weight, bias, groups, eps, GN_MODES[mode], res, scale)


class _GluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ops._req(x, "x")
        x = x.contiguous()
        N, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (N * Cc)
        shp = list(x.shape)
        shp[1] = Cc // 2
        y = torch.empty(shp, device=x.device, dtype=torch.float32)
        check(_lib.lib().rfx_glu_fwd(_ptr(x), _ptr(y), N, Cc, S, _stream()), "rfx_glu_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        N, Cc = x.shape[0], x.shape[1]
        S = x.numel() // (N * Cc)
        gx = torch.empty_like(x)
        check(_lib.lib().rfx_glu_bwd(_ptr(x), _ptr(gy.contiguous()), _ptr(gx), N, Cc, S, _stream()), "rfx_glu_bwd")
        return gx


def glu(x, dim=1):
    if dim != 1:
        raise ValueError("glu: channel axis (dim=1) only")
    return _GluFn.apply(x)


def lstm(module, x):
    """module: nn.LSTM parameter container; x: (T, B, C)."""
    _interim("lstm")
    return module(x)[0]


def linear(x, weight, bias):
    """x: (..., Cin) -> (..., Cout) as a 1x1 gather-GEMM over the flattened rows."""
    shp = x.shape
    x2 = x.reshape(1, -1, shp[-1]).transpose(1, 2)              # (1, Cin, rows) strided view
    y = ops.conv1d(x2, weight.unsqueeze(-1), bias)              # (1, Cout, rows)
    return y.transpose(1, 2).reshape(*shp[:-1], weight.shape[0])


def softmax(x, dim):
    _interim("softmax")
    return torch.softmax(x, dim=dim)


def einsum(eq, *xs):
    _interim("einsum:" + eq)
    return torch.einsum(eq, *xs)

"""Fused DConv depth-layers on channels-last bf16 samples (csrc/cl_dconv.hip): torchaudio HDemucs `_DConv` of the norm-free
frequency encoder layers (reference call site remfx/models.py:319), bf16 arithmetic mode.

Host side: the index tables that put W1 / W2 into the MFMA fragments the kernels read (packed by rfx_cl_pack once per weight
version), the launch descriptors, and the autograd node.  The two weight-gradient GEMMs of a layer run on csrc/cl_wgrad.hip from
the dz / dh tensors the backward kernel leaves (deterministic, GradSink-aware through clchain._wgrad).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, clast, clchain, ops
from ._lib import ClDconvDesc, check

T = 256
import os as _os
GRID = 256            # persistent workgroups of the backward kernels: one per CU (a workgroup holds a sample's images in LDS)
GRID_FWD = int(_os.environ.get("RFX_CLD_GRID_FWD", "256"))   # forward kernels: 128 registers, two workgroups fit a CU


class _Idx:
    """A gather index for rfx_cl_pack (the interface clast.pack expects of a form)."""

    def __init__(self, idx):
        self.idx = np.ascontiguousarray(idx.reshape(-1).astype(np.int32))
        self._dev = {}


_TABLES = {}


def tables(Cc, H):
    key = (Cc, H)
    t = _TABLES.get(key)
    if t is not None:
        return t
    HP, KC, NTV = -(-H // 16) * 16, Cc // 16, -(-Cc // 32)
    KH, NT2, KZ = HP // 16, 2 * NTV, 2 * Cc // 16
    lane = np.arange(64, dtype=np.int64)[:, None]
    e = np.arange(8, dtype=np.int64)[None, :]
    l31, kh = lane & 31, lane >> 5

    def grid(*dims):
        return np.meshgrid(*[np.arange(d, dtype=np.int64) for d in dims], indexing="ij", sparse=True)

    # GEMM1 A: [tap][ks] fragments, row h = lane & 31, k = channel 16 ks + 8 khalf + e
    tp, ks = grid(3, KC)
    tp, ks = tp[..., None, None], ks[..., None, None]
    c = 16 * ks + 8 * kh + e
    w1p = np.where(l31 < H, (np.minimum(l31, H - 1) * Cc + c) * 3 + tp, -1)
    # GEMM2^T B (forward and backward recompute): [ks][tile] fragments, column = value / gate channel, k in C/D register order
    ks, tl = grid(KH, NT2)
    ks, tl = ks[..., None, None], tl[..., None, None]
    hh = (e & 3) + 8 * (2 * ks + (e >> 2)) + 4 * kh
    cc = 32 * (tl % NTV) + l31
    m = np.where(tl < NTV, cc, Cc + cc)
    w2p = np.where((cc < Cc) & (hh < H), np.minimum(m, 2 * Cc - 1) * H + np.minimum(hh, H - 1), -1)
    # da^T B: [kz] fragments, column = hidden channel, k = the 2 C channels in natural order
    kz = np.arange(KZ, dtype=np.int64)[:, None, None]
    m = 16 * kz + 8 * kh + e
    w2dp = np.where(l31 < H, m * H + np.minimum(l31, H - 1), -1) + 0 * kz
    # dx^T B: [tap][ks][tile] fragments, column = channel, k = hidden channel 16 ks + 8 khalf + e
    tp, ks, tv = grid(3, KH, NTV)
    tp, ks, tv = tp[..., None, None], ks[..., None, None], tv[..., None, None]
    c = 32 * tv + l31
    h = 16 * ks + 8 * kh + e
    w1dp = np.where((c < Cc) & (h < H), (np.minimum(h, H - 1) * Cc + np.minimum(c, Cc - 1)) * 3 + tp, -1)
    t = {"w1p": _Idx(np.broadcast_to(w1p, (3, KC, 64, 8))), "w2p": _Idx(np.broadcast_to(w2p, (KH, NT2, 64, 8))),
         "w2dp": _Idx(np.broadcast_to(w2dp, (KZ, 64, 8))), "w1dp": _Idx(np.broadcast_to(w1dp, (3, KH, NTV, 64, 8))), "HP": HP}
    _TABLES[key] = t
    return t


def supported(Cc, H, backward, positions=T):
    return bool(_lib.lib().rfx_cl_dconv_ok(int(Cc), int(H), int(positions), int(backward)))


def _dx_form(Cc, H, dil):
    """dx = gy + sum_t dh(pos - (t - 1) dil) W1_t as a channels-last convolution over dh (HP stored channels): rows = C."""
    key = ("dconv_dx", Cc, H, dil)
    f = clchain._FORMS.get(key)
    if f is None:
        HP = -(-H // 16) * 16

        def widx(m, r, t, ch):
            return np.where(ch < H, (np.minimum(ch, H - 1) * Cc + m) * 3 + (2 - t), -1)
        f = clast.ConvForm(Cc, HP, 1, 3, 0, 0, -dil, dil, 1, widx, KS=1)
        clchain._FORMS[key] = f
    return f


def _wforms(Cc, H, dil):
    key = ("dconv", Cc, H, dil)
    f = clchain._FORMS.get(key)
    if f is None:
        f2 = clast.WgradForm(2 * Cc, H, 1, 1, 1, 0, 0, 0, lambda m, r, t, c: m * H + c, 2 * Cc * H)
        f1 = clast.WgradForm(H, Cc, 1, 3, 1, 0, -dil, dil, lambda m, r, t, c: (m * Cc + c) * 3 + t, H * Cc * 3)
        f = (f1, f2)
        clchain._FORMS[key] = f
    return f


def _p(t):
    return t.data_ptr() if t is not None else None


def _desc(Cc, H, dil, eps, S, b1, g1w, g1b, b2, g2w, g2b, scale):
    d = ClDconvDesc()
    d.S, d.C, d.H, d.dil, d.grid, d.eps = S, Cc, H, dil, GRID, eps
    d.b1, d.g1w, d.g1b, d.b2, d.g2w, d.g2b, d.scale = _p(b1), _p(g1w), _p(g1b), _p(b2), _p(g2w), _p(g2b), _p(scale)
    return d


class ClDconvLayerFn(torch.autograd.Function):
    """x (Bn, A, 256, C) channels-last bf16 -> y of the same shape; parameters as the upstream modules hold them."""

    @staticmethod
    def forward(ctx, x, w1, b1, g1w, g1b, w2, b2, g2w, g2b, scale, dil, eps):
        Bn, A, Tt, Cc = x.shape
        H = w1.shape[0]
        if Tt % T or not x.is_contiguous() or x.dtype != torch.bfloat16:
            raise ValueError("ClDconvLayerFn: dense (N, A, 256 k, C) bf16 input")
        tb = tables(Cc, H)
        TPS = Tt // T                                   # tiles per sample: 1 = frequency branch, > 1 = a time-branch clip
        HP, S, dev = tb["HP"], Bn * A * TPS, x.device
        train = any(ctx.needs_input_grad)
        y = torch.empty_like(x)
        d = _desc(Cc, H, dil, eps, S, b1, g1w, g1b, b2, g2w, g2b, scale)
        d.TPS = TPS
        d.grid = GRID_FWD
        d.x, d.y = x.data_ptr(), y.data_ptr()
        d.w1p = clchain.packed(tb["w1p"], w1).data_ptr()
        d.w2p = clchain.packed(tb["w2p"], w2).data_ptr()
        a = hpre = stats = part = None
        if train or TPS > 1:
            stats = torch.empty((Bn * A, 4), device=dev, dtype=torch.float32)
            d.stats = stats.data_ptr()
        if TPS > 1:
            part = torch.empty((S, 2), device=dev, dtype=torch.float32)
            d.partial = part.data_ptr()
        if train:
            a = torch.empty((Bn, A, Tt, HP), device=dev, dtype=torch.bfloat16)
            hpre = torch.empty_like(a)
            d.a, d.hpre = a.data_ptr(), hpre.data_ptr()
        check(_lib.lib().rfx_cl_dconv_fwd(C.byref(d), C.c_void_p(ops.raw_stream())), "rfx_cl_dconv_fwd")
        if train:
            ctx.save_for_backward(x, a, hpre, stats, w1, b1, g1w, g1b, w2, b2, g2w, g2b, scale)
            ctx.cfg = (dil, eps)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, a, hpre, stats, w1, b1, g1w, g1b, w2, b2, g2w, g2b, scale = ctx.saved_tensors
        dil, eps = ctx.cfg
        Bn, A, Tt, Cc = x.shape
        H = w1.shape[0]
        tb = tables(Cc, H)
        TPS = Tt // T
        HP, S, dev = tb["HP"], Bn * A * TPS, x.device
        passes = TPS > 1 or Cc != 48                    # several tiles per sample, or images too large for the one-pass kernel's LDS
        if not gy.is_contiguous():
            gy = gy.contiguous()
        dx = torch.empty_like(x)
        dz = torch.empty((Bn, A, Tt, 2 * Cc), device=dev, dtype=torch.bfloat16)
        dh = torch.empty((Bn, A, Tt, HP), device=dev, dtype=torch.bfloat16)
        npg = 5 * Cc + 2 * H
        partial = torch.empty((min(GRID, S), npg), device=dev, dtype=torch.float32)
        # the five small gradients go straight into the parameters' slices of the flat gradient buffer when a GradSink is armed
        # (autograd would add each returned tensor into .grad with a launch of its own)
        sink = ops.SINK
        tgt = [sink.lookup(p) for p in (scale, g2w, g2b, g1w, g1b)] if sink is not None else [None]
        direct = all(t is not None for t in tgt)
        pg = None if direct else torch.empty(npg, device=dev, dtype=torch.float32)
        d = _desc(Cc, H, dil, eps, S, b1, g1w, g1b, b2, g2w, g2b, scale)
        d.TPS = TPS
        d.gy, d.y, d.a, d.hpre, d.stats = gy.data_ptr(), dx.data_ptr(), a.data_ptr(), hpre.data_ptr(), stats.data_ptr()
        d.dz, d.dh, d.partial = dz.data_ptr(), dh.data_ptr(), partial.data_ptr()
        d.w2p = clchain.packed(tb["w2p"], w2).data_ptr()
        d.w2dp = clchain.packed(tb["w2dp"], w2).data_ptr()
        if passes:
            tsum = torch.empty((S, 2), device=dev, dtype=torch.float32)
            sums = torch.empty((Bn * A, 4), device=dev, dtype=torch.float32)
            d.tsum, d.sums = tsum.data_ptr(), sums.data_ptr()
        else:
            d.w1dp = clchain.packed(tb["w1dp"], w1).data_ptr()
        if direct:
            for q, t in enumerate(tgt):
                d.pg_dst[q] = t[1].data_ptr()
        check(_lib.lib().rfx_cl_dconv_bwd(C.byref(d), C.c_void_p(pg.data_ptr() if pg is not None else None), C.c_void_p(ops.raw_stream())),
              "rfx_cl_dconv_bwd")
        if direct:
            for t in tgt:
                sink.wrote(t[0])
        if passes:                                      # dx = gy + the transposed 3-tap convolution of dh (taps cross tile edges)
            fx = _dx_form(Cc, H, dil)
            clast.conv(fx, clchain.packed(fx, w1), dh, Bn, A, Tt, A, "store", out0=dx, res=gy)
        f1, f2 = _wforms(Cc, H, dil)
        dw2, db2 = clchain._wgrad(f2, dz, a, Bn, A, A, Tt, w2, b2)
        dw1, db1 = clchain._wgrad(f1, dh, x, Bn, A, A, Tt, w1, b1)
        if direct:
            dscale = dg2w = dg2b = dg1w = dg1b = None
        else:
            dscale, dg2w, dg2b = pg[:Cc], pg[Cc:3 * Cc], pg[3 * Cc:5 * Cc]
            dg1w, dg1b = pg[5 * Cc:5 * Cc + H], pg[5 * Cc + H:]
        return dx, dw1, db1, dg1w, dg1b, dw2, db2, dg2w, dg2b, dscale, None, None


def dconv_layer(x, conv1, gn1, conv2, gn2, scale, dil):
    return ClDconvLayerFn.apply(x, conv1.weight, conv1.bias, gn1.weight, gn1.bias, conv2.weight, conv2.bias, gn2.weight, gn2.bias,
                                scale, int(dil), float(gn1.eps))

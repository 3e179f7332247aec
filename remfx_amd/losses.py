"""Losses and metrics of the removal path on HIP kernels.

Mirrors the auraloss objects the reference instantiates (remfx/models.py:7-8,
35-44, 171-176, 289-291, 312-314, 351-353, 374-376):
  MultiResolutionSTFTLoss(fft 1024/2048/512, hop 120/240/50, win 600/1200/240):
      mean over resolutions of [ ||Y|-|X||_F / ||Y||_F  +  mean |log|X| - log|Y|| ]
  SISDRLoss(zero_mean=True, eps=1e-8)
  nn.L1Loss
`n_bins` / `sample_rate` kwargs are accepted and inert, as upstream with scale=None
(SURVEY App. B Q4).
"""
import ctypes as C
import os

import torch
import torch.nn as nn

from . import _lib, stft
from ._lib import check
from .ops import _ptr, _req, _stream, zeros

FFT_SIZES = (1024, 2048, 512)
HOP_SIZES = (120, 240, 50)
WIN_LENGTHS = (600, 1200, 240)


FUSED_GRAD = os.environ.get("RFX_LOSS_FUSED_GRAD", "1") != "0"    # A/B: 0 = rfx_stft_loss_grad_m + rfx_fft_synthesis as two launches

_MEMO = None


class stft_memo:
    """Scope in which the spectra the MRSTFT loss / metric take are computed once per (signal, resolution).

    RemFX.common_step evaluates MRSTFT three times on one batch -- loss(output, target), metric(output, target) and
    metric(input, target) (models.py:220-255) -- i.e. the same STFT of the target three times and of the output twice.
    Inside ``with stft_memo():`` a repeated request for the spectrum of the same storage at the same version returns
    the tensor computed the first time.  The memo dies with the scope, so nothing is carried from one step to the next.
    """

    def __enter__(self):
        global _MEMO
        self._prev, _MEMO = _MEMO, {}
        return self

    def __exit__(self, *exc):
        global _MEMO
        _MEMO = self._prev
        return False


# The loss kernels are elementwise over (prediction, target) spectra plus row sums: layout-free.  Frame-major spectra
# ([R][frames][bins][2]) let the FFT kernels store / load whole lines instead of pieces of a [bin][frame] transpose.
_SPEC_MODE = stft.MODES["complex_fm"]


def _spectrum(sig, n_fft, hop, win, w):
    if _MEMO is None:
        return stft.stft_raw(sig, n_fft, hop, win, w, _SPEC_MODE)
    key = (sig.data_ptr(), sig._version, tuple(sig.shape), tuple(sig.stride()), n_fft, hop, win)
    hit = _MEMO.get(key)
    if hit is None:
        hit = (sig, stft.stft_raw(sig, n_fft, hop, win, w, _SPEC_MODE))     # holding sig keeps its storage from being reused
        _MEMO[key] = hit
        _MEMO[("spec", hit[1].data_ptr())] = True
    return hit[1]


def _sig_key(sig):
    return (sig.data_ptr(), sig._version, tuple(sig.shape), tuple(sig.stride()))


def _pair_sums(x2, y2, n_fft, hop, win, w, eps, store):
    """One resolution in one launch (rfx_stft_pair_loss): row sums [R, 3]; with store=True also the prediction's spectrum
    (frame-major complex) and the clamped target magnitudes, which the backward reads.  Memoised like _spectrum."""
    key = ("pair", _sig_key(x2), _sig_key(y2), n_fft, hop, win, eps)
    hit = _MEMO.get(key) if _MEMO is not None else None
    if hit is not None and (hit[1] is not None or not store):
        return hit[0], hit[1], hit[2]
    R, L = x2.shape
    if n_fft // 2 >= L:
        raise ValueError("reflect padding needs n_fft/2 < signal length")
    frames, bins = 1 + L // hop, n_fft // 2 + 1
    sums = torch.empty((R, 3), device=x2.device, dtype=torch.float32)       # written, not accumulated: no zero fill
    X = torch.empty((R, frames, bins, 2), device=x2.device, dtype=torch.float32) if store else None
    ym = torch.empty((R, frames, bins), device=x2.device, dtype=torch.float32) if store else None
    d = stft._desc(R, L, n_fft, hop, win, bins, 0, frames, _SPEC_MODE)
    ws = torch.empty(int(_lib.lib().rfx_stft_pair_loss_ws(C.byref(d))), device=x2.device, dtype=torch.float64)   # one slot per workgroup
    check(_lib.lib().rfx_stft_pair_loss(C.byref(d), _ptr(x2), _ptr(y2), _ptr(w), eps, _ptr(ws), _ptr(sums), _ptr(X), _ptr(ym),
                                        _stream()), "rfx_stft_pair_loss")
    if _MEMO is not None:
        _MEMO[key] = (sums, X, ym, x2, y2)      # holding the signals keeps their storage from being reused under the key
    return sums, X, ym


class _MRSTFTFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, fft_sizes, hops, wins, eps, per_example_sc):
        _req(x, "input"); _req(y, "target")
        L = x.shape[-1]
        x2, y2 = x.reshape(-1, L).contiguous(), y.reshape(-1, L).contiguous()
        R = x2.shape[0]
        saved, total = [], None
        for n_fft, hop, win in zip(fft_sizes, hops, wins):
            w = stft.hann(win, x.device)
            if n_fft in (512, 1024, 2048):
                # both spectra of a frame from one complex FFT, sums in its epilogue: no spectrum is written unless the backward
                # needs it, and then only the prediction's + the target's magnitudes
                sums, X, Y = _pair_sums(x2, y2, n_fft, hop, win, w, eps, ctx.needs_input_grad[0])
                n = (1 + L // hop) * (n_fft // 2 + 1)
                paired = True
            else:
                X = _spectrum(x2, n_fft, hop, win, w)
                Y = _spectrum(y2, n_fft, hop, win, w)
                n = X.shape[1] * X.shape[2]
                paired = False
                skey = ("sums", X.data_ptr(), Y.data_ptr(), eps)
                sums = _MEMO.get(skey) if _MEMO is not None else None   # metric(output, target) repeats the loss's row sums
                if sums is None:
                    sums = torch.empty((R, 3), device=x.device, dtype=torch.float32)
                    ws = torch.empty(3 * R * 64, device=x.device, dtype=torch.float64)        # RFX_STFT_REDUCE_SLOTS per row
                    check(_lib.lib().rfx_stft_loss_reduce(_ptr(X), _ptr(Y), R, n, eps, _ptr(ws), _ptr(sums), _stream()),
                          "rfx_stft_loss_reduce")
                    if _MEMO is not None and ("spec", X.data_ptr()) in _MEMO and ("spec", Y.data_ptr()) in _MEMO:
                        _MEMO[skey] = sums                               # both spectra are held by the memo: pointers stay valid
            saved.append((paired, X, Y, sums, n, n_fft, hop, win))
        # sc + lm of every resolution and their mean: one launch on the row sums (was ~24 one-element torch launches per evaluation)
        nres = len(saved)
        total = None
        for c0 in range(0, nres, 8):               # rfx_mrstft_combine takes up to 8 resolutions per launch (auraloss's default has 3)
            part, m = saved[c0:c0 + 8], len(saved[c0:c0 + 8])
            t = torch.empty((), device=x.device, dtype=torch.float32)
            sp = (C.c_void_p * m)(*[_ptr(e[3]) for e in part])
            nn_ = (C.c_int64 * m)(*[int(e[4]) for e in part])
            check(_lib.lib().rfx_mrstft_combine(sp, nn_, m, R, 1 if per_example_sc else 0, _ptr(t), _stream()),
                  "rfx_mrstft_combine")
            total = t if nres <= 8 else (t * (m / nres) if total is None else total + t * (m / nres))
        ctx.saved = saved
        ctx.meta = (x.shape, R, L, eps, per_example_sc, nres)
        return total

    @staticmethod
    def backward(ctx, g):
        shape, R, L, eps, per_example_sc, nres = ctx.meta
        gx = torch.empty((R, L), device=g.device, dtype=torch.float32)    # the first resolution writes it, the others add (accum)
        gval = 1.0               # the scalar upstream gradient stays on the device: the kernels multiply their weights by *gup
        gup = g.detach().reshape(1).float().contiguous()
        for ires, (paired, X, Y, sums, n, n_fft, hop, win) in enumerate(ctx.saved):
            if not per_example_sc:      # whole-batch Frobenius norm: same A, B for every row
                sums = sums.clone()
                sums[:, 0] = sums[:, 0].sum()
                sums[:, 1] = sums[:, 1].sum()
                w_sc = gval / nres
            else:
                w_sc = gval / (nres * R)
            w_lm = gval / (nres * R * n)
            w = stft.hann(win, g.device)
            d = stft._desc(R, L, n_fft, hop, win, X.shape[2], 0, X.shape[1], _SPEC_MODE, in_mode=0, herm=0, scale=1.0,
                           accum=1 if ires else 0)
            if paired and FUSED_GRAD:   # Y = the clamped target magnitudes; the gradient spectrum is formed inside the synthesis
                check(_lib.lib().rfx_fft_synthesis_lossgrad(C.byref(d), _ptr(X), _ptr(Y), _ptr(sums), w_sc, w_lm, eps, _ptr(gup),
                                                            _ptr(w), _ptr(stft.syn_ws(d, g.device)), _ptr(gx), _stream()),
                      "rfx_fft_synthesis_lossgrad")
                continue
            G = torch.empty_like(X)
            if paired:
                check(_lib.lib().rfx_stft_loss_grad_m(_ptr(X), _ptr(Y), R, n, eps, _ptr(sums), w_sc, w_lm, _ptr(gup), _ptr(G),
                                                      _stream()), "rfx_stft_loss_grad_m")
            else:
                check(_lib.lib().rfx_stft_loss_grad(_ptr(X), _ptr(Y), R, n, eps, _ptr(sums), w_sc, w_lm, _ptr(gup), _ptr(G),
                                                    _stream()), "rfx_stft_loss_grad")
            check(_lib.lib().rfx_fft_synthesis(C.byref(d), _ptr(G), _ptr(w), None, _ptr(stft.syn_ws(d, g.device)), _ptr(gx), _stream()),
                  "rfx_fft_synthesis")
        ctx.saved = None
        return gx.view(shape), None, None, None, None, None, None


class MultiResolutionSTFTLoss(nn.Module):
    def __init__(self, fft_sizes=FFT_SIZES, hop_sizes=HOP_SIZES, win_lengths=WIN_LENGTHS, n_bins=None,
                 sample_rate=None, eps=1e-8, per_example_sc=True, **kwargs):
        super().__init__()
        self.fft_sizes, self.hop_sizes, self.win_lengths = tuple(fft_sizes), tuple(hop_sizes), tuple(win_lengths)
        self.eps, self.per_example_sc = eps, per_example_sc

    def forward(self, input, target):
        return _MRSTFTFn.apply(input, target, self.fft_sizes, self.hop_sizes, self.win_lengths, self.eps,
                               self.per_example_sc)


class _L1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        _req(a, "input"); _req(b, "target")
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty((), device=a.device, dtype=torch.float32)
        ws = torch.empty(512, device=a.device, dtype=torch.float64)                 # RFX_L1_SLOTS per-workgroup partials
        check(_lib.lib().rfx_l1_sum(_ptr(a), _ptr(b), a.numel(), _ptr(ws), 1.0 / a.numel(), _ptr(out), _stream()), "rfx_l1_sum")
        ctx.save_for_backward(a, b)
        return out

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        ga = torch.empty_like(a)
        gup = g.detach().reshape(1).float().contiguous()          # upstream scalar gradient, multiplied in on the device (no host sync)
        check(_lib.lib().rfx_l1_grad(_ptr(a), _ptr(b), a.numel(), 1.0 / a.numel(), _ptr(gup), _ptr(ga), _stream()),
              "rfx_l1_grad")
        return ga, None


class L1Loss(nn.Module):
    def forward(self, input, target):
        return _L1Fn.apply(input, target)


class SISDRLoss(nn.Module):
    """-SI-SDR (auraloss.time.SISDRLoss, zero_mean=True, eps=1e-8, reduction='mean');
    metric only (no gradient), models.py:227-255."""

    def __init__(self, zero_mean=True, eps=1e-8):
        super().__init__()
        self.zero_mean, self.eps = zero_mean, eps

    @torch.no_grad()
    def forward(self, input, target):
        _req(input, "input"); _req(target, "target")
        L = input.shape[-1]
        x, t = input.reshape(-1, L), target.reshape(-1, L)
        if x.stride(-1) != 1:
            x = x.contiguous()
        if t.stride(-1) != 1:
            t = t.contiguous()
        R = x.shape[0]
        s = torch.empty((R, 5), device=x.device, dtype=torch.float64)
        ws = torch.empty(5 * R * 128, device=x.device, dtype=torch.float64)          # RFX_SISDR_SLOTS per-workgroup partials per row
        check(_lib.lib().rfx_sisdr_sums(_ptr(x), _ptr(t), R, L, x.stride(0), t.stride(0), _ptr(ws), _ptr(s), _stream()),
              "rfx_sisdr_sums")
        out = torch.empty((), device=x.device, dtype=torch.float32)      # the scalar tail in one launch (fp64 inside)
        check(_lib.lib().rfx_sisdr_finish(_ptr(s), R, L, 1 if self.zero_mean else 0, float(self.eps), _ptr(out), _stream()),
              "rfx_sisdr_finish")
        return out

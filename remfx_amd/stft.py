"""STFT / iSTFT on the framed-FFT HIP kernels (csrc/fft.hip), with autograd.

Semantics follow torch.stft(center=True, pad_mode="reflect", onesided=True) and
torch.istft(center=True), the ops behind utils.py:148-154, HDemucs _spec/_ispec,
auraloss STFTLoss, Open-Unmix Separator and MelSpectrogram in the reference.
"""
import ctypes as C
import math

import torch

from . import _lib
from ._lib import StftDesc, check
from .ops import _ptr, _req, _stream, zeros

MODES = {"complex": 0, "cac": 1, "mag": 2, "pow": 3, "magpow": 4, "complex_fm": 5}
_WINDOWS = {}


def hann(win, device):
    k = (win, str(device))
    w = _WINDOWS.get(k)
    if w is None:
        w = torch.hann_window(win, dtype=torch.float32).to(device)   # periodic hann, as torch.hann_window
        _WINDOWS[k] = w
    return w


def _desc(R, T, n_fft, hop, win, bins, frame0, frames_out, mode, extra_pad=(0, 0), in_mode=0,
          in_offset=0, herm=0, scale=1.0, eps=0.0, alpha=1.0, accum=0):
    if n_fft not in (512, 1024, 2048, 4096):
        raise ValueError(f"n_fft={n_fft}: the HIP FFT kernels cover 512/1024/2048/4096")
    d = StftDesc()
    d.R, d.T, d.n_fft, d.hop, d.win = R, T, n_fft, hop, win
    d.bins, d.frame0, d.frames_out, d.mode = bins, frame0, frames_out, mode
    d.extra_pad_l, d.extra_pad_r = extra_pad
    d.in_mode, d.in_offset, d.herm = in_mode, in_offset, herm
    d.scale, d.eps, d.alpha, d.accum = scale, eps, alpha, accum
    return d


def syn_ws(d, device):
    """Scratch of one rfx_fft_synthesis launch (uninitialised): carried sums + padded edge zones (include/remfx_hip.h)."""
    return torch.empty(int(_lib.lib().rfx_fft_synthesis_ws(C.byref(d))), device=device, dtype=torch.float32)


def _out_shape(R, bins, frames, mode):
    if mode == 0:
        return (R, bins, frames, 2)
    if mode == 1:
        return (R, 2, bins, frames)
    if mode == 5:                      # frame-major complex (layout-free consumers: the MR-STFT loss kernels)
        return (R, frames, bins, 2)
    return (R, bins, frames)


def stft_raw(x, n_fft, hop, win, window, mode, normalized=False, eps=0.0, alpha=1.0, bins=None,
             frame0=0, frames_out=None, extra_pad=(0, 0)):
    """x: (R, T) fp32 contiguous -> spectrum tensor (no autograd)."""
    _req(x, "x")
    R, T = x.shape
    Tp = T + extra_pad[0] + extra_pad[1]
    frames = 1 + Tp // hop
    bins = n_fft // 2 + 1 if bins is None else bins
    frames_out = frames - frame0 if frames_out is None else frames_out
    if n_fft // 2 >= Tp:
        raise ValueError("reflect padding needs n_fft/2 < signal length")
    out = torch.empty(_out_shape(R, bins, frames_out, mode), device=x.device, dtype=torch.float32)
    d = _desc(R, T, n_fft, hop, win, bins, frame0, frames_out, mode, extra_pad,
              scale=(1.0 / math.sqrt(n_fft)) if normalized else 1.0, eps=eps, alpha=alpha)
    check(_lib.lib().rfx_fft_analysis(C.byref(d), _ptr(x), _ptr(window), None, _ptr(out), _stream()),
          "rfx_fft_analysis")
    return out


class STFTFn(torch.autograd.Function):
    """Complex / complex-as-channels STFT with the adjoint (synthesis kernel) as backward."""

    @staticmethod
    def forward(ctx, x, window, n_fft, hop, win, mode, normalized, bins, frame0, frames_out, extra_pad):
        x = x.contiguous()
        ctx.cfg = (x.shape, n_fft, hop, win, mode, normalized, bins, frame0, frames_out, extra_pad)
        ctx.save_for_backward(window)
        return stft_raw(x, n_fft, hop, win, window, mode, normalized, bins=bins, frame0=frame0,
                        frames_out=frames_out, extra_pad=extra_pad)

    @staticmethod
    def backward(ctx, g):
        (window,) = ctx.saved_tensors
        (R, T), n_fft, hop, win, mode, normalized, bins, frame0, frames_out, extra_pad = ctx.cfg
        g = g.contiguous()
        fo = g.shape[2] if mode == 0 else g.shape[1] if mode == 5 else g.shape[3]
        nb = g.shape[1] if mode == 0 else g.shape[2]
        gx = torch.empty((R, T), device=g.device, dtype=torch.float32)      # rfx_fft_synthesis writes every element (accum = 0)
        d = _desc(R, T, n_fft, hop, win, nb, frame0, fo, mode, extra_pad, in_mode=0, herm=0,
                  scale=(1.0 / math.sqrt(n_fft)) if normalized else 1.0)
        check(_lib.lib().rfx_fft_synthesis(C.byref(d), _ptr(g), _ptr(window), None, _ptr(syn_ws(d, g.device)), _ptr(gx), _stream()),
              "rfx_fft_synthesis")
        return (gx,) + (None,) * 10


def stft(x, n_fft, hop, win=None, window=None, mode="complex", normalized=False, eps=0.0, alpha=1.0,
         bins=None, frame0=0, frames_out=None, extra_pad=(0, 0)):
    """(R, T) -> spectrum.  'complex' (R,bins,frames,2) and 'cac' (R,2,bins,frames) are
    differentiable; 'mag' / 'pow' / 'magpow' are forward-only fused epilogues."""
    win = n_fft if win is None else win
    window = hann(win, x.device) if window is None else window
    m = MODES[mode]
    if m <= 1 or m == 5:
        return STFTFn.apply(x, window, n_fft, hop, win, m, normalized, bins, frame0, frames_out, extra_pad)
    return stft_raw(x.contiguous(), n_fft, hop, win, window, m, normalized, eps, alpha, bins, frame0,
                    frames_out, extra_pad)


_ENV = {}


def _inv_envelope(window, n_fft, hop, win, frames, device):
    """1 / sum_f w^2[p - f*hop] over the padded sample axis (torch.istft's window envelope)."""
    key = (n_fft, hop, win, frames, str(device), window.data_ptr())
    e = _ENV.get(key)
    if e is None:
        w = torch.zeros(n_fft, dtype=torch.float64)
        off = (n_fft - win) // 2
        w[off:off + win] = window.detach().double().cpu()
        total = (frames - 1) * hop + n_fft
        env = torch.zeros(total, dtype=torch.float64)
        idx = (torch.arange(frames)[:, None] * hop + torch.arange(n_fft)[None, :]).reshape(-1)
        env.index_add_(0, idx, (w * w).repeat(frames))
        inv = torch.where(env > 1e-11, 1.0 / env, torch.zeros_like(env))
        e = inv.float().to(device)
        _ENV[key] = e
    return e


class ISTFTFn(torch.autograd.Function):
    """torch.istft(center=True, length=...) followed by an optional crop, on a spectrum that
    stores frames [frame0, frame0+frames_in) of `frames` (the rest are zero) and `bins`
    bins (missing Nyquist = zero): HDemucs _ispec, Separator."""

    @staticmethod
    def forward(ctx, spec, window, n_fft, hop, win, mode, normalized, frames, frame0, crop, length):
        spec = spec.contiguous()
        R = spec.shape[0]
        nb = spec.shape[1] if mode == 0 else spec.shape[2]                 # complex (R, bins, frames, 2) | cac (R, 2, bins, frames) | fm (R, frames, bins, 2)
        fi = spec.shape[2] if mode == 0 else spec.shape[1] if mode == 5 else spec.shape[3]
        inv_env = _inv_envelope(window, n_fft, hop, win, frames, spec.device)
        scale = (math.sqrt(n_fft) if normalized else 1.0) / n_fft
        out = torch.empty((R, length), device=spec.device, dtype=torch.float32)   # every sample is stored exactly once (overlap-add by ownership)
        d = _desc(R, length, n_fft, hop, win, nb, frame0, fi, mode, in_mode=1, in_offset=n_fft // 2 + crop,
                  herm=1, scale=scale)
        check(_lib.lib().rfx_fft_synthesis(C.byref(d), _ptr(spec), _ptr(window), _ptr(inv_env), _ptr(syn_ws(d, spec.device)), _ptr(out),
                                           _stream()), "rfx_fft_synthesis")
        ctx.save_for_backward(window, inv_env)
        ctx.cfg = (spec.shape, n_fft, hop, win, mode, frames, frame0, crop, length, scale, nb, fi)
        return out

    @staticmethod
    def backward(ctx, g):
        window, inv_env = ctx.saved_tensors
        shape, n_fft, hop, win, mode, frames, frame0, crop, length, scale, nb, fi = ctx.cfg
        g = g.contiguous()
        R = shape[0]
        gs = torch.empty(shape, device=g.device, dtype=torch.float32)
        d = _desc(R, length, n_fft, hop, win, nb, frame0, fi, mode, in_mode=1, in_offset=n_fft // 2 + crop,
                  herm=1, scale=scale)
        check(_lib.lib().rfx_fft_analysis(C.byref(d), _ptr(g), _ptr(window), _ptr(inv_env), _ptr(gs),
                                          _stream()), "rfx_fft_analysis")
        return (gs,) + (None,) * 10


def istft(spec, n_fft, hop, win=None, window=None, mode="complex", normalized=False, frames=None,
          frame0=0, crop=0, length=None):
    win = n_fft if win is None else win
    window = hann(win, spec.device) if window is None else window
    m = MODES[mode]
    if m not in (0, 1, 5):
        raise ValueError("istft: mode 'complex', 'cac' or 'complex_fm'")
    fi = spec.shape[2] if m == 0 else spec.shape[1] if m == 5 else spec.shape[3]
    frames = fi + frame0 if frames is None else frames
    if length is None:
        length = (frames - 1) * hop
    return ISTFTFn.apply(spec, window, n_fft, hop, win, m, normalized, frames, frame0, crop, length)

"""DCUNet "Large-DCUNet-20" (asteroid.models.DCUNet as RemFX configures it: reference
remfx/models.py:347-367, cfg/model/dcunet.yaml:11-16) on the HIP kernels.

Same constructor keywords and state_dict names as upstream (encoder.filterbank._filters,
decoder.filterbank._filters, masker.encoders.{i}.conv.{re_module,im_module}.weight,
masker.encoders.{i}.norm.{Wrr,Wri,Wii,Br,Bi,RMr,RMi,RVrr,RVri,RVii,num_batches_tracked},
masker.decoders.{i}.deconv.*, masker.output_layer.0.*).

MI355X design:
  * complex tensors are real (N, 2C, H, W) tensors [real channels | imaginary channels]; every
    complex (transposed) convolution is ONE gather-GEMM with the block weight [[Wr,-Wi],[Wi,Wr]]
    (M = 2*Cout, K = 2*Cin*kh*kw) -- the same MFMA kernel as every other convolution;
  * ComplexBatchNorm + LeakyReLU: fp64 moment reduction -> 2x2 whitening/affine algebra on C-length
    vectors (host) -> one fused apply kernel; backward mirrors it;
  * U-Net skips are zero-copy: encoder / decoder activations are written straight into channel slices of
    the pre-allocated concatenation buffer (the apply kernel takes an output stride / offset);
  * the STFT filterbank conv writes into the time-padded (B, 2, 513, 1025) buffer the masker reads.
"""
import ctypes as C

import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import check
from .ops import _ptr, _stream

ENCODERS = ((1, 45, (7, 1), (1, 1)), (45, 45, (1, 7), (1, 1)), (45, 90, (7, 5), (2, 2)), (90, 90, (7, 5), (2, 1)),
            (90, 90, (5, 3), (2, 2)), (90, 90, (5, 3), (2, 1)), (90, 90, (5, 3), (2, 2)), (90, 90, (5, 3), (2, 1)),
            (90, 90, (5, 3), (2, 2)), (90, 128, (5, 3), (2, 1)))
DECODERS = ((128, 90, (5, 3), (2, 1)), (180, 90, (5, 3), (2, 2)), (180, 90, (5, 3), (2, 1)), (180, 90, (5, 3), (2, 2)),
            (180, 90, (5, 3), (2, 1)), (180, 90, (5, 3), (2, 2)), (180, 90, (7, 5), (2, 1)), (180, 90, (7, 5), (2, 2)),
            (135, 90, (1, 7), (1, 1)), (135, 1, (7, 1), (1, 1)))
LEAKY = 0.01


def stft_filters(n_filters=1024, kernel_size=512, stride=256, form="slice"):
    """asteroid-filterbanks STFTFB buffer: (n_filters + 2, 1, kernel_size)."""
    cutoff = n_filters // 2 + 1
    window = np.hanning(kernel_size + 1)[:-1] ** 0.5
    filt = np.fft.fft(np.eye(n_filters))
    filt /= 0.5 * np.sqrt(kernel_size * n_filters / stride)
    lpad = (n_filters - kernel_size) // 2
    idx = list(range(lpad, lpad + kernel_size))
    filt = np.vstack([np.real(filt[:cutoff, idx]), np.imag(filt[:cutoff, idx])])
    filt[0, :] /= np.sqrt(2)
    filt[n_filters // 2, :] /= np.sqrt(2)
    filt = filt * window
    if form == "zero_pad":       # window zero-padded to n_filters: (n_filters + 2, 1, n_filters) buffers, kernel = n_filters
        full = np.zeros((filt.shape[0], n_filters))
        full[:, lpad:lpad + kernel_size] = filt
        filt = full
    elif form != "slice":
        raise ValueError(f"stft filter form {form!r}: 'slice' or 'zero_pad'")
    return torch.from_numpy(filt).unsqueeze(1).float()


class _FB(nn.Module):
    def __init__(self, filt):
        super().__init__()
        self.register_buffer("_filters", filt)


class _Coder(nn.Module):
    def __init__(self, filt):
        super().__init__()
        self.filterbank = _FB(filt)


class _ComplexConv(nn.Module):
    """Parameter container: re_module / im_module as upstream's ComplexConv2d / ComplexConvTranspose2d."""

    def __init__(self, cin, cout, k, s, transposed, bias):
        super().__init__()
        pad = (k[0] // 2, k[1] // 2)
        klass = nn.ConvTranspose2d if transposed else nn.Conv2d
        self.re_module = klass(cin, cout, k, s, pad, bias=bias)
        self.im_module = klass(cin, cout, k, s, pad, bias=bias)
        self.transposed, self.k, self.s, self.pad = transposed, k, s, pad

    def block_weight(self, in_split=None):
        """Real block weight.  in_split = (Cd, Ce): the input is a skip buffer laid out
        [dec_r, dec_i, enc_r, enc_i] in memory (complex channel order [dec, enc])."""
        wr, wi = self.re_module.weight, self.im_module.weight
        if not self.transposed:                       # (Cout, Cin, kh, kw): rows = outputs
            return torch.cat([torch.cat([wr, -wi], 1), torch.cat([wi, wr], 1)], 0)
        top, bot = torch.cat([wr, wi], 1), torch.cat([-wi, wr], 1)        # rows = inputs (real / imag)
        if in_split is None:
            return torch.cat([top, bot], 0)
        cd = in_split[0]
        return torch.cat([top[:cd], bot[:cd], top[cd:], bot[cd:]], 0)

    def block_bias(self):
        if self.re_module.bias is None:
            return None
        br, bi = self.re_module.bias, self.im_module.bias
        return torch.cat([br - bi, br + bi], 0)


class _ComplexBatchNorm(nn.Module):
    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self.Wrr = nn.Parameter(torch.ones(c))
        self.Wri = nn.Parameter(torch.empty(c).uniform_(-0.9, 0.9))
        self.Wii = nn.Parameter(torch.ones(c))
        self.Br, self.Bi = nn.Parameter(torch.zeros(c)), nn.Parameter(torch.zeros(c))
        for n, v in (("RMr", 0.0), ("RMi", 0.0), ("RVrr", 1.0), ("RVri", 0.0), ("RVii", 1.0)):
            self.register_buffer(n, torch.full((c,), v))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def coef(self, Mr, Mi, Vrr, Vri, Vii):
        """(6, C): Zrr, Zri, Zir, Zii, Br', Bi' with y = Z x + B' (mean folded into the bias)."""
        Vrr, Vii = Vrr + self.eps, Vii + self.eps
        tau, delta = Vrr + Vii, Vrr * Vii - Vri * Vri
        s = delta.sqrt()
        t = (tau + 2 * s).sqrt()
        rst = (s * t).reciprocal()
        Urr, Uii, Uri = (s + Vii) * rst, (s + Vrr) * rst, -Vri * rst
        Zrr, Zri = self.Wrr * Urr + self.Wri * Uri, self.Wrr * Uri + self.Wri * Uii
        Zir, Zii = self.Wri * Urr + self.Wii * Uri, self.Wri * Uri + self.Wii * Uii
        return torch.stack([Zrr, Zri, Zir, Zii, self.Br - (Zrr * Mr + Zri * Mi), self.Bi - (Zir * Mr + Zii * Mi)])


def _alias(buf, c0, c1):
    """A tensor over channels [c0, c1) of `buf` that shares its memory but is NOT an autograd view
    (kernels write into it; a view + in-place write would insert CopySlices clones of the buffer)."""
    t = torch.empty(0, device=buf.device, dtype=buf.dtype)
    return t.set_(buf.untyped_storage(), buf.storage_offset() + c0 * buf.stride(1),
                  (buf.shape[0], c1 - c0, buf.shape[2], buf.shape[3]), buf.stride())


class _Dst:
    """Carries a destination tensor into an autograd.Function without making it a graph input."""

    def __init__(self, t):
        self.t = t


class _CplxNormActFn(torch.autograd.Function):
    """ComplexBatchNorm + LeakyReLU on a stacked (N, 2C, H, W) tensor, writing into `dst` (a channel
    slice of a skip buffer or a fresh tensor)."""

    @staticmethod
    def forward(ctx, y, norm, dst, *params):
        L = _lib.lib()
        y = y.contiguous()
        N, C2, H, W = y.shape
        Cc, S = C2 // 2, H * W
        # coefficients of y = Z x + B' in ONE launch (rfx_cplx_coef_fwd; running statistics updated by the same kernel) -- the torch
        # form (`norm.coef`, kept as the readable statement of the math) cost ~45 launches on (C,) tensors per layer
        pw = [p.detach() for p in params]
        coefc = torch.empty((6, Cc), device=y.device, dtype=torch.float32)
        if norm.training:
            sums = torch.empty(Cc * 5, device=y.device, dtype=torch.float64)
            ws = torch.empty(5 * Cc * int(L.rfx_cplx_slots(N, S)), device=y.device, dtype=torch.float64)
            check(L.rfx_cplx_moments(_ptr(y), N, Cc, S, _ptr(ws), _ptr(sums), _stream()), "rfx_cplx_moments")
            inv = 1.0 / float(N * S)
            check(L.rfx_cplx_coef_fwd(_ptr(sums), inv, None, *[_ptr(t) for t in pw], norm.eps, Cc, _ptr(coefc), None,
                                      _ptr(norm.RMr), _ptr(norm.RMi), _ptr(norm.RVrr), _ptr(norm.RVri), _ptr(norm.RVii),
                                      norm.momentum, _stream()), "rfx_cplx_coef_fwd")
            with torch.no_grad():
                norm.num_batches_tracked += 1
            ctx.stat = (sums, inv, None)
        else:
            stats_in = torch.stack([norm.RMr, norm.RMi, norm.RVrr, norm.RVri, norm.RVii]).float().contiguous()
            check(L.rfx_cplx_coef_fwd(None, 0.0, _ptr(stats_in), *[_ptr(t) for t in pw], norm.eps, Cc, _ptr(coefc), None,
                                      None, None, None, None, None, 0.0, _stream()), "rfx_cplx_coef_fwd")
            ctx.stat = (None, 0.0, stats_in)
        dst = dst.t if dst is not None else torch.empty_like(y)
        assert dst.shape == y.shape and dst.stride(1) == S and dst.stride(3) == 1
        check(L.rfx_cplx_affine_act_fwd(_ptr(y), _ptr(coefc), N, Cc, S, LEAKY, _ptr(dst), dst.stride(0), Cc, _stream()),
              "rfx_cplx_affine_act_fwd")
        ctx.save_for_backward(y, coefc, *pw)
        ctx.norm = norm
        return dst

    @staticmethod
    def backward(ctx, g):
        L = _lib.lib()
        y, coefc, *pw = ctx.saved_tensors
        norm = ctx.norm
        N, C2, H, W = y.shape
        Cc, S = C2 // 2, H * W
        if g.stride(1) != S or g.stride(3) != 1 or g.stride(2) != W:
            g = g.contiguous()
        gx = torch.empty_like(y)
        gcoef = torch.empty((6, Cc), device=y.device, dtype=torch.float32)
        ws = torch.empty(6 * Cc * int(L.rfx_cplx_slots(N, S)), device=y.device, dtype=torch.float64)
        check(L.rfx_cplx_affine_act_bwd(_ptr(y), _ptr(coefc), _ptr(g), g.stride(0), Cc, N, Cc, S, LEAKY, _ptr(gx),
                                        _ptr(ws), _ptr(gcoef), _stream()), "rfx_cplx_affine_act_bwd")
        sums, inv, stats_in = ctx.stat
        gw = torch.empty((5, Cc), device=y.device, dtype=torch.float32)
        cm = torch.empty((5, Cc), device=y.device, dtype=torch.float32) if sums is not None else None
        check(L.rfx_cplx_coef_bwd(_ptr(sums) if sums is not None else None, inv, _ptr(stats_in) if stats_in is not None else None,
                                  *[_ptr(t) for t in pw], norm.eps, Cc, _ptr(gcoef), _ptr(gw), _ptr(cm) if cm is not None else None,
                                  _stream()), "rfx_cplx_coef_bwd")
        if cm is not None:
            check(L.rfx_cplx_moments_bwd(_ptr(y), _ptr(cm), N, Cc, S, _ptr(gx), _stream()), "rfx_cplx_moments_bwd")
        ctx.stat = None
        return (gx, None, None, gw[0], gw[1], gw[2], gw[3], gw[4])


def _norm_act(y, norm, dst=None):
    return _CplxNormActFn.apply(y, norm, _Dst(dst) if dst is not None else None, norm.Wrr, norm.Wri, norm.Wii,
                                norm.Br, norm.Bi)


class _JoinFn(torch.autograd.Function):
    """Declares `buf` = [a | b] along channels (both halves were written in place): zero-copy torch.cat."""

    @staticmethod
    def forward(ctx, a, b, buf):
        ctx.ca = a.shape[1]
        return _alias(buf.t, 0, buf.t.shape[1])

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.ca], g[:, ctx.ca:], None


class _BoundMaskFn(torch.autograd.Function):
    """out = tanh(|m|) m/|m| (*) tf over stacked (N, 2, H, W) planes."""

    @staticmethod
    def forward(ctx, m, tf):
        m = m.contiguous()
        N, _, H, W = m.shape
        P = H * W
        out = torch.empty_like(m)
        check(_lib.lib().rfx_bound_mask_fwd(_ptr(m), _ptr(tf), _ptr(out), N, P, 2 * P, tf.stride(0), 2 * P, _stream()),
              "rfx_bound_mask_fwd")
        ctx.save_for_backward(m, tf)
        return out

    @staticmethod
    def backward(ctx, g):
        m, tf = ctx.saved_tensors
        N, _, H, W = m.shape
        P = H * W
        g = g.contiguous()
        gm = torch.empty_like(m)
        check(_lib.lib().rfx_bound_mask_bwd(_ptr(m), _ptr(tf), _ptr(g), _ptr(gm), N, P, 2 * P, tf.stride(0), 2 * P,
                                            2 * P, _stream()), "rfx_bound_mask_bwd")
        return gm, None


class _EncBlock(nn.Module):
    def __init__(self, cin, cout, k, s):
        super().__init__()
        self.conv = _ComplexConv(cin, cout, k, s, False, bias=False)
        self.norm = _ComplexBatchNorm(cout)


class _DecBlock(nn.Module):
    def __init__(self, cin, cout, k, s):
        super().__init__()
        self.deconv = _ComplexConv(cin, cout, k, s, True, bias=False)
        self.norm = _ComplexBatchNorm(cout)


class _Masker(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoders = nn.ModuleList([_EncBlock(*a) for a in ENCODERS])
        self.decoders = nn.ModuleList([_DecBlock(*a) for a in DECODERS[:-1]])
        cin, cout, k, s = DECODERS[-1]
        self.output_layer = nn.Sequential(_ComplexConv(cin, cout, k, s, True, bias=True))

    def forward(self, x):
        """x: stacked (B, 2, 513, Wp) mixture STFT (time already padded).  Returns the raw mask (B, 2, 513, Wp)."""
        B = x.shape[0]
        nenc = len(self.encoders)
        enc_out, slots = [], []
        for i, enc in enumerate(self.encoders):
            cv = enc.conv
            y = ops.conv2d(x, cv.block_weight(), None, cv.s, cv.pad)
            if i < nenc - 1:           # lives in the skip buffer [dec_r, dec_i, enc_r, enc_i]
                cd, ce = DECODERS[nenc - 2 - i][1], ENCODERS[i][1]
                buf = torch.empty((B, 2 * cd + 2 * ce, y.shape[2], y.shape[3]), device=x.device, dtype=torch.float32)
                x = _norm_act(y, enc.norm, _alias(buf, 2 * cd, 2 * cd + 2 * ce))
                slots.append((buf, cd, ce))
            else:
                x = _norm_act(y, enc.norm)
            enc_out.append(x)
        for j, dec in enumerate(self.decoders):
            cv = dec.deconv
            split = None if j == 0 else (DECODERS[j - 1][1], ENCODERS[nenc - 1 - j][1])
            buf, cd, ce = slots[nenc - 2 - j]
            ho, wo = buf.shape[2], buf.shape[3]
            y = ops.conv_transpose2d(x, cv.block_weight(split), None, cv.s, (1, 1), cv.pad, (ho, wo))
            d = _norm_act(y, dec.norm, _alias(buf, 0, 2 * cd))
            x = _JoinFn.apply(d, enc_out[nenc - 2 - j], _Dst(buf))
        cv = self.output_layer[0]
        split = (DECODERS[-2][1], ENCODERS[0][1])
        ho = (x.shape[2] - 1) * cv.s[0] - 2 * cv.pad[0] + cv.k[0]
        wo = (x.shape[3] - 1) * cv.s[1] - 2 * cv.pad[1] + cv.k[1]
        return ops.conv_transpose2d(x, cv.block_weight(split), cv.block_bias(), cv.s, (1, 1), cv.pad, (ho, wo))


class DCUNet(nn.Module):
    def __init__(self, architecture="Large-DCUNet-20", stft_n_filters=1024, stft_kernel_size=1024, stft_stride=256,
                 sample_rate=16000.0, fix_length_mode=None, stft_filter_form=None, **kwargs):
        super().__init__()
        if architecture != "Large-DCUNet-20" or fix_length_mode != "pad":
            raise NotImplementedError("only the configuration RemFX uses (cfg/model/dcunet.yaml) is built")
        # Shape of asteroid-filterbanks' STFTFB buffers (unpinned: the package is in neither tree): "slice" (default) keeps the
        # kernel_size centred columns, (1026, 1, 512) at RemFX's setting; "zero_pad" pads the window to n_filters,
        # (1026, 1, 1024).  A released checkpoint loads strictly with exactly one of them: RFX_DCUNET_FILTER_FORM or the
        # keyword selects it (oracle/ref_dcunet.py takes the same keyword).
        form = stft_filter_form or os.environ.get("RFX_DCUNET_FILTER_FORM", "slice")
        filt = stft_filters(stft_n_filters, stft_kernel_size, stft_stride, form)
        self.stride, self.kernel_size, self.n_filters = stft_stride, filt.shape[-1], stft_n_filters
        self.encoder, self.decoder = _Coder(filt), _Coder(filt.clone())
        self.masker = _Masker()

    def forward(self, wav):
        ops._req(wav, "wav")
        x = wav.unsqueeze(1) if wav.dim() == 2 else wav
        B, _, T = x.shape
        F = self.n_filters // 2 + 1
        frames = (T - self.kernel_size) // self.stride + 1
        if (F - 1) % 256:
            raise ValueError("frequency axis must satisfy (F - 1) % 256 == 0")
        wp = frames + (-(frames - 1)) % 16                    # masker wants (frames - 1) % 16 == 0: zero pad right
        tf = torch.zeros((B, 2, F, wp), device=x.device, dtype=torch.float32)
        with torch.no_grad():                                 # fixed STFT filterbank, input needs no gradient
            out_view = tf.as_strided((B, 2 * F, 1, frames), (2 * F * wp, wp, wp, 1))
            ops.conv2d_forward(x.unsqueeze(2), self.encoder.filterbank._filters.unsqueeze(2), None, (1, self.stride),
                               (0, 0), (1, 1), out=out_view)
        m = self.masker(tf)
        masked = _BoundMaskFn.apply(m, tf)                    # (B, 2, F, wp); columns >= frames are never read
        rep = masked.as_strided((B, 2 * F, frames), (2 * F * wp, wp, 1))
        out_len = (frames - 1) * self.stride + self.kernel_size
        out = ops.conv_transpose1d(rep, self.decoder.filterbank._filters, None, self.stride, 1, 0, out_len)
        if out_len < T:
            out = torch.nn.functional.pad(out, (0, T - out_len))
        return out[..., :T]                                   # (B, n_src = 1, T)

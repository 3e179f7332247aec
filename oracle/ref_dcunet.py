"""Oracle (test infrastructure): asteroid DCUNet "Large-DCUNet-20" as RemFX configures it.

asteroid / asteroid-filterbanks are un-vendored dependencies (setup.py: bare ``asteroid``;
absent from /root/reference and from this image) -> PARITY UNPINNED.  Restates the
published architecture following SURVEY.md appendix A.2 over torch CPU ops.
Reference call sites: remfx/models.py:347-367 (DCUNetModel), cfg/model/dcunet.yaml:11-16
(architecture="Large-DCUNet-20", stft_kernel_size=512, fix_length_mode="pad";
stft_n_filters=1024 and stft_stride=256 stay at their defaults).

state_dict names follow upstream: encoder.filterbank._filters, decoder.filterbank._filters,
masker.encoders.{i}.conv.{re_module,im_module}.weight, masker.encoders.{i}.norm.*,
masker.decoders.{i}.deconv.{re_module,im_module}.weight, masker.decoders.{i}.norm.*,
masker.output_layer.0.{re_module,im_module}.{weight,bias}.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ENCODERS = ((1, 45, (7, 1), (1, 1)), (45, 45, (1, 7), (1, 1)), (45, 90, (7, 5), (2, 2)), (90, 90, (7, 5), (2, 1)),
            (90, 90, (5, 3), (2, 2)), (90, 90, (5, 3), (2, 1)), (90, 90, (5, 3), (2, 2)), (90, 90, (5, 3), (2, 1)),
            (90, 90, (5, 3), (2, 2)), (90, 128, (5, 3), (2, 1)))
DECODERS = ((128, 90, (5, 3), (2, 1)), (180, 90, (5, 3), (2, 2)), (180, 90, (5, 3), (2, 1)), (180, 90, (5, 3), (2, 2)),
            (180, 90, (5, 3), (2, 1)), (180, 90, (5, 3), (2, 2)), (180, 90, (7, 5), (2, 1)), (180, 90, (7, 5), (2, 2)),
            (135, 90, (1, 7), (1, 1)), (135, 1, (7, 1), (1, 1)))


def stft_filters(n_filters=1024, kernel_size=512, stride=256, form="slice"):
    """asteroid-filterbanks STFTFB: windowed, centred part of the DFT basis, (n_filters+2, 1, kernel)."""
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


class ComplexConv(nn.Module):
    def __init__(self, cin, cout, k, s, transposed, bias):
        super().__init__()
        pad = (k[0] // 2, k[1] // 2)
        klass = nn.ConvTranspose2d if transposed else nn.Conv2d
        self.re_module = klass(cin, cout, k, s, pad, bias=bias)
        self.im_module = klass(cin, cout, k, s, pad, bias=bias)

    def forward(self, x):
        return torch.complex(self.re_module(x.real) - self.im_module(x.imag),
                             self.re_module(x.imag) + self.im_module(x.real))


class ComplexBatchNorm(nn.Module):
    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.eps, self.momentum = eps, momentum
        self.Wrr, self.Wri, self.Wii = nn.Parameter(torch.ones(c)), nn.Parameter(torch.empty(c).uniform_(-0.9, 0.9)), nn.Parameter(torch.ones(c))
        self.Br, self.Bi = nn.Parameter(torch.zeros(c)), nn.Parameter(torch.zeros(c))
        for n, v in (("RMr", 0.0), ("RMi", 0.0), ("RVrr", 1.0), ("RVri", 0.0), ("RVii", 1.0)):
            self.register_buffer(n, torch.full((c,), v))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

    def forward(self, x):
        xr, xi = x.real, x.imag
        v = (1, -1, 1, 1)
        if self.training:
            Mr, Mi = xr.mean((0, 2, 3), keepdim=True), xi.mean((0, 2, 3), keepdim=True)
            with torch.no_grad():
                self.num_batches_tracked += 1
                self.RMr.lerp_(Mr.reshape(-1), self.momentum); self.RMi.lerp_(Mi.reshape(-1), self.momentum)
        else:
            Mr, Mi = self.RMr.view(v), self.RMi.view(v)
        xr, xi = xr - Mr, xi - Mi
        if self.training:
            Vrr, Vri, Vii = (xr * xr).mean((0, 2, 3), keepdim=True), (xr * xi).mean((0, 2, 3), keepdim=True), (xi * xi).mean((0, 2, 3), keepdim=True)
            with torch.no_grad():
                self.RVrr.lerp_(Vrr.reshape(-1), self.momentum); self.RVri.lerp_(Vri.reshape(-1), self.momentum)
                self.RVii.lerp_(Vii.reshape(-1), self.momentum)
        else:
            Vrr, Vri, Vii = self.RVrr.view(v), self.RVri.view(v), self.RVii.view(v)
        Vrr, Vii = Vrr + self.eps, Vii + self.eps
        tau, delta = Vrr + Vii, Vrr * Vii - Vri * Vri
        s = delta.sqrt()
        t = (tau + 2 * s).sqrt()
        rst = (s * t).reciprocal()
        Urr, Uii, Uri = (s + Vii) * rst, (s + Vrr) * rst, -Vri * rst
        Wrr, Wri, Wii = self.Wrr.view(v), self.Wri.view(v), self.Wii.view(v)
        Zrr, Zri = Wrr * Urr + Wri * Uri, Wrr * Uri + Wri * Uii
        Zir, Zii = Wri * Urr + Wii * Uri, Wri * Uri + Wii * Uii
        return torch.complex(Zrr * xr + Zri * xi + self.Br.view(v), Zir * xr + Zii * xi + self.Bi.view(v))


class _EncBlock(nn.Module):
    def __init__(self, cin, cout, k, s):
        super().__init__()
        self.conv = ComplexConv(cin, cout, k, s, False, bias=False)
        self.norm = ComplexBatchNorm(cout)

    def forward(self, x):
        y = self.norm(self.conv(x))
        return torch.complex(F.leaky_relu(y.real, 0.01), F.leaky_relu(y.imag, 0.01))


class _DecBlock(nn.Module):
    def __init__(self, cin, cout, k, s):
        super().__init__()
        self.deconv = ComplexConv(cin, cout, k, s, True, bias=False)
        self.norm = ComplexBatchNorm(cout)

    def forward(self, x):
        y = self.norm(self.deconv(x))
        return torch.complex(F.leaky_relu(y.real, 0.01), F.leaky_relu(y.imag, 0.01))


class _Masker(nn.Module):
    def __init__(self):
        super().__init__()
        self.encoders = nn.ModuleList([_EncBlock(*a) for a in ENCODERS])
        self.decoders = nn.ModuleList([_DecBlock(*a) for a in DECODERS[:-1]])
        cin, cout, k, s = DECODERS[-1]
        self.output_layer = nn.Sequential(ComplexConv(cin, cout, k, s, True, bias=True))

    def forward(self, tf):                        # tf: (B, F, frames) complex
        frames = tf.shape[-1]
        padn = (-(frames - 1)) % 16                # time: (frames - 1) % 16 == 0 ("pad" mode, zeros on the right)
        x = F.pad(torch.view_as_real(tf), (0, 0, 0, padn))
        x = torch.view_as_complex(x.contiguous()).unsqueeze(1)
        outs = []
        for enc in self.encoders:
            x = enc(x)
            outs.append(x)
        for eo, dec in zip(reversed(outs[:-1]), self.decoders):
            x = torch.cat([dec(x), eo], dim=1)
        m = self.output_layer(x)
        mag = m.abs()
        m = torch.tanh(mag) * m / mag             # BoundComplexMask("tanh")
        return m[..., :frames]


class DCUNet(nn.Module):
    def __init__(self, architecture="Large-DCUNet-20", stft_n_filters=1024, stft_kernel_size=1024, stft_stride=256,
                 sample_rate=16000.0, fix_length_mode=None, stft_filter_form="slice"):
        super().__init__()
        assert architecture == "Large-DCUNet-20" and fix_length_mode == "pad"
        # stft_filter_form (not an upstream keyword): asteroid-filterbanks is absent here, so the buffer shape of its STFTFB
        # cannot be pinned -- "slice" = the kernel_size centred columns of the DFT basis, (1026, 1, 512) at RemFX's setting
        # (this restatement's reading of the published code); "zero_pad" = the window padded to n_filters, (1026, 1, 1024)
        filt = stft_filters(stft_n_filters, stft_kernel_size, stft_stride, stft_filter_form)
        self.stride = stft_stride
        self.encoder, self.decoder = _Coder(filt), _Coder(filt.clone())
        self.masker = _Masker()

    def forward(self, wav):
        x = wav.unsqueeze(1) if wav.dim() == 2 else wav
        spec = F.conv1d(x, self.encoder.filterbank._filters, stride=self.stride)       # (B, 1026, frames)
        nf = spec.shape[1] // 2
        tf = torch.complex(spec[:, :nf], spec[:, nf:])
        masked = self.masker(tf) * tf.unsqueeze(1)                                    # (B, 1, F, frames)
        rep = torch.cat([masked.real, masked.imag], dim=2).squeeze(1)                 # (B, 2F, frames)
        out = F.conv_transpose1d(rep, self.decoder.filterbank._filters, stride=self.stride)
        T = x.shape[-1]
        out = F.pad(out, (0, T - out.shape[-1])) if out.shape[-1] < T else out[..., :T]
        return out                                                                    # (B, n_src=1, T)

"""Oracle (test infrastructure): Hybrid Demucs as RemFX configures it.

``torchaudio.models.HDemucs`` is an un-vendored dependency (setup.py:
``torchaudio>=0.13.0``; absent from /root/reference and from this image) ->
PARITY UNPINNED.  This is a restatement of the published Hybrid Demucs v3
algorithm following SURVEY.md appendix A.1, written over plain torch CPU ops.
Reference call sites: remfx/models.py:307-324 (DemucsModel),
cfg/model/demucs.yaml:11-16 (sources=["mixture"], audio_channels=1, nfft=4096,
channels=48).

Module / parameter names follow the upstream state_dict contract
(freq_encoder.{i}.conv, .norm1, .rewrite, .norm2, .dconv.layers.{d}.{idx},
time_encoder, freq_decoder.{j}.conv_tr/.norm2/.rewrite/.norm1, time_decoder,
freq_emb.embedding.weight) so strict checkpoint loads line up.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class ScaledEmbedding(nn.Module):
    def __init__(self, n, dim, scale=10.0, smooth=False):
        super().__init__()
        self.embedding = nn.Embedding(n, dim)
        if smooth:
            w = torch.cumsum(self.embedding.weight.data, dim=0)
            w = w / torch.arange(1, n + 1).sqrt()[:, None]
            self.embedding.weight.data[:] = w
        self.embedding.weight.data /= scale
        self.scale = scale

    def forward(self, idx):
        return self.embedding(idx) * self.scale


class LayerScale(nn.Module):
    def __init__(self, channels, init=0.0):
        super().__init__()
        self.scale = nn.Parameter(torch.full((channels,), float(init)))

    def forward(self, x):
        return self.scale[:, None] * x


class BLSTM(nn.Module):
    def __init__(self, dim, layers=2, skip=True):
        super().__init__()
        self.max_steps = 200
        self.lstm = nn.LSTM(bidirectional=True, num_layers=layers, hidden_size=dim, input_size=dim)
        self.linear = nn.Linear(2 * dim, dim)
        self.skip = skip

    def forward(self, x):
        B, C, T = x.shape
        y = x
        framed = T > self.max_steps
        if framed:
            width, stride = self.max_steps, self.max_steps // 2
            nfr = math.ceil(T / stride)
            xp = F.pad(x, (0, (nfr - 1) * stride + width - T))
            frames = xp.unfold(-1, width, stride)              # (B, C, nfr, width)
            x = frames.permute(0, 2, 1, 3).reshape(-1, C, width)
        x = x.permute(2, 0, 1)
        x = self.linear(self.lstm(x)[0]).permute(1, 2, 0)
        if framed:
            fr = x.reshape(B, nfr, C, width)
            lim = stride // 2
            parts = []
            for k in range(nfr):
                if k == 0:
                    parts.append(fr[:, k, :, :-lim])
                elif k == nfr - 1:
                    parts.append(fr[:, k, :, lim:])
                else:
                    parts.append(fr[:, k, :, lim:-lim])
            x = torch.cat(parts, -1)[..., :T]
        return x + y if self.skip else x


class LocalState(nn.Module):
    def __init__(self, channels, heads=4, ndecay=4):
        super().__init__()
        self.heads, self.ndecay = heads, ndecay
        self.content = nn.Conv1d(channels, channels, 1)
        self.query = nn.Conv1d(channels, channels, 1)
        self.key = nn.Conv1d(channels, channels, 1)
        self.query_decay = nn.Conv1d(channels, heads * ndecay, 1)
        self.query_decay.weight.data *= 0.01
        self.query_decay.bias.data[:] = -2
        self.proj = nn.Conv1d(channels, channels, 1)

    def forward(self, x):
        B, C, T = x.shape
        h = self.heads
        idx = torch.arange(T, device=x.device, dtype=x.dtype)
        delta = idx[:, None] - idx[None, :]
        q = self.query(x).view(B, h, -1, T)
        k = self.key(x).view(B, h, -1, T)
        dots = torch.einsum("bhct,bhcs->bhts", k, q) / math.sqrt(k.shape[2])
        decays = torch.arange(1, self.ndecay + 1, device=x.device, dtype=x.dtype)
        dq = torch.sigmoid(self.query_decay(x).view(B, h, -1, T)) / 2
        kern = -decays.view(-1, 1, 1) * delta.abs() / math.sqrt(self.ndecay)
        dots = dots + torch.einsum("fts,bhfs->bhts", kern, dq)
        dots = dots.masked_fill(torch.eye(T, device=x.device, dtype=torch.bool), -100)
        w = torch.softmax(dots, dim=2)
        c = self.content(x).view(B, h, -1, T)
        res = torch.einsum("bhts,bhct->bhcs", w, c).reshape(B, -1, T)
        return x + self.proj(res)


class DConv(nn.Module):
    def __init__(self, channels, compress=4, depth=2, init=1e-4, attn=False, heads=4,
                 ndecay=4, lstm=False, kernel_size=3):
        super().__init__()
        hidden = int(channels / compress)
        self.layers = nn.ModuleList()
        for d in range(depth):
            dil = 2 ** d
            mods = [nn.Conv1d(channels, hidden, kernel_size, dilation=dil, padding=dil * (kernel_size // 2)),
                    nn.GroupNorm(1, hidden), nn.GELU(),
                    nn.Conv1d(hidden, 2 * channels, 1), nn.GroupNorm(1, 2 * channels), nn.GLU(1),
                    LayerScale(channels, init)]
            if attn:
                mods.insert(3, LocalState(hidden, heads=heads, ndecay=ndecay))
            if lstm:
                mods.insert(3, BLSTM(hidden, layers=2, skip=True))
            self.layers.append(nn.Sequential(*mods))

    def forward(self, x):
        for layer in self.layers:
            x = x + layer(x)
        return x


def _norm(groups, ch, on):
    return nn.GroupNorm(groups, ch) if on else nn.Identity()


class HEncLayer(nn.Module):
    def __init__(self, chin, chout, kernel_size=8, stride=4, norm_groups=4, empty=False,
                 freq=True, norm=False, context=0, dconv_kw=None, pad=True):
        super().__init__()
        padv = kernel_size // 4 if pad else 0
        self.freq, self.stride, self.empty = freq, stride, empty
        if freq:
            self.conv = nn.Conv2d(chin, chout, (kernel_size, 1), (stride, 1), (padv, 0))
        else:
            self.conv = nn.Conv1d(chin, chout, kernel_size, stride, padv)
        self.norm1 = _norm(norm_groups, chout, norm)
        if empty:
            self.rewrite, self.norm2, self.dconv = nn.Identity(), nn.Identity(), nn.Identity()
        else:
            k = 1 + 2 * context
            klass = nn.Conv2d if freq else nn.Conv1d
            self.rewrite = klass(chout, 2 * chout, k, 1, context)
            self.norm2 = _norm(norm_groups, 2 * chout, norm)
            self.dconv = DConv(chout, **(dconv_kw or {}))

    def forward(self, x, inject=None):
        if not self.freq and x.dim() == 4:
            x = x.view(x.shape[0], -1, x.shape[-1])
        if not self.freq and x.shape[-1] % self.stride:
            x = F.pad(x, (0, self.stride - x.shape[-1] % self.stride))
        y = self.conv(x)
        if self.empty:
            return y
        if inject is not None:
            if inject.dim() == 3 and y.dim() == 4:
                inject = inject[:, :, None]
            y = y + inject
        y = F.gelu(self.norm1(y))
        if self.freq:
            B, C, Fr, T = y.shape
            y = self.dconv(y.permute(0, 2, 1, 3).reshape(-1, C, T))
            y = y.view(B, Fr, C, T).permute(0, 2, 1, 3)
        else:
            y = self.dconv(y)
        return F.glu(self.norm2(self.rewrite(y)), dim=1)


class HDecLayer(nn.Module):
    def __init__(self, chin, chout, last=False, kernel_size=8, stride=4, norm_groups=1,
                 empty=False, freq=True, norm=False, context=1, pad=True):
        super().__init__()
        self.pad = (kernel_size - stride) // 2 if pad else 0
        self.last, self.freq, self.chin, self.empty = last, freq, chin, empty
        if freq:
            self.conv_tr = nn.ConvTranspose2d(chin, chout, (kernel_size, 1), (stride, 1))
        else:
            self.conv_tr = nn.ConvTranspose1d(chin, chout, kernel_size, stride)
        self.norm2 = _norm(norm_groups, chout, norm)
        if empty:
            self.rewrite, self.norm1 = nn.Identity(), nn.Identity()
        else:
            klass = nn.Conv2d if freq else nn.Conv1d
            self.rewrite = klass(chin, 2 * chin, 1 + 2 * context, 1, context)
            self.norm1 = _norm(norm_groups, 2 * chin, norm)

    def forward(self, x, skip, length):
        if self.freq and x.dim() == 3:
            x = x.view(x.shape[0], self.chin, -1, x.shape[-1])
        if not self.empty:
            x = x + skip
            y = F.glu(self.norm1(self.rewrite(x)), dim=1)
        else:
            y = x
        z = self.norm2(self.conv_tr(y))
        if self.freq:
            if self.pad:
                z = z[..., self.pad:-self.pad, :]
        else:
            z = z[..., self.pad:self.pad + length]
        if not self.last:
            z = F.gelu(z)
        return z, y


class HDemucs(nn.Module):
    def __init__(self, sources, audio_channels=2, channels=48, growth=2, nfft=4096, depth=6,
                 freq_emb=0.2, emb_scale=10, emb_smooth=True, kernel_size=8, time_stride=2,
                 stride=4, context=1, context_enc=0, norm_starts=4, norm_groups=4,
                 dconv_depth=2, dconv_comp=4, dconv_attn=4, dconv_lstm=4, dconv_init=1e-4):
        super().__init__()
        self.depth, self.nfft, self.audio_channels, self.sources = depth, nfft, audio_channels, list(sources)
        self.hop_length = nfft // 4
        self.freq_emb = None
        self.freq_encoder, self.freq_decoder = nn.ModuleList(), nn.ModuleList()
        self.time_encoder, self.time_decoder = nn.ModuleList(), nn.ModuleList()
        chin, chin_z = audio_channels, audio_channels * 2
        chout, chout_z = channels, channels
        freqs = nfft // 2
        for index in range(depth):
            lstm, attn = index >= dconv_lstm, index >= dconv_attn
            norm = index >= norm_starts
            freq = freqs > 1
            stri, ker = stride, kernel_size
            if not freq:
                ker, stri = time_stride * 2, time_stride
            pad, last_freq = True, False
            if freq and freqs <= kernel_size:
                ker, pad, last_freq = freqs, False, True
            dkw = dict(lstm=lstm, attn=attn, depth=dconv_depth, compress=dconv_comp, init=dconv_init)
            kw = dict(kernel_size=ker, stride=stri, freq=freq, pad=pad, norm=norm, norm_groups=norm_groups)
            kwt = dict(kw, freq=False, kernel_size=kernel_size, stride=stride, pad=True)
            if last_freq:
                chout_z = max(chout, chout_z)
                chout = chout_z
            self.freq_encoder.append(HEncLayer(chin_z, chout_z, context=context_enc, dconv_kw=dkw, **kw))
            if freq:
                if last_freq and nfft == 2048:
                    kwt["stride"], kwt["kernel_size"] = 2, 4
                self.time_encoder.append(HEncLayer(chin, chout, context=context_enc, empty=last_freq,
                                                   dconv_kw=dkw, **kwt))
            if index == 0:
                chin = audio_channels * len(self.sources)
                chin_z = chin * 2
            self.freq_decoder.insert(0, HDecLayer(chout_z, chin_z, last=index == 0, context=context, **kw))
            if freq:
                self.time_decoder.insert(0, HDecLayer(chout, chin, empty=last_freq, last=index == 0,
                                                      context=context, **kwt))
            chin, chin_z = chout, chout_z
            chout, chout_z = int(growth * chout), int(growth * chout_z)
            if freq:
                freqs = 1 if freqs <= kernel_size else freqs // stride
            if index == 0 and freq_emb:
                self.freq_emb = ScaledEmbedding(freqs, chin_z, smooth=emb_smooth, scale=emb_scale)
                self.freq_emb_scale = freq_emb
        # init-time rescale of every conv / transposed conv (reference=0.1)
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.ConvTranspose1d, nn.Conv2d, nn.ConvTranspose2d)):
                s = (m.weight.std().detach() / 0.1) ** 0.5
                m.weight.data /= s
                if m.bias is not None:
                    m.bias.data /= s

    # -- spectral front / back end ------------------------------------------------
    def _spec(self, x):
        hl, nfft = self.hop_length, self.nfft
        T = x.shape[-1]
        le = math.ceil(T / hl)
        pad = hl // 2 * 3
        x = F.pad(x, (pad, pad + le * hl - T), mode="reflect")
        shp = x.shape[:-1]
        z = torch.stft(x.reshape(-1, x.shape[-1]), nfft, hl, window=torch.hann_window(nfft).to(x),
                       win_length=nfft, normalized=True, center=True, return_complex=True,
                       pad_mode="reflect")
        z = z.view(*shp, z.shape[-2], z.shape[-1])[..., :-1, :]
        assert z.shape[-1] == le + 4
        return z[..., 2:2 + le]

    def _ispec(self, z, length):
        hl = self.hop_length
        z = F.pad(F.pad(z, (0, 0, 0, 1)), (2, 2))
        pad = hl // 2 * 3
        le = hl * math.ceil(length / hl) + 2 * pad
        shp = z.shape[:-2]
        nfft = 2 * z.shape[-2] - 2
        x = torch.istft(z.reshape(-1, z.shape[-2], z.shape[-1]), nfft, hl,
                        window=torch.hann_window(nfft).to(z.real), win_length=nfft,
                        normalized=True, length=le, center=True)
        return x.view(*shp, x.shape[-1])[..., pad:pad + length]

    def forward(self, inp):
        length = inp.shape[-1]
        z = self._spec(inp)
        B, C, Fq, T = z.shape
        x = torch.view_as_real(z).permute(0, 1, 4, 2, 3).reshape(B, C * 2, Fq, T)
        mean = x.mean(dim=(1, 2, 3), keepdim=True)
        std = x.std(dim=(1, 2, 3), keepdim=True)
        x = (x - mean) / (1e-5 + std)
        xt = inp
        meant = xt.mean(dim=(1, 2), keepdim=True)
        stdt = xt.std(dim=(1, 2), keepdim=True)
        xt = (xt - meant) / (1e-5 + stdt)
        saved, saved_t, lengths, lengths_t = [], [], [], []
        for idx, enc in enumerate(self.freq_encoder):
            lengths.append(x.shape[-1])
            inject = None
            if idx < len(self.time_encoder):
                lengths_t.append(xt.shape[-1])
                tenc = self.time_encoder[idx]
                xt = tenc(xt)
                if not tenc.empty:
                    saved_t.append(xt)
                else:
                    inject = xt
            x = enc(x, inject)
            if idx == 0 and self.freq_emb is not None:
                frs = torch.arange(x.shape[-2], device=x.device)
                emb = self.freq_emb(frs).t()[None, :, :, None].expand_as(x)
                x = x + self.freq_emb_scale * emb
            saved.append(x)
        x = torch.zeros_like(x)
        xt = torch.zeros_like(x)
        offset = self.depth - len(self.time_decoder)
        for idx, dec in enumerate(self.freq_decoder):
            x, pre = dec(x, saved.pop(-1), lengths.pop(-1))
            if idx >= offset:
                tdec = self.time_decoder[idx - offset]
                length_t = lengths_t.pop(-1)
                if tdec.empty:
                    xt, _ = tdec(pre[:, :, 0], None, length_t)
                else:
                    xt, _ = tdec(xt, saved_t.pop(-1), length_t)
        S = len(self.sources)
        x = x.view(B, S, -1, Fq, T) * std[:, None] + mean[:, None]
        zc = torch.view_as_complex(x.view(B, S, -1, 2, Fq, T).permute(0, 1, 2, 4, 5, 3).contiguous())
        x = self._ispec(zc, length)
        xt = xt.view(B, S, -1, length) * stdt[:, None] + meant[:, None]
        return xt + x

"""Open-Unmix (sigsep/open-unmix-pytorch `OpenUnmix` + `Separator` as RemFX configures them:
reference remfx/models.py:259-304, cfg/model/umx.yaml:11-16) on the HIP kernels.

Same parameter names as upstream (fc1/bn1/lstm/fc2/bn2/fc3/bn3, input_mean, input_scale,
output_scale, output_mean).  Layout: the network runs channel-major, (1, features, frames*batch), so
every Linear is a 1x1 gather-GEMM with the position axis contiguous and BatchNorm1d is the same
per-channel reduction kernel the classifier uses.  STFT / magnitude, the magnitude x mixture-phase
product and the iSTFT are HIP kernels; the 3-layer BiLSTM is the persistent HIP recurrence of remfx_amd/lstm.py.
"""
import torch
import torch.nn as nn

from . import _lib, lstm, nnops, ops, stft
from ._lib import check
from .ops import _ptr, _stream


class OpenUnmix(nn.Module):
    def __init__(self, nb_bins=4096, nb_channels=2, hidden_size=512, nb_layers=3, unidirectional=False,
                 input_mean=None, input_scale=None, max_bin=None):
        super().__init__()
        if max_bin is not None or input_mean is not None or input_scale is not None:
            raise NotImplementedError("RemFX constructs OpenUnmix(nb_channels, nb_bins) only (models.py:278-281)")
        self.nb_output_bins = self.nb_bins = nb_bins
        self.hidden_size = hidden_size
        self.fc1 = nn.Linear(nb_bins * nb_channels, hidden_size, bias=False)
        self.bn1 = nn.BatchNorm1d(hidden_size)
        lstm_hidden = hidden_size if unidirectional else hidden_size // 2
        self.lstm = nn.LSTM(hidden_size, lstm_hidden, num_layers=nb_layers, bidirectional=not unidirectional,
                            batch_first=False, dropout=0.4 if nb_layers > 1 else 0)
        self.fc2 = nn.Linear(hidden_size * 2, hidden_size, bias=False)
        self.bn2 = nn.BatchNorm1d(hidden_size)
        self.fc3 = nn.Linear(hidden_size, nb_bins * nb_channels, bias=False)
        self.bn3 = nn.BatchNorm1d(nb_bins * nb_channels)
        self.input_mean = nn.Parameter(torch.zeros(nb_bins))
        self.input_scale = nn.Parameter(torch.ones(nb_bins))
        self.output_scale = nn.Parameter(torch.ones(nb_bins))
        self.output_mean = nn.Parameter(torch.ones(nb_bins))

    def forward(self, x):
        """x: (B, C=1, bins, frames) magnitude -> same shape."""
        ops._req(x, "x")
        B, Cc, nb, nf = x.shape
        if Cc != 1:
            raise NotImplementedError("mono only (cfg/model/umx.yaml: n_channels 1)")
        P = nf * B
        # channel-major: (1, bins, frames*batch), position p = f*B + b  (== upstream's (F, B, 1, bins) rows)
        mix = x.detach().reshape(B, nb, nf).permute(1, 2, 0).reshape(1, nb, P).contiguous()
        h = (mix + self.input_mean.view(1, -1, 1)) * self.input_scale.view(1, -1, 1)
        h = ops.conv1d(h, self.fc1.weight.unsqueeze(-1))
        h = ops.activation(nnops.batch_norm(h, self.bn1, self.bn1.training), "tanh")            # (1, 512, P)
        lo = lstm.blstm(self.lstm, h, nf, B)                                                         # (1, 512, P)
        cat = torch.cat([h, lo], 1)
        h = nnops.batch_norm(ops.conv1d(cat, self.fc2.weight.unsqueeze(-1)), self.bn2, self.bn2.training, relu=True)
        h = nnops.batch_norm(ops.conv1d(h, self.fc3.weight.unsqueeze(-1)), self.bn3, self.bn3.training)
        h = h * self.output_scale.view(1, -1, 1) + self.output_mean.view(1, -1, 1)
        y = ops.activation(h, "relu") * mix                                                          # (1, bins, P)
        return y.view(nb, nf, B).permute(2, 0, 1).reshape(B, 1, nb, nf)


class _PhaseMaskFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mag, xc):
        mag = mag.contiguous()
        out = torch.empty_like(xc)
        check(_lib.lib().rfx_phase_mask_fwd(_ptr(mag), _ptr(xc), _ptr(out), mag.numel(), _stream()), "rfx_phase_mask_fwd")
        ctx.save_for_backward(xc)
        return out

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        gm = torch.empty(xc.shape[:-1], device=xc.device, dtype=torch.float32)
        check(_lib.lib().rfx_phase_mask_bwd(_ptr(xc), _ptr(g.contiguous()), _ptr(gm), gm.numel(), _stream()),
              "rfx_phase_mask_bwd")
        return gm, None


class Separator(nn.Module):
    """umx Separator(target_models={"other": model}, niter=0, softmask=False, residual=False):
    (B, 1, T) -> (B, 1, 1, T)."""

    def __init__(self, target_models, niter=0, softmask=False, residual=False, sample_rate=44100.0, n_fft=4096,
                 n_hop=1024, nb_channels=2, wiener_win_len=300, filterbank="torch"):
        super().__init__()
        if niter != 0 or softmask or residual or len(target_models) != 1:
            raise NotImplementedError("RemFX uses the default niter=0 single-target Separator (models.py:282-288)")
        self.target_models = nn.ModuleDict(target_models)
        self.n_fft, self.n_hop, self.nb_channels = n_fft, n_hop, nb_channels
        self.register_buffer("sample_rate", torch.as_tensor(sample_rate), persistent=False)

    def forward(self, audio):
        ops._req(audio, "audio")
        B, Cc, T = audio.shape
        xc = stft.stft(audio.reshape(B * Cc, T), self.n_fft, self.n_hop, mode="complex").detach()   # (B, bins, F, 2)
        X = stft.stft(audio.reshape(B * Cc, T), self.n_fft, self.n_hop, mode="mag", eps=0.0)        # |X|
        model = next(iter(self.target_models.values()))
        mag = model(X.view(B, Cc, X.shape[-2], X.shape[-1]))
        Y = _PhaseMaskFn.apply(mag.reshape(B * Cc, mag.shape[-2], mag.shape[-1]), xc)
        y = stft.istft(Y, self.n_fft, self.n_hop, mode="complex", length=T)
        return y.view(B, 1, Cc, T)

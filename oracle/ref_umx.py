"""Oracle (test infrastructure): Open-Unmix model + Separator as RemFX configures them.

sigsep/open-unmix-pytorch is an EMPTY git submodule in the reference (`umx/`, .gitmodules:1-3, no
recorded SHA) and is not installed here -> PARITY UNPINNED.  Restates the published model following
SURVEY.md appendix A.3 over torch CPU ops.  Reference call sites: remfx/models.py:259-304
(OpenUnmixModel: spectrogram -> dead `Y = self.model(X)` -> `self.separator(x).squeeze(1)`),
cfg/model/umx.yaml:11-16 (n_fft 2048, hop 512, 1 channel, alpha 0.3).

state_dict names follow upstream OpenUnmix: fc1/bn1/lstm/fc2/bn2/fc3/bn3, input_mean, input_scale,
output_scale, output_mean.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class OpenUnmix(nn.Module):
    def __init__(self, nb_bins=4096, nb_channels=2, hidden_size=512, nb_layers=3, unidirectional=False):
        super().__init__()
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

    def forward(self, x):                                   # (B, C, bins, frames)
        x = x.permute(3, 0, 1, 2)
        nf, ns, nc, nb = x.shape
        mix = x.detach().clone()
        x = (x + self.input_mean) * self.input_scale
        x = self.fc1(x.reshape(-1, nc * nb))
        x = torch.tanh(self.bn1(x).reshape(nf, ns, self.hidden_size))
        x = torch.cat([x, self.lstm(x)[0]], -1)
        x = F.relu(self.bn2(self.fc2(x.reshape(-1, x.shape[-1]))))
        x = self.bn3(self.fc3(x)).reshape(nf, ns, nc, nb)
        x = x * self.output_scale + self.output_mean
        return (F.relu(x) * mix).permute(1, 2, 3, 0)


def separator(model, wav, n_fft=2048, n_hop=512):
    """Separator(target_models={"other": model}, niter=0, softmask=False, residual=False):
    (B, 1, T) -> (B, n_targets=1, 1, T)."""
    B, C, T = wav.shape
    win = torch.hann_window(n_fft)
    X = torch.stft(wav.reshape(-1, T), n_fft, n_hop, window=win, center=True, normalized=False, onesided=True,
                   pad_mode="reflect", return_complex=True).reshape(B, C, n_fft // 2 + 1, -1)
    mag = model(X.abs().detach().clone())                   # (B, C, bins, frames)
    ang = torch.atan2(X.imag, X.real)                       # niter=0 "wiener": magnitude x mixture phase
    Y = torch.complex(mag * torch.cos(ang), mag * torch.sin(ang))
    y = torch.istft(Y.reshape(-1, Y.shape[-2], Y.shape[-1]), n_fft, n_hop, window=win, center=True, length=T)
    return y.reshape(B, 1, C, T)

"""PANNs-style Cnn14 effect classifier on the HIP kernels (mirror of reference
remfx/classifier.py:134-284: Cnn14, ConvBlock).

Same constructor arguments and state_dict keys as the reference (window,
melspec.spectrogram.window, melspec.mel_scale.fb, bn0.*, conv_block{1..6}.{conv1,conv2}.weight,
.bn{1,2}.*, fc1.*, heads.{k}.*).  The mel front end restates
torchaudio.transforms.MelSpectrogram (power-2 STFT + HTK mel filterbank):
the power spectrogram comes out of the framed-FFT kernel's epilogue and the filterbank
is a 1025 -> n_mels 1x1 gather-GEMM.  Quirks kept: no log, per-clip
standardisation with unbiased std and no epsilon, bn0 / window present but unused.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import nnops, ops, stft
from .utils import init_bn, init_layer


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate):
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


class _Spectrogram(nn.Module):
    def __init__(self, n_fft):
        super().__init__()
        self.register_buffer("window", torch.hann_window(n_fft))


class _MelScale(nn.Module):
    def __init__(self, n_mels, sample_rate, n_stft):
        super().__init__()
        self.register_buffer("fb", melscale_fbanks(n_stft, 0.0, float(sample_rate // 2), n_mels, int(sample_rate)))


class MelSpectrogram(nn.Module):
    """torchaudio.transforms.MelSpectrogram(sample_rate, n_fft, hop_length=, n_mels=) defaults."""

    def __init__(self, sample_rate, n_fft, hop_length=None, n_mels=128):
        super().__init__()
        self.n_fft, self.hop_length, self.n_mels = n_fft, hop_length or n_fft // 2, n_mels
        self.spectrogram = _Spectrogram(n_fft)
        self.mel_scale = _MelScale(n_mels, sample_rate, n_fft // 2 + 1)

    def forward(self, x):
        b, c, t = x.shape
        spec = stft.stft(x.reshape(b * c, t), self.n_fft, self.hop_length, self.n_fft,
                         self.spectrogram.window, mode="pow")            # (b*c, bins, frames)
        mel = ops.conv1d(spec, self.mel_scale.fb.t().unsqueeze(-1).contiguous())   # bins -> mels
        return mel.view(b, c, self.n_mels, mel.shape[-1])


def _iid_span(n, size, mask_param, device):
    """torchaudio mask_along_axis_iid bookkeeping: per row, length ~ U[0, mask_param), start ~ U[0, size - length)."""
    if mask_param > size:                       # torchaudio.functional.mask_along_axis_iid raises here too
        raise ValueError(f"mask_param ({mask_param}) must not be longer than the axis ({size})")
    value = torch.rand(n, device=device) * mask_param
    start = (torch.rand(n, device=device) * (size - value)).long()
    return start.int(), (start + value.long()).int()


def spec_augment(x, freq_mask_param, time_mask_param):
    """FrequencyMasking(freq_mask_param, iid_masks=True) then TimeMasking(time_mask_param, iid_masks=True), mask
    value 0, one span each per (batch, channel) row.  x: (B, C, F, T); returns a masked copy (no parameters
    upstream of the mel spectrogram, so no gradient is needed through the mask)."""
    B, Cc, Fq, T = x.shape
    y = x.detach().clone()
    f0, f1 = _iid_span(B * Cc, Fq, freq_mask_param, x.device)
    t0, t1 = _iid_span(B * Cc, T, time_mask_param, x.device)
    from . import _lib
    from .ops import _ptr, _stream
    _lib.check(_lib.lib().rfx_span_mask(_ptr(y), B * Cc, Fq, T, _ptr(f0), _ptr(f1), _ptr(t0), _ptr(t1), _stream()),
               "rfx_span_mask")
    return y


class _HearClassifier(nn.Module):
    """The HEAR-embedding classifiers of reference remfx/classifier.py:16-128 (PANNs, Wav2CLIP, VGGish, wav2vec2): a FROZEN
    pretrained scene-embedding model behind the HEAR API (`load_model`, `get_scene_embeddings(audio (B, T) at the model's rate)
    -> (B, D)`), fed the clip resampled to the model's rate, and a trainable three-layer MLP head `proj` that emits the
    (B, num_classes) logits FXClassifier trains with cross-entropy (models.py:457-476).

    What runs where: the resampling is the device-side polyphase kernel and the head is three gather-GEMM launches with fused
    ReLU; the embedding model is third-party pretrained code (hearbaseline / wav2clip_hear / panns_hear, weights downloaded by
    those packages) and runs as the package provides it, under no_grad, exactly as upstream.  None of the packages ships in
    this image, so the constructor raises ImportError unless an `embedder` is passed: any object with
    `get_scene_embeddings(audio) -> (B, embed_dim)` (or a plain callable) -- the hook tests use and the place to plug a
    locally available model.  Same constructor arguments and `proj.{0,2,4}.{weight,bias}` state_dict keys as upstream."""

    embed_dim, model_rate, package = 0, 16000, ""

    def __init__(self, num_classes: int, sample_rate: float, hidden_dim: int = 256, embedder=None) -> None:
        super().__init__()
        self.num_classes = num_classes
        self._from_package = embedder is None
        self.model = embedder if embedder is not None else self._load_model()
        if isinstance(self.model, nn.Module):               # frozen upstream by the no_grad around it: say so to the optimiser
            for p in self.model.parameters():               # (torch's AdamW skips parameters without a gradient; the flat
                p.requires_grad_(False)                     # optimiser takes every trainable one)
        from .resample import Resample
        self.resample = Resample(orig_freq=sample_rate, new_freq=self.model_rate)
        self.proj = nn.Sequential(nn.Linear(self.embed_dim, hidden_dim), nn.ReLU(), nn.Linear(hidden_dim, hidden_dim), nn.ReLU(),
                                  nn.Linear(hidden_dim, num_classes))

    def _load_model(self):
        raise NotImplementedError

    def _package(self):
        import importlib
        try:
            return importlib.import_module(self.package)
        except ImportError as e:
            raise ImportError(f"{type(self).__name__} needs the pretrained HEAR embedding package `{self.package}` (reference "
                              "remfx/classifier.py:4-9), which is not installed; install it or pass embedder=<object with "
                              "get_scene_embeddings(audio)>") from e

    def _embed(self, audio):
        if self._from_package:                              # <package>.get_scene_embeddings(audio, model), as upstream
            return self._package().get_scene_embeddings(audio, self.model)
        m = self.model
        return m.get_scene_embeddings(audio) if hasattr(m, "get_scene_embeddings") else m(audio)

    def forward(self, x: torch.Tensor, **kwargs):
        with torch.no_grad():
            x = self.resample(x)
            embed = self._embed(x.reshape(x.shape[0], -1))
        if embed.shape[-1] != self.embed_dim:
            raise ValueError(f"{type(self).__name__}: embedding width {embed.shape[-1]} != {self.embed_dim}")
        h = embed.float()
        for i in (0, 2):
            h = ops.activation(nnops.linear(h, self.proj[i].weight, self.proj[i].bias), "relu")
        return nnops.linear(h, self.proj[4].weight, self.proj[4].bias)


class PANNs(_HearClassifier):                           # classifier.py:16-40
    embed_dim, model_rate, package = 2048, 32000, "panns_hear"

    def _load_model(self):
        return self._package().load_model("hear2021-panns_hear.pth")


class Wav2CLIP(_HearClassifier):                        # classifier.py:43-69
    embed_dim, model_rate, package = 512, 16000, "wav2clip_hear"

    def _load_model(self):
        return self._package().load_model("")


class VGGish(_HearClassifier):                          # classifier.py:72-96
    embed_dim, model_rate, package = 128, 16000, "hearbaseline.vggish"

    def _load_model(self):
        return self._package().load_model()


class wav2vec2(_HearClassifier):                        # classifier.py:99-124
    embed_dim, model_rate, package = 1024, 16000, "hearbaseline.wav2vec2"

    def _load_model(self):
        return self._package().load_model()


class ConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv1 = nn.Conv2d(in_channels, out_channels, (3, 3), (1, 1), (1, 1), bias=False)
        self.conv2 = nn.Conv2d(out_channels, out_channels, (3, 3), (1, 1), (1, 1), bias=False)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.bn2 = nn.BatchNorm2d(out_channels)
        self.init_weight()

    def init_weight(self):
        init_layer(self.conv1); init_layer(self.conv2); init_bn(self.bn1); init_bn(self.bn2)

    def forward(self, input, pool_size=(2, 2), pool_type="avg"):
        if pool_type != "avg":
            raise ValueError("Cnn14 uses avg pooling only (classifier.py:209-220)")
        x = ops.conv2d(input, self.conv1.weight, None, (1, 1), (1, 1))
        x = nnops.batch_norm(x, self.bn1, self.bn1.training, relu=True)
        x = ops.conv2d(x, self.conv2.weight, None, (1, 1), (1, 1))
        x = nnops.batch_norm(x, self.bn2, self.bn2.training, relu=True)
        return nnops.avg_pool2d(x, pool_size)


class Cnn14(nn.Module):
    def __init__(self, num_classes: int, sample_rate: float, model_sample_rate: float, n_fft: int = 1024,
                 hop_length: int = 256, n_mels: int = 128, specaugment: bool = False):
        super().__init__()
        self.num_classes, self.n_fft, self.hop_length = num_classes, n_fft, hop_length
        self.sample_rate, self.model_sample_rate, self.specaugment = sample_rate, model_sample_rate, specaugment
        self.register_buffer("window", torch.hann_window(n_fft))
        self.melspec = MelSpectrogram(model_sample_rate, n_fft, hop_length=hop_length, n_mels=n_mels)
        self.bn0 = nn.BatchNorm2d(n_mels)
        widths = (64, 128, 256, 512, 1024, 2048)
        cin = 1
        for i, w in enumerate(widths, 1):
            setattr(self, f"conv_block{i}", ConvBlock(cin, w))
            cin = w
        self.fc1 = nn.Linear(2048, 2048, bias=True)
        self.heads = nn.ModuleList([nn.Linear(2048, 1, bias=True) for _ in range(num_classes)])
        self.init_weight()
        # sample_rate != model_sample_rate: polyphase resampling on the device (classifier.py:180-183); the transform's
        # filter bank is a persistent buffer upstream (state_dict key `resample.kernel`), so it is one here
        if sample_rate != model_sample_rate:
            from .resample import Resample
            self.resample = Resample(orig_freq=sample_rate, new_freq=model_sample_rate)
        # specaugment: iid frequency / time span masks in training only (classifier.py:185-187, 198-204)
        self.freq_mask_param, self.time_mask_param = 64, 128

    def init_weight(self):
        init_bn(self.bn0)
        init_layer(self.fc1)

    def forward(self, x: torch.Tensor, train: bool = False):
        """train=False (detection / validation / test, reference models.py:60-64, 520-560): the network runs at fp32 parity
        even under trainer.precision=bf16-mixed, so the thresholded labels are those of the fp32 reference."""
        if train:
            return self._forward(x, True)
        with ops.at_least_fp32_parity():
            return self._forward(x, False)

    def _forward(self, x: torch.Tensor, train: bool):
        if self.sample_rate != self.model_sample_rate:
            x = self.resample(x)
        x = self.melspec(x)                                                     # (B, 1, n_mels, frames)
        if self.specaugment and train:
            x = spec_augment(x, self.freq_mask_param, self.time_mask_param)
        # per-clip standardisation, unbiased std, no epsilon (classifier.py:207)
        x = (x - x.mean(dim=(2, 3), keepdim=True)) / x.std(dim=(2, 3), keepdim=True)
        for i in range(1, 7):
            x = getattr(self, f"conv_block{i}")(x, pool_size=(2, 2) if i < 6 else (1, 1), pool_type="avg")
            x = nnops.dropout(x, 0.2, train)
        x = torch.mean(x, dim=3)
        (x1, _) = torch.max(x, dim=2)
        x = x1 + torch.mean(x, dim=2)
        x = nnops.dropout(x, 0.5, train)
        x = ops.activation(nnops.linear(x, self.fc1.weight, self.fc1.bias), "relu")
        w = torch.cat([h.weight for h in self.heads], 0)                         # (num_classes, 2048)
        b = torch.cat([h.bias for h in self.heads], 0)
        out = ops.activation(nnops.linear(x, w, b), "sigmoid")                   # (B, num_classes)
        return [out[:, k:k + 1] for k in range(self.num_classes)]

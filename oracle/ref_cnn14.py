"""Oracle (test infrastructure): PANNs-style Cnn14 effect classifier.

Restates /root/reference/remfx/classifier.py:134-284 (Cnn14, ConvBlock) with
functional torch CPU ops.  The conv stack after the mel front end is PINNED by
tests/golden/cnn14_small.npz (imported reference module, same weights).

The mel front end is ``torchaudio.transforms.MelSpectrogram`` (classifier.py:
156-161), an un-vendored dependency absent from this image -> PARITY UNPINNED
for ``mel_spectrogram``; it restates the published algorithm (SURVEY.md A.5:
power-2 STFT, HTK mel scale, norm=None triangles).

state_dict keys follow the reference: conv_block{1..6}.{conv1,conv2}.weight,
.bn{1,2}.{weight,bias,running_mean,running_var}, fc1.{weight,bias},
heads.{k}.{weight,bias}.
"""
import math

import torch
import torch.nn.functional as F


def melscale_fbanks(n_freqs, f_min, f_max, n_mels, sample_rate):
    """HTK mel triangles, norm=None -> (n_freqs, n_mels)."""
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


def mel_spectrogram(x, sample_rate, n_fft, hop_length, n_mels):
    """(B, 1, T) -> (B, 1, n_mels, frames); power spectrogram @ mel fb, no log."""
    b, c, t = x.shape
    X = torch.stft(x.reshape(b * c, t), n_fft, hop_length, n_fft,
                   torch.hann_window(n_fft), center=True, pad_mode="reflect",
                   return_complex=True)
    spec = (X.real ** 2 + X.imag ** 2).reshape(b, c, X.shape[-2], X.shape[-1])
    fb = melscale_fbanks(n_fft // 2 + 1, 0.0, sample_rate / 2.0, n_mels, int(sample_rate))
    return torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)


def _bn(x, sd, p, train, eps=1e-5):
    if train:   # batch statistics (biased var), as nn.BatchNorm2d in train mode
        return F.batch_norm(x, None, None, sd[p + "weight"], sd[p + "bias"], True, 0.1, eps)
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"],
                        sd[p + "weight"], sd[p + "bias"], False, 0.1, eps)


def conv_block(x, sd, p, pool, bn_train=False):
    # classifier.py:269-276
    x = F.relu(_bn(F.conv2d(x, sd[p + "conv1.weight"], padding=1), sd, p + "bn1.", bn_train))
    x = F.relu(_bn(F.conv2d(x, sd[p + "conv2.weight"], padding=1), sd, p + "bn2.", bn_train))
    return F.avg_pool2d(x, kernel_size=pool)


def cnn14_from_mel(mel, sd, num_classes=5, bn_train=False):
    """mel: (B,1,n_mels,frames) -> list of num_classes (B,1) sigmoids.
    classifier.py:207-233 (eval path: dropout off)."""
    x = (mel - mel.mean(dim=(2, 3), keepdim=True)) / mel.std(dim=(2, 3), keepdim=True)
    for i in range(1, 7):
        x = conv_block(x, sd, f"conv_block{i}.", (2, 2) if i < 6 else (1, 1), bn_train)
    x = x.mean(dim=3)
    x = x.max(dim=2)[0] + x.mean(dim=2)
    x = F.relu(F.linear(x, sd["fc1.weight"], sd["fc1.bias"]))
    return [torch.sigmoid(F.linear(x, sd[f"heads.{k}.weight"], sd[f"heads.{k}.bias"]))
            for k in range(num_classes)]


def cnn14_forward(x, sd, sample_rate=48000, n_fft=2048, hop_length=512, n_mels=128,
                  num_classes=5, bn_train=False):
    return cnn14_from_mel(mel_spectrogram(x, sample_rate, n_fft, hop_length, n_mels), sd,
                          num_classes, bn_train)


def cnn14_init_state_dict(widths=(64, 128, 256, 512, 1024, 2048), num_classes=5, seed=0):
    """xavier-uniform conv/fc weights, zero bias, BN weight 1 / bias 0
    (utils.py:162-174; classifier.py:189-191,263-267); heads use torch's default
    Linear init scale.  Deterministic generator, not the reference RNG stream."""
    g = torch.Generator().manual_seed(seed)

    def xavier(*shape):
        rf = 1
        for s in shape[2:]:
            rf *= s
        a = math.sqrt(6.0 / (shape[1] * rf + shape[0] * rf))
        return (torch.rand(*shape, generator=g) * 2 - 1) * a

    sd, cin = {}, 1
    for i, w in enumerate(widths, 1):
        p = f"conv_block{i}."
        sd[p + "conv1.weight"] = xavier(w, cin, 3, 3)
        sd[p + "conv2.weight"] = xavier(w, w, 3, 3)
        for bn in ("bn1.", "bn2."):
            sd[p + bn + "weight"] = torch.ones(w)
            sd[p + bn + "bias"] = torch.zeros(w)
            sd[p + bn + "running_mean"] = torch.zeros(w)
            sd[p + bn + "running_var"] = torch.ones(w)
        cin = w
    sd["fc1.weight"] = xavier(cin, cin)
    sd["fc1.bias"] = torch.zeros(cin)
    b = 1.0 / math.sqrt(cin)
    for k in range(num_classes):
        sd[f"heads.{k}.weight"] = (torch.rand(1, cin, generator=g) * 2 - 1) * b
        sd[f"heads.{k}.bias"] = (torch.rand(1, generator=g) * 2 - 1) * b
    return sd

"""Oracle (test infrastructure): spectrogram and crops.

Restates /root/reference/remfx/utils.py:138-159 (spectrogram) and
utils.py:202-211 (center_crop, causal_crop).  PINNED by tests/golden/utils_*.npz.
"""
import torch


def spectrogram(x, window, n_fft, hop_length, alpha):
    # utils.py:145-159: fold channels into batch, torch.stft (center/reflect
    # defaults), (|X| + 1e-8) ** alpha.
    b, c, t = x.shape
    X = torch.stft(x.reshape(b * c, t), n_fft=n_fft, hop_length=hop_length,
                   window=window, return_complex=True)
    X = X.reshape(b, c, X.shape[-2], X.shape[-1])
    return torch.pow(X.abs() + 1e-8, alpha)


def center_crop(x, length):
    # utils.py:202-205
    start = (x.shape[-1] - length) // 2
    return x[..., start:start + length]


def causal_crop(x, length):
    # utils.py:208-211 -- note: drops the final sample
    stop = x.shape[-1] - 1
    return x[..., stop - length:stop]

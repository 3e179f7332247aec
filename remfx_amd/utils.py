"""Host-side mirror of remfx.utils for the hot path (reference remfx/utils.py).

center_crop / causal_crop are views (utils.py:202-211; causal_crop drops the last
sample).  spectrogram() runs on the HIP framed-FFT kernel.
"""
import torch
import torch.nn as nn


def center_crop(x, length: int):
    start = (x.shape[-1] - length) // 2
    return x[..., start:start + length]


def causal_crop(x, length: int):
    stop = x.shape[-1] - 1
    return x[..., stop - length:stop]


def crop_start(fn_is_causal: bool, in_len: int, out_len: int) -> int:
    """First kept index of center_crop / causal_crop."""
    return (in_len - 1 - out_len) if fn_is_causal else (in_len - out_len) // 2


def init_layer(layer):
    """utils.py:162-168"""
    nn.init.xavier_uniform_(layer.weight)
    if getattr(layer, "bias", None) is not None:
        layer.bias.data.fill_(0.0)


def init_bn(bn):
    """utils.py:171-174"""
    bn.bias.data.fill_(0.0)
    bn.weight.data.fill_(1.0)


def spectrogram(x, window, n_fft: int, hop_length: int, alpha: float):
    """utils.py:138-159: (|STFT(x)| + 1e-8) ** alpha, (B, C, T) -> (B, C, bins, frames)."""
    from . import stft
    b, c, t = x.shape
    X = stft.stft(x.reshape(b * c, t), n_fft, hop_length, n_fft, window, mode="magpow", eps=1e-8,
                  alpha=alpha)
    return X.reshape(b, c, X.shape[-2], X.shape[-1])

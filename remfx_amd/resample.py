"""Polyphase windowed-sinc sample-rate conversion on the gather-GEMM kernels.

Restates torchaudio's `sinc_interp_hann` resampler (lowpass_filter_width 6, rolloff 0.99) -- the one behind
`torchaudio.transforms.Resample` (reference scripts/remfx_detect.py:44-50, remfx/classifier.py:180-183) and
`torchaudio.functional.resample` (remfx/datasets.py:604-606).  torchaudio is not in the reference tree:
SURVEY 8(f) rank 1 / App. A, parity unpinned; the CPU restatement is oracle/ref_resample.py.

After dividing both rates by their gcd, output sample  o = new*i + p  (phase p of frame i) is
    y[o] = sum_k h[p][k] * x[orig*i + k - width],   k = 0 .. 2*width + orig - 1,
i.e. a strided 1-D convolution with `new` output channels: ONE gather-GEMM launch (M = new, K = 2*width + orig,
stride orig) whose store interleaves the phases through the output strides -- no transposed copy afterwards.
The filter table (at most a few hundred taps per phase) is built on the host in float64, like the transform does
at construction time.
"""
import math

import torch

from . import ops


def sinc_kernel(orig_freq, new_freq, lowpass_filter_width=6, rolloff=0.99, dtype=torch.float64):
    """(new, 2*width + orig) filter bank and `width` for gcd-reduced rates.  dtype float64 = the Resample transform
    (table computed in double, stored in float32); float32 = functional.resample on a float32 waveform."""
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base_freq = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base_freq)
    idx = torch.arange(-width, width + orig, dtype=dtype)[None, :] / orig
    t = torch.arange(0, -new, -1, dtype=dtype)[:, None] / new + idx
    t = (t * base_freq).clamp_(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    scale = base_freq / orig
    kern = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * scale
    return kern.to(torch.float32), width, orig, new


_TABLES = {}


def resample(x, orig_freq, new_freq, table_dtype=torch.float64, kernel=None):
    """x: (..., L) fp32 on the GPU -> (..., ceil(new * L / orig)).  Identity when the rates are equal.
    kernel: the (new, 1, K) filter bank to use instead of the cached one (the `Resample` module's buffer)."""
    if int(orig_freq) == int(new_freq):
        return x
    ops._req(x, "waveform")
    key = (int(orig_freq), int(new_freq), table_dtype, str(x.device))
    if key not in _TABLES:
        kern, width, orig, new = sinc_kernel(orig_freq, new_freq, dtype=table_dtype)
        _TABLES[key] = (kern.to(x.device).unsqueeze(1).contiguous(), width, orig, new)     # (new, 1, K)
    w, width, orig, new = _TABLES[key]
    if kernel is not None:
        if kernel.shape != w.shape:
            raise ValueError(f"resample kernel {tuple(kernel.shape)} does not fit {orig_freq} -> {new_freq} Hz {tuple(w.shape)}")
        w = kernel.to(x.device, torch.float32).contiguous()
    shape = x.shape
    L = shape[-1]
    rows = x.reshape(-1, 1, L)
    xp = torch.zeros((rows.shape[0], 1, L + 2 * width + orig), device=x.device, dtype=torch.float32)
    xp[..., width:width + L].copy_(rows)                      # zero padding (width, width + orig), as upstream
    frames = (xp.shape[-1] - w.shape[-1]) // orig + 1
    buf = torch.empty((rows.shape[0], frames, new), device=x.device, dtype=torch.float32)
    out = buf.permute(0, 2, 1).unsqueeze(2)                   # (R, new, 1, frames) view: phase p of frame i at i*new + p
    ops.conv2d_forward(xp.unsqueeze(2), w.unsqueeze(2), None, (1, orig), (0, 0), (1, 1), out=out)
    target = math.ceil(new * L / orig)
    return buf.reshape(rows.shape[0], frames * new)[:, :target].reshape(*shape[:-1], target)


class Resample(torch.nn.Module):
    """torchaudio.transforms.Resample(orig_freq, new_freq) (classifier.py:24-26, 50-52, 75-77, 101-103, 180-183): holds the
    filter bank as the persistent buffer ``kernel`` of shape (new / gcd, 1, 2 * width + orig / gcd) when the rates differ, as
    upstream, so checkpoints of resampling classifiers (``network.resample.kernel``) load strictly; forward resamples on
    the device with THAT buffer."""

    def __init__(self, orig_freq=16000, new_freq=16000):
        super().__init__()
        self.orig_freq, self.new_freq = int(orig_freq), int(new_freq)
        if self.orig_freq != self.new_freq:
            kern, self.width, _, _ = sinc_kernel(self.orig_freq, self.new_freq, dtype=torch.float64)
            self.register_buffer("kernel", kern.unsqueeze(1).contiguous())

    def forward(self, x):
        if self.orig_freq == self.new_freq:
            return x
        return resample(x, self.orig_freq, self.new_freq, kernel=self.kernel)

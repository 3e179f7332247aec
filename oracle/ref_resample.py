"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): CPU restatement of torchaudio's windowed-sinc resampler
(`sinc_interp_hann`, lowpass_filter_width 6, rolloff 0.99), the algorithm behind
  * torchaudio.transforms.Resample          -- reference scripts/remfx_detect.py:44-50, remfx/classifier.py:180-183
  * torchaudio.functional.resample          -- reference remfx/datasets.py:604-606 (InferenceDataset)
torchaudio is not in the reference tree nor in this image: **parity unpinned** (SURVEY 8c), restated from the published
algorithm; pinned here only by the analytic property tests in tests/test_oracle_golden.py (band-limited sinusoids are
reproduced at the new rate, equal rates are the identity, output length ceil(new * L / orig)).

Written as the textbook polyphase form (one dot product per output sample) rather than as the strided convolution the
product path uses, so that the two do not share structure."""
import math

import torch


def _filter(orig, new, lowpass_filter_width=6, rolloff=0.99, dtype=torch.float64):
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    k = torch.arange(-width, width + orig, dtype=dtype)
    phases = []
    for p in range(new):
        t = (-p / new + k / orig) * base
        t = t.clamp(-lowpass_filter_width, lowpass_filter_width)
        win = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
        tp = t * math.pi
        sinc = torch.where(tp == 0, torch.ones_like(tp), torch.sin(tp) / tp)
        phases.append(sinc * win * (base / orig))
    return torch.stack(phases).to(torch.float32), width


def resample(x, orig_freq, new_freq, table_dtype=torch.float64):
    """x: (..., L) -> (..., ceil(new * L / orig)); table_dtype float64 = transforms.Resample, float32 = functional.resample."""
    orig_freq, new_freq = int(orig_freq), int(new_freq)
    if orig_freq == new_freq:
        return x
    g = math.gcd(orig_freq, new_freq)
    orig, new = orig_freq // g, new_freq // g
    h, width = _filter(orig, new, dtype=table_dtype)
    shape = x.shape
    L = shape[-1]
    rows = x.reshape(-1, L).to(torch.float32)
    target = math.ceil(new * L / orig)
    xp = torch.nn.functional.pad(rows, (width, width + orig))
    K = h.shape[1]
    out = torch.zeros(rows.shape[0], target)
    for o in range(target):                      # output sample o = new * i + p reads x[orig * i - width + k]
        i, p = divmod(o, new)
        out[:, o] = (xp[:, orig * i: orig * i + K] * h[p]).sum(-1)
    return out.reshape(*shape[:-1], target)

"""Audio effects of the data side, rendered ON THE DEVICE (reference remfx/effects.py:297-616, 619-629, 699-707).

The reference renders its five training effects on the CPU with pedalboard (JUCE C++) and normalises loudness
with pyloudnorm while it builds the dataset (datasets.py:109-202) or augments on the fly (datasets.py:205-330).
Here the same classes -- same names (they are the dict keys of RemFXChainInference.model and of cfg ``ckpts:`` /
``inference_effects_ordering``, models.py:81,96), same constructor arguments (``cfg/effects/all.yaml`` instantiates
unchanged), same parameter ranges, same random draws in the same order (``rand`` = torch.rand(1), ``loguniform`` =
scipy) -- render through the HIP kernels of csrc/fx.hip, one launch per effect for a whole batch of clips with
per-clip parameters.  ``forward(x)`` takes the reference's ``(channels, samples)`` tensor or a batch
``(B, 1, samples)``; tensors must live on the GPU (no CPU path: the library raises otherwise).

Algorithms are restatements of the published JUCE / pedalboard / pyloudnorm code (oracle/ref_effects.py lists
them; both packages are absent here, parity unpinned).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import check


def loguniform(low=0, high=1):                         # effects.py:25-26
    import scipy.stats
    return scipy.stats.loguniform.rvs(low, high)


def rand(low=0, high=1):                               # effects.py:29-30
    return (torch.rand(1).numpy()[0] * (high - low)) + low


def randint(low=0, high=1):                            # effects.py:33-34
    return torch.randint(low, high + 1, (1,)).numpy()[0]


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _as_clips(x):
    """(channels, T) or (B, C, T) -> contiguous (N, T) fp32 device view + a function restoring the shape."""
    if not (isinstance(x, torch.Tensor) and x.is_cuda):
        raise ValueError("remfx_amd.effects render on the GPU: pass a CUDA tensor (there is no CPU path)")
    shape = x.shape
    if x.dim() not in (2, 3):
        raise ValueError(f"effects take (channels, samples) or (batch, channels, samples), got {tuple(shape)}")
    flat = x.reshape(-1, shape[-1]).to(torch.float32).contiguous()
    return flat, (lambda y: y.view(shape))


def _vec(vals, device, dtype=torch.float32):
    return torch.tensor(np.asarray(vals), dtype=dtype).to(device)


class _RandomEffect(torch.nn.Module):
    """Keeps every ``min_* / max_*`` range of the reference constructor.  ``draw()`` samples ONE parameter set with the
    reference's calls in the reference's order; ``forward`` draws one set per clip of a batch (one set for all channels of
    a 2-D input, as pedalboard applies one board to all channels) and renders."""

    defaults = {}

    def __init__(self, sample_rate: float, **ranges):
        super().__init__()
        unknown = set(ranges) - set(self.defaults)
        if unknown:                                   # same failure mode as a wrong kwarg upstream
            raise TypeError(f"{type(self).__name__}.__init__() got unexpected keyword arguments {sorted(unknown)}")
        self.sample_rate = sample_rate
        self.ranges = dict(self.defaults)
        self.ranges.update(ranges)
        for k, v in self.ranges.items():
            setattr(self, k, v)

    def draw(self):
        raise NotImplementedError

    def render(self, clips, params):
        """clips: (N, T) device tensor; params: list of N dicts (draw()).  Returns (N, T)."""
        raise NotImplementedError

    def forward(self, x: torch.Tensor):
        clips, restore = _as_clips(x)
        nsets = x.shape[0] if x.dim() == 3 else 1
        sets = [self.draw() for _ in range(nsets)]
        per = clips.shape[0] // nsets
        params = [s for s in sets for _ in range(per)]
        self.last_params = sets
        return restore(self.render(clips, params))


class RandomPedalboardReverb(_RandomEffect):
    defaults = dict(min_room_size=0.0, max_room_size=1.0, min_damping=0.0, max_damping=1.0, min_wet_dry=0.0,
                    max_wet_dry=0.7, min_width=0.0, max_width=1.0)

    def draw(self):                                    # effects.py:579-583
        return dict(room_size=rand(self.min_room_size, self.max_room_size), damping=rand(self.min_damping, self.max_damping),
                    wet_dry=rand(self.min_wet_dry, self.max_wet_dry), width=rand(self.min_width, self.max_width))

    def render(self, clips, params):
        dev = clips.device
        damp = _vec([p["damping"] * 0.4 for p in params], dev)                       # juce::Reverb::setParameters
        fb = _vec([p["room_size"] * 0.28 + 0.7 for p in params], dev)
        wet1 = _vec([0.5 * (p["wet_dry"] * 3.0) * (1.0 + p["width"]) for p in params], dev)
        dry = _vec([(1.0 - p["wet_dry"]) * 2.0 for p in params], dev)
        y = torch.empty_like(clips)
        check(_lib.lib().rfx_fx_reverb(_ptr(clips), _ptr(y), clips.shape[0], clips.shape[1], int(self.sample_rate), _ptr(damp),
                                       _ptr(fb), _ptr(wet1), _ptr(dry), _stream()), "rfx_fx_reverb")
        return y


class RandomPedalboardChorus(_RandomEffect):
    defaults = dict(min_rate_hz=0.25, max_rate_hz=4.0, min_depth=0.0, max_depth=0.6, min_centre_delay_ms=5.0,
                    max_centre_delay_ms=10.0, min_feedback=0.1, max_feedback=0.6, min_mix=0.1, max_mix=0.7)

    def draw(self):                                    # effects.py:398-402
        return dict(rate_hz=rand(self.min_rate_hz, self.max_rate_hz), depth=rand(self.min_depth, self.max_depth),
                    centre_delay_ms=rand(self.min_centre_delay_ms, self.max_centre_delay_ms),
                    feedback=rand(self.min_feedback, self.max_feedback), mix=rand(self.min_mix, self.max_mix))

    def render(self, clips, params):
        dev = clips.device
        v = {k: _vec([p[k] for p in params], dev) for k in ("rate_hz", "depth", "centre_delay_ms", "feedback", "mix")}
        y = torch.empty_like(clips)
        check(_lib.lib().rfx_fx_chorus(_ptr(clips), _ptr(y), clips.shape[0], clips.shape[1], float(self.sample_rate),
                                       _ptr(v["rate_hz"]), _ptr(v["depth"]), _ptr(v["centre_delay_ms"]), _ptr(v["feedback"]),
                                       _ptr(v["mix"]), _stream()), "rfx_fx_chorus")
        return y


class RandomPedalboardDelay(_RandomEffect):
    # `max_delay_sconds` is the reference's own spelling (effects.py:346; cfg/effects/all.yaml)
    defaults = dict(min_delay_seconds=0.1, max_delay_sconds=1.0, min_feedback=0.05, max_feedback=0.6, min_mix=0.0,
                    max_mix=0.7)

    def draw(self):                                    # effects.py:362-364
        return dict(delay_seconds=loguniform(self.min_delay_seconds, self.max_delay_sconds),
                    feedback=rand(self.min_feedback, self.max_feedback), mix=rand(self.min_mix, self.max_mix))

    def render(self, clips, params):
        dev = clips.device
        d = _vec([int(p["delay_seconds"] * self.sample_rate) for p in params], dev, torch.int32)
        fb, mix = _vec([p["feedback"] for p in params], dev), _vec([p["mix"] for p in params], dev)
        y = torch.empty_like(clips)
        check(_lib.lib().rfx_fx_delay(_ptr(clips), _ptr(y), clips.shape[0], clips.shape[1], _ptr(d), _ptr(fb), _ptr(mix),
                                      _stream()), "rfx_fx_delay")
        return y


class RandomPedalboardDistortion(_RandomEffect):
    defaults = dict(min_drive_db=-20.0, max_drive_db=12.0)

    def draw(self):                                    # effects.py:495
        return dict(drive_db=rand(self.min_drive_db, self.max_drive_db))

    def render(self, clips, params):
        g = _vec([10.0 ** (p["drive_db"] / 20.0) for p in params], clips.device)
        y = torch.empty_like(clips)
        check(_lib.lib().rfx_fx_distortion(_ptr(clips), _ptr(y), clips.shape[0], clips.shape[1], _ptr(g), _stream()),
              "rfx_fx_distortion")
        return y


class RandomPedalboardCompressor(_RandomEffect):
    defaults = dict(min_threshold_db=-42.0, max_threshold_db=-6.0, min_ratio=1.5, max_ratio=4.0, min_attack_ms=1.0,
                    max_attack_ms=50.0, min_release_ms=10.0, max_release_ms=250.0)

    def draw(self):                                    # effects.py:323-326
        return dict(threshold_db=rand(self.min_threshold_db, self.max_threshold_db), ratio=rand(self.min_ratio, self.max_ratio),
                    attack_ms=rand(self.min_attack_ms, self.max_attack_ms), release_ms=rand(self.min_release_ms, self.max_release_ms))

    def render(self, clips, params):
        dev = clips.device
        ef = -2.0 * math.pi * 1000.0 / float(self.sample_rate)                       # juce::dsp::BallisticsFilter
        cte = lambda ms: 0.0 if ms < 1e-3 else math.exp(ef / ms)
        thr = _vec([10.0 ** (p["threshold_db"] / 20.0) for p in params], dev)
        ratio = _vec([p["ratio"] for p in params], dev)
        ca, cr = _vec([cte(p["attack_ms"]) for p in params], dev), _vec([cte(p["release_ms"]) for p in params], dev)
        y, ws = torch.empty_like(clips), torch.empty_like(clips)
        check(_lib.lib().rfx_fx_compressor(_ptr(clips), _ptr(y), _ptr(ws), clips.shape[0], clips.shape[1], _ptr(thr), _ptr(ratio),
                                           _ptr(ca), _ptr(cr), _stream()), "rfx_fx_compressor")
        return y


def _k_weighting(rate):
    """pyloudnorm "K-weighting": high shelf (+4 dB, 1500 Hz, Q 1/sqrt2) then high pass (38 Hz, Q 0.5), normalised by a0."""
    def coef(G, Q, fc, shelf):
        A = 10.0 ** (G / 40.0)
        w0 = 2.0 * np.pi * (fc / rate)
        alpha = np.sin(w0) / (2.0 * Q)
        c = np.cos(w0)
        if shelf:
            b = [A * ((A + 1) + (A - 1) * c + 2 * np.sqrt(A) * alpha), -2 * A * ((A - 1) + (A + 1) * c),
                 A * ((A + 1) + (A - 1) * c - 2 * np.sqrt(A) * alpha)]
            a = [(A + 1) - (A - 1) * c + 2 * np.sqrt(A) * alpha, 2 * ((A - 1) - (A + 1) * c),
                 (A + 1) - (A - 1) * c - 2 * np.sqrt(A) * alpha]
        else:
            b = [(1 + c) / 2, -(1 + c), (1 + c) / 2]
            a = [1 + alpha, -2 * c, 1 - alpha]
        return np.array(b, dtype=np.float64) / a[0], np.array(a, dtype=np.float64) / a[0]
    return coef(4.0, 1.0 / np.sqrt(2.0), 1500.0, True), coef(0.0, 0.5, 38.0, False)


def _transition(b1, a1, b2, a2, steps):
    """Zero-input state transition over `steps` samples of the two cascaded biquads in transposed direct form II
    (states s1a, s1b, s2a, s2b): y1 = s1a, s1a' = -a1[1] y1 + s1b, s1b' = -a1[2] y1; y2 = b2[0] y1 + s2a, ..."""
    A = np.zeros((4, 4))
    A[0, 0], A[0, 1], A[1, 0] = -a1[1], 1.0, -a1[2]
    # y2 = b2[0] * s1a + s2a
    A[2, 0], A[2, 2], A[2, 3] = b2[1] - a2[1] * b2[0], -a2[1], 1.0
    A[3, 0], A[3, 2] = b2[2] - a2[2] * b2[0], -a2[2]
    return np.linalg.matrix_power(A, int(steps))


class LoudnessNormalize(torch.nn.Module):
    """effects.py:619-629: scale to `target_lufs_db` by the BS.1770 integrated loudness (pyloudnorm.Meter)."""

    def __init__(self, sample_rate: float, target_lufs_db: float = -32.0) -> None:
        super().__init__()
        self.sample_rate, self.target_lufs_db = sample_rate, target_lufs_db
        self._cache = {}

    def _plan(self, T):
        p = self._cache.get(T)
        if p is None:
            rate = self.sample_rate
            T_g, step = 0.4, 0.25
            if T < T_g * rate:
                raise ValueError("Audio must have length greater than the block size.")      # pyloudnorm's own error
            nblk = int(np.round(((T / rate - T_g) / (T_g * step))) + 1)
            hop = int(round(T_g * step * rate))
            for j in range(nblk):        # the kernel sums 100 ms hops: pyloudnorm's block bounds must be hop multiples
                if int(T_g * (j * step) * rate) != j * hop or int(T_g * (j * step + 1) * rate) != (j + 4) * hop:
                    raise NotImplementedError(f"loudness blocks are not multiples of a 100 ms hop at {rate} Hz")
            chunk = -(-T // 64)
            (b1, a1), (b2, a2) = _k_weighting(rate)
            M = _transition(b1, a1, b2, a2, chunk)
            coef = np.concatenate([b1, a1, b2, a2, M.reshape(-1)]).astype(np.float64)
            p = dict(nblk=nblk, hop=hop, nhop=nblk + 3, chunk=chunk, coef=coef, inv=1.0 / (T_g * rate))
            self._cache[T] = p
        return p

    def measure(self, clips):
        """(N, T) device clips -> (lufs (N,), gain (N,)) device tensors."""
        N, T = clips.shape
        p = self._plan(T)
        hop_ws = torch.empty((N, p["nhop"]), device=clips.device, dtype=torch.float64)
        lufs = torch.empty(N, device=clips.device, dtype=torch.float32)
        gain = torch.empty_like(lufs)
        coef = p["coef"]
        check(_lib.lib().rfx_fx_loudness(_ptr(clips), N, T, p["chunk"], p["hop"], p["nhop"], p["nblk"], p["inv"],
                                         coef.ctypes.data_as(C.c_void_p), float(self.target_lufs_db), _ptr(hop_ws), _ptr(lufs),
                                         _ptr(gain), _stream()), "rfx_fx_loudness")
        return lufs, gain

    def forward(self, x: torch.Tensor):
        clips, restore = _as_clips(x)
        if x.dim() == 2 and x.shape[0] != 1:
            raise NotImplementedError("LoudnessNormalize: mono clips (the reference sums to mono before its effects)")
        _, gain = self.measure(clips)
        y = torch.empty_like(clips)
        check(_lib.lib().rfx_fx_scale(_ptr(clips), _ptr(y), clips.shape[0], clips.shape[1], _ptr(gain), _stream()), "rfx_fx_scale")
        return restore(y)


# label order: column k of dry / wet label tensors (effects.py:699-707)
Pedalboard_Effects = [
    RandomPedalboardReverb,
    RandomPedalboardChorus,
    RandomPedalboardDelay,
    RandomPedalboardDistortion,
    RandomPedalboardCompressor,
]

"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the effect-rendering side, SURVEY 8(f) rank 3.

PARITY UNPINNED: the reference renders these effects with `pedalboard` (JUCE DSP, C++) and measures
loudness with `pyloudnorm` (remfx/effects.py:10-20, 297-629); neither package nor its source is in
/root/reference or in this image.  The functions below restate the published algorithms in plain numpy
float64, one sample at a time where the algorithm is recursive -- the same restatement csrc/fx.hip
implements with blocked / scanned recurrences in fp32, so the GPU tests check the PARALLELISATION and the
arithmetic, not fidelity to pedalboard.  Recalled sources (versions the reference's setup.py leaves open):
  * pedalboard.Distortion  = juce::dsp::Gain(drive_db) -> juce::dsp::WaveShaper(std::tanh)
  * pedalboard.Delay       = juce::dsp::DelayLine, delay (int)(seconds * sr), pop -> push(x + fb * popped),
                              y = (1 - mix) x + mix * popped
  * pedalboard.Chorus      = juce::dsp::Chorus (sine LFO from phase -pi, depth / 2, 20 ms modulation span, 1 ms
                              floor, linear-interpolated delay line, one-sample feedback, linear dry/wet);
                              parameter smoothing ramps (50 ms) are NOT modelled
  * pedalboard.Compressor  = juce::dsp::Compressor (peak BallisticsFilter, cte = exp(-2 pi 1000 / (sr ms)))
  * pedalboard.Reverb      = juce::Reverb (Freeverb: 8 combs + 4 all-passes, tunings scaled by sr / 44100,
                              damp * 0.4, room * 0.28 + 0.7, wet * 3, dry * 2, input gain 0.015), mono path
  * pyloudnorm.Meter       = ITU-R BS.1770-4 K-weighting (high shelf 1500 Hz +4 dB Q 1/sqrt2, high pass 38 Hz
                              Q 0.5), 400 ms blocks at 75 % overlap, -70 LUFS absolute / -10 LU relative gates
Reference call sites: remfx/effects.py:297-616, 619-629; remfx/datasets.py:109-202, 205-330.
"""
import numpy as np
import scipy.signal


def distortion(x, drive_db):
    return np.tanh(x.astype(np.float64) * 10.0 ** (drive_db / 20.0))


def delay(x, sample_rate, delay_seconds, feedback, mix):
    x = x.astype(np.float64)
    D = int(delay_seconds * sample_rate)
    w = np.zeros_like(x)
    y = np.empty_like(x)
    for n in range(x.shape[-1]):
        d = w[..., n - D] if (D > 0 and n >= D) else 0.0
        w[..., n] = x[..., n] + feedback * d
        y[..., n] = (1.0 - mix) * x[..., n] + mix * d
    return y


def chorus(x, sample_rate, rate_hz, depth, centre_delay_ms, feedback, mix):
    x = x.astype(np.float64)
    T = x.shape[-1]
    pushed = np.zeros(T)
    y = np.empty(T)
    last = 0.0
    for n in range(T):
        lfo = np.sin(2.0 * np.pi * rate_hz * n / sample_rate - np.pi) * depth * 0.5
        d = max(1.0, 20.0 * lfo + centre_delay_ms) * sample_rate / 1000.0
        di = int(d)
        fr = d - di
        pushed[n] = x[n] - last
        i1, i2 = n - di, n - di - 1
        v1 = pushed[i1] if i1 >= 0 else 0.0
        v2 = pushed[i2] if i2 >= 0 else 0.0
        pop = v1 + fr * (v2 - v1)
        y[n] = (1.0 - mix) * x[n] + mix * pop
        last = pop * feedback
    return y


def compressor(x, sample_rate, threshold_db, ratio, attack_ms, release_ms):
    x = x.astype(np.float64)
    thr = 10.0 ** (threshold_db / 20.0) if threshold_db > -200.0 else 0.0
    ef = -2.0 * np.pi * 1000.0 / sample_rate
    ca = 0.0 if attack_ms < 1e-3 else np.exp(ef / attack_ms)
    cr = 0.0 if release_ms < 1e-3 else np.exp(ef / release_ms)
    y = np.empty_like(x)
    env = 0.0
    for n in range(x.shape[-1]):
        a = abs(x[n])
        c = ca if a > env else cr
        env = a + c * (env - a)
        g = 1.0 if env < thr else (env / thr) ** (1.0 / ratio - 1.0)
        y[n] = g * x[n]
    return y


COMB_TUNINGS = (1116, 1188, 1277, 1356, 1422, 1491, 1557, 1617)
ALLPASS_TUNINGS = (556, 441, 341, 225)


def reverb(x, sample_rate, room_size, damping, wet_level, dry_level, width):
    x = x.astype(np.float64)
    sr = int(sample_rate)
    combs = [np.zeros((sr * t) // 44100) for t in COMB_TUNINGS]
    aps = [np.zeros((sr * t) // 44100) for t in ALLPASS_TUNINGS]
    ci, ai = [0] * 8, [0] * 4
    last = [0.0] * 8
    damp, fb = damping * 0.4, room_size * 0.28 + 0.7
    wet = wet_level * 3.0
    wet1, dry = 0.5 * wet * (1.0 + width), dry_level * 2.0
    y = np.empty_like(x)
    for n in range(x.shape[-1]):
        inp = x[n] * 0.015
        out = 0.0
        for j in range(8):
            o = combs[j][ci[j]]
            last[j] = o * (1.0 - damp) + last[j] * damp
            combs[j][ci[j]] = inp + last[j] * fb
            ci[j] = (ci[j] + 1) % len(combs[j])
            out += o
        for j in range(4):
            b = aps[j][ai[j]]
            aps[j][ai[j]] = out + b * 0.5
            ai[j] = (ai[j] + 1) % len(aps[j])
            out = b - out
        y[n] = out * wet1 + x[n] * dry
    return y


def k_weighting_coefficients(rate):
    """The two biquads of pyloudnorm's "K-weighting" filter class, normalised by a0 (pyloudnorm/iirfilter.py)."""
    def coef(G, Q, fc, kind):
        A = 10.0 ** (G / 40.0)
        w0 = 2.0 * np.pi * (fc / rate)
        alpha = np.sin(w0) / (2.0 * Q)
        c = np.cos(w0)
        if kind == "high_shelf":
            b = [A * ((A + 1) + (A - 1) * c + 2 * np.sqrt(A) * alpha), -2 * A * ((A - 1) + (A + 1) * c),
                 A * ((A + 1) + (A - 1) * c - 2 * np.sqrt(A) * alpha)]
            a = [(A + 1) - (A - 1) * c + 2 * np.sqrt(A) * alpha, 2 * ((A - 1) - (A + 1) * c),
                 (A + 1) - (A - 1) * c - 2 * np.sqrt(A) * alpha]
        else:                                                  # high_pass
            b = [(1 + c) / 2, -(1 + c), (1 + c) / 2]
            a = [1 + alpha, -2 * c, 1 - alpha]
        return np.array(b) / a[0], np.array(a) / a[0]
    return coef(4.0, 1.0 / np.sqrt(2.0), 1500.0, "high_shelf"), coef(0.0, 0.5, 38.0, "high_pass")


def integrated_loudness(x, rate):
    """pyloudnorm.Meter(rate).integrated_loudness of a mono signal (T,)."""
    x = x.astype(np.float64)
    T_g, step = 0.4, 0.25
    if x.shape[0] < T_g * rate:
        raise ValueError("Audio must have length greater than the block size.")
    for b, a in k_weighting_coefficients(rate):
        x = scipy.signal.lfilter(b, a, x)
    T = x.shape[0] / rate
    nblk = int(np.round(((T - T_g) / (T_g * step))) + 1)
    z = np.zeros(nblk)
    for j in range(nblk):
        lo, hi = int(T_g * (j * step) * rate), int(T_g * (j * step + 1) * rate)
        z[j] = (1.0 / (T_g * rate)) * np.sum(np.square(x[lo:hi]))
    with np.errstate(divide="ignore"):
        l = -0.691 + 10.0 * np.log10(z)
        J = [j for j in range(nblk) if l[j] >= -70.0]
        zg = np.nan_to_num(np.mean(z[J])) if J else 0.0
        gamma_r = -0.691 + 10.0 * np.log10(zg) - 10.0
        J = [j for j in range(nblk) if l[j] > gamma_r and l[j] > -70.0]
        zg = np.nan_to_num(np.mean(z[J])) if J else 0.0
        return -0.691 + 10.0 * np.log10(zg)


def loudness_normalize(x, rate, target_lufs_db=-32.0):
    """effects.py:619-629: gain = 10^(clamp(target - L, -120, 40) / 20)."""
    L = integrated_loudness(x, rate)
    delta = float(np.clip(np.float32(target_lufs_db - L), -120.0, 40.0))
    return x.astype(np.float64) * 10.0 ** (delta / 20.0), L

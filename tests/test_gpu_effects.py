"""GPU: device-side effect rendering (remfx_amd.effects / csrc/fx.hip) against the numpy float64
restatement of the same published algorithms (oracle/ref_effects.py; pedalboard / pyloudnorm are absent: parity unpinned),
per effect with per-clip parameters, plus the dataset paths that use them (process_effects, DynamicEffectDataset,
EffectDataset(render_files=True))."""
import os

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.one_mode]
DEV = "cuda:0"
SR = 48000


def _clips(B, T, seed=0):
    g = torch.Generator().manual_seed(seed)
    t = torch.arange(T) / SR
    rows = []
    for b in range(B):
        env = 0.2 + 0.8 * (torch.sin(2 * torch.pi * (0.7 + 0.3 * b) * t) > 0).float()        # on / off bursts (compressor, gating)
        rows.append(env * (0.4 * torch.sin(2 * torch.pi * (180.0 + 90.0 * b) * t) + 0.05 * torch.randn(T, generator=g)))
    return torch.stack(rows)


def _rel(got, ref):
    ref = np.asarray(ref, dtype=np.float64)
    return float(np.sqrt(((got.double().cpu().numpy() - ref) ** 2).mean()) / max(np.sqrt((ref ** 2).mean()), 1e-30))


def test_distortion_delay_vs_oracle():
    from oracle import ref_effects as R
    from remfx_amd import effects as E
    x = _clips(3, 30011)
    fx = E.RandomPedalboardDistortion(SR)
    params = [dict(drive_db=v) for v in (-20.0, 3.3, 12.0)]
    y = fx.render(x.to(DEV), params)
    for b, p in enumerate(params):
        assert _rel(y[b], R.distortion(x[b].numpy(), **p)) < 2e-6
    fx = E.RandomPedalboardDelay(SR)
    params = [dict(delay_seconds=0.1, feedback=0.6, mix=0.7), dict(delay_seconds=0.2537, feedback=0.05, mix=0.3),
              dict(delay_seconds=0.9, feedback=0.4, mix=0.0)]
    y = fx.render(x.to(DEV), params)
    for b, p in enumerate(params):
        assert _rel(y[b], R.delay(x[b].numpy(), SR, **p)) < 2e-6, b


def test_chorus_compressor_reverb_vs_oracle():
    from oracle import ref_effects as R
    from remfx_amd import effects as E
    x = _clips(3, 24000, seed=1)
    fx = E.RandomPedalboardChorus(SR)
    params = [dict(rate_hz=0.25, depth=0.6, centre_delay_ms=5.0, feedback=0.6, mix=0.7),
              dict(rate_hz=4.0, depth=0.0, centre_delay_ms=10.0, feedback=0.1, mix=0.1),
              dict(rate_hz=1.7, depth=0.6, centre_delay_ms=6.0, feedback=0.35, mix=0.5)]     # depth 0.6 at centre 5-6 ms reaches the 1 ms floor
    y = fx.render(x.to(DEV), params)
    for b, p in enumerate(params):
        # the delay is an fp32 quantity on the device: a sample whose delay sits within rounding of an integer interpolates from
        # the neighbouring pair -- a continuous function, so the error stays at fp32 level
        assert _rel(y[b], R.chorus(x[b].numpy(), SR, **p)) < 2e-5, b
    fx = E.RandomPedalboardCompressor(SR)
    params = [dict(threshold_db=-42.0, ratio=4.0, attack_ms=1.0, release_ms=10.0),
              dict(threshold_db=-6.0, ratio=1.5, attack_ms=50.0, release_ms=250.0),
              dict(threshold_db=-20.0, ratio=2.5, attack_ms=7.0, release_ms=80.0)]
    y = fx.render(x.to(DEV), params)
    for b, p in enumerate(params):
        assert _rel(y[b], R.compressor(x[b].numpy(), SR, **p)) < 2e-5, b
    fx = E.RandomPedalboardReverb(SR)
    params = [dict(room_size=1.0, damping=0.0, wet_dry=0.7, width=1.0), dict(room_size=0.0, damping=1.0, wet_dry=0.2, width=0.0),
              dict(room_size=0.6, damping=0.45, wet_dry=0.5, width=0.3)]
    y = fx.render(x.to(DEV), params)
    for b, p in enumerate(params):
        ref = R.reverb(x[b].numpy(), SR, p["room_size"], p["damping"], p["wet_dry"], 1.0 - p["wet_dry"], p["width"])
        # room_size 1 = comb feedback 0.98: fp32 rounding recirculates; measured ~1e-5
        assert _rel(y[b], ref) < 1e-4, b


def test_loudness_normalize_vs_oracle():
    """BS.1770 integrated loudness + gain (effects.py:619-629), incl. a clip whose quiet half is gated out, at a length that is
    not a multiple of anything (tail chunk of the blocked IIR, partial last hop) and at the full clip length."""
    from oracle import ref_effects as R
    from remfx_amd import effects as E
    for T in (48000 + 1234, 262144):
        x = _clips(4, T, seed=2)
        x[1, : T // 2] *= 1e-4                              # half the clip below the absolute / relative gates
        x[2] *= 3.0
        x[3] = 0.0                                          # silence: L = -inf, gain clamps at +40 dB
        norm = E.LoudnessNormalize(SR, target_lufs_db=-20.0)
        lufs, gain = norm.measure(x.to(DEV))
        y = norm(x.to(DEV).unsqueeze(1)).squeeze(1)
        for b in range(3):
            yr, L = R.loudness_normalize(x[b].numpy(), SR, -20.0)
            assert abs(float(lufs[b]) - L) < 2e-3, (T, b, float(lufs[b]), L)
            assert _rel(y[b], yr) < 5e-4, (T, b)
        assert float(lufs[3]) == float("-inf") and abs(float(gain[3]) - 100.0) < 1e-3
    with pytest.raises(ValueError, match="block size"):
        E.LoudnessNormalize(SR)(torch.zeros(1, 1000, device=DEV))


def test_forward_draws_like_the_reference_and_batches():
    """forward(): (channels, samples) like the reference, one parameter set per call drawn with the reference's calls in its
    order; (B, 1, samples): one set per clip; identical to rendering each clip alone with the same draws."""
    from remfx_amd import effects as E
    x = _clips(3, 20000, seed=3).to(DEV)
    fx = E.RandomPedalboardChorus(SR)
    torch.manual_seed(11)
    yb = fx(x.unsqueeze(1))
    sets = fx.last_params
    assert len(sets) == 3 and list(sets[0]) == ["rate_hz", "depth", "centre_delay_ms", "feedback", "mix"]
    torch.manual_seed(11)
    for b in range(3):
        y1 = fx(x[b:b + 1])                                   # (1, T): the reference's call
        assert fx.last_params[0] == sets[b]
        assert torch.equal(y1, yb[b])
    with pytest.raises(ValueError, match="no CPU path"):
        fx(torch.zeros(1, 100))


def test_process_effects_and_dynamic_dataset(tmp_path):
    """datasets.process_effects / DynamicEffectDataset (datasets.py:205-330) and EffectDataset(render_files=True)
    (datasets.py:399-452) on the device: labels match the applied effects, outputs sit at the -20 LUFS the in-between
    normalisation targets, the rendered layout reads back."""
    from oracle import ref_effects as R
    from remfx_amd import datasets as D, effects as E
    fx = {"reverb": E.RandomPedalboardReverb(SR), "chorus": E.RandomPedalboardChorus(SR), "delay": E.RandomPedalboardDelay(SR),
          "distortion": E.RandomPedalboardDistortion(SR), "compressor": E.RandomPedalboardCompressor(SR)}
    torch.manual_seed(5)
    np.random.seed(5)
    with pytest.warns(UserWarning, match="white-noise"):
        ds = D.DynamicEffectDataset(root=None, sample_rate=SR, chunk_size=65536, total_chunks=4, effect_modules=fx,
                                    effects_to_keep=["compressor"], effects_to_remove=["distortion", "reverb", "chorus", "delay"],
                                    num_kept_effects=[0, 1], num_removed_effects=[1, 4], mode="train")
    seen = 0
    for i in range(4):
        wet, dry, dl, wl = ds[i]
        assert wet.is_cuda and wet.shape == dry.shape == (1, 65536) and dl.shape == wl.shape == (5,)
        assert 1 <= int(wl.sum()) <= 4 and wl[4] == 0 and int(dl.sum()) <= 1 and dl[:4].sum() == 0
        for t in (wet, dry):
            assert abs(R.integrated_loudness(t[0].cpu().numpy(), SR) + 20.0) < 0.05
        assert float((wet - dry).abs().max()) > 1e-3
        seen += int(wl.sum())
    assert seen >= 4
    dl = D.EffectDatamodule(ds, ds, ds, train_batch_size=2, test_batch_size=2, num_workers=4).train_dataloader()
    xb, yb, dlb, wlb = next(iter(dl))
    assert xb.shape == (2, 1, 65536) and xb.is_cuda and wlb.shape == (2, 5)
    # rendering a corpus to disk in the reference's layout
    corpus = tmp_path / "corpus" / "audio_mono-mic"
    corpus.mkdir(parents=True)
    g = torch.Generator().manual_seed(9)
    for k in range(3):
        D.save_wav(corpus / f"0{k}_clip.wav", torch.randn(1, 44100 * 3, generator=g) * 0.1, 44100)       # resampled on the device
    kw = dict(root=str(tmp_path / "corpus"), sample_rate=SR, chunk_size=32768, total_chunks=3, effect_modules=fx,
              effects_to_keep=[], effects_to_remove=["distortion", "delay"], num_kept_effects=[0, 0], num_removed_effects=[2, 2],
              render_root=str(tmp_path / "render"), mode="train")
    ds2 = D.EffectDataset(render_files=True, **kw)
    assert len(ds2) == 3
    x0, y0, d0, w0 = ds2[0]
    assert x0.shape == y0.shape == (1, 32768) and w0.tolist() == [0, 0, 1, 1, 0] and d0.sum() == 0
    ds3 = D.EffectDataset(render_files=False, **kw)           # the consumer path of the reference reads it back
    assert len(ds3) == 3 and torch.equal(ds3[0][0], x0)

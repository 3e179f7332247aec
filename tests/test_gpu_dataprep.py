"""GPU: the data path on the device: polyphase resampling as one strided gather-GEMM, the
InferenceDataset resample / mono / pad chain, Cnn14's resampling front end and its training-time spectrogram masks."""
import os

import pytest
import torch

from tests.conftest import check

pytestmark = [pytest.mark.gpu, pytest.mark.one_mode]
DEV = "cuda:0"


@pytest.mark.parametrize("rates", [(44100, 48000), (48000, 16000), (22050, 48000), (48000, 44100)])
def test_resample_vs_oracle(rates):
    from oracle import ref_resample
    from remfx_amd.resample import resample
    orig, new = rates
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 1, 3001, generator=g)
    y = ref_resample.resample(x, orig, new)
    yd = resample(x.to(DEV), orig, new).cpu()
    assert yd.shape == y.shape
    check(float(((yd - y) ** 2).mean().sqrt()), 1e-6, max(1.0, float(y.abs().max())))       # exact-fp32 kernel (one input channel)


def test_inference_dataset_resamples_on_device(tmp_path):
    """clean/ and effected/ WAV folders at different rates and channel counts -> (effected, clean, 0s, 1s) at the target
    rate, mono, effected padded / trimmed to clean (reference datasets.py:598-620), against the oracle chain."""
    from oracle import ref_resample
    from remfx_amd import datasets
    g = torch.Generator().manual_seed(1)
    (tmp_path / "clean").mkdir(); (tmp_path / "effected").mkdir()
    clean = torch.randn(2, 4410, generator=g) * 0.1             # stereo, 44.1 kHz
    eff = torch.randn(1, 4000, generator=g) * 0.1               # mono, 44.1 kHz, shorter
    datasets.save_wav(tmp_path / "clean" / "a.wav", clean, 44100)
    datasets.save_wav(tmp_path / "effected" / "a.wav", eff, 44100)
    ds = datasets.InferenceDataset(str(tmp_path), 48000, device=DEV)
    assert len(ds) == 1
    e, c, dry, wet = ds[0]
    cr = ref_resample.resample(clean, 44100, 48000, table_dtype=torch.float32).sum(0, keepdim=True)
    er = ref_resample.resample(eff, 44100, 48000, table_dtype=torch.float32).sum(0, keepdim=True)
    er = torch.nn.functional.pad(er, (0, cr.shape[1] - er.shape[1]))
    assert c.shape == cr.shape and e.shape == cr.shape and e.is_cuda
    check(float((c.cpu() - cr).abs().max()), 1e-5)
    check(float((e.cpu() - er).abs().max()), 1e-5)
    assert float(dry.sum()) == 0.0 and float(wet.sum()) == 5.0


def test_cnn14_resampling_front_end_and_specaugment():
    """sample_rate != model_sample_rate (cfg/model/cls_panns_16k.yaml) and specaugment=True (cfg/exp/remfx_detect.yaml:60):
    eval output = the 16 kHz network applied to the oracle-resampled clip; train=True masks one frequency and one time span
    per clip and leaves eval untouched (reference classifier.py:180-187, 198-204)."""
    from oracle import ref_resample
    from remfx_amd.classifier import Cnn14, spec_augment
    torch.manual_seed(2)
    net = Cnn14(num_classes=5, sample_rate=48000, model_sample_rate=16000, n_fft=1024, hop_length=256, n_mels=64,
                specaugment=True).to(DEV).eval()
    same = Cnn14(num_classes=5, sample_rate=16000, model_sample_rate=16000, n_fft=1024, hop_length=256, n_mels=64).to(DEV).eval()
    sd = net.state_dict()
    assert "resample.kernel" in sd             # torchaudio.transforms.Resample keeps its filter bank as a buffer (strict ckpt loads)
    same.load_state_dict({k: v for k, v in sd.items() if not k.startswith("resample.")})
    x = torch.randn(2, 1, 48000, generator=torch.Generator().manual_seed(3)) * 0.1
    with torch.no_grad():
        out = torch.hstack(net(x.to(DEV)))
        ref = torch.hstack(same(ref_resample.resample(x, 48000, 16000).to(DEV)))
        again = torch.hstack(net(x.to(DEV), train=False))
    check(float((out - ref).abs().max()), 2e-4)
    assert torch.equal(out, again)                                   # specaugment is inert outside training
    mel = torch.rand(3, 1, 64, 188, device=DEV) + 0.5
    torch.manual_seed(4)
    masked = spec_augment(mel, 64, 128)
    zero = masked == 0
    assert zero.any() and torch.equal(masked[~zero], mel[~zero])
    for b in range(3):
        z = zero[b, 0]
        rows, cols = z.all(1), z.all(0)                              # fully masked mel bins / frames
        assert torch.equal(z, rows[:, None] | cols[None, :])         # the zeros are exactly one row band + one column band
        for band in (rows, cols):
            idx = band.nonzero().flatten()
            assert idx.numel() == 0 or int(idx[-1] - idx[0]) + 1 == idx.numel()      # contiguous span

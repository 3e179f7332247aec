"""GPU parity: Cnn14 vs outputs of the imported reference module (tests/golden/cnn14_full.npz),
and the RemFX / RemFXChainInference control flow vs a trace of the reference's own
remfx/models.py (tests/golden/flow.npz, oracle/gen_golden.py:gen_flow)."""
import os

import numpy as np
import pytest
import torch

from tests.conftest import check, mode, tol
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _cnn14_sd():
    from oracle import ref_cnn14
    sd = ref_cnn14.cnn14_init_state_dict(seed=7)
    gen = torch.Generator().manual_seed(8)
    for k in list(sd):
        if k.endswith("running_mean"):
            sd[k] = torch.randn(sd[k].shape, generator=gen) * 0.05
        elif k.endswith("running_var"):
            sd[k] = torch.rand(sd[k].shape, generator=gen) * 0.5 + 0.75
        elif ".bn" in k and k.endswith("weight"):
            sd[k] = torch.rand(sd[k].shape, generator=gen) * 0.4 + 0.8
        elif ".bn" in k and k.endswith("bias"):
            sd[k] = torch.randn(sd[k].shape, generator=gen) * 0.05
    for k in list(sd):
        if k.startswith("heads.") and k.endswith("weight"):
            sd[k] = sd[k] * 40.0
    return sd


def test_cnn14_golden(golden_dir):
    from remfx_amd.classifier import Cnn14
    g = np.load(os.path.join(golden_dir, "cnn14_full.npz"))
    net = Cnn14(num_classes=5, sample_rate=48000, model_sample_rate=48000, n_fft=2048, hop_length=512, n_mels=128)
    missing = net.load_state_dict(_cnn14_sd(), strict=False)
    assert not missing.unexpected_keys
    net = net.to(DEV).eval()
    x = torch.from_numpy(g["x"]).to(DEV)
    with torch.no_grad():
        mel = net.melspec(x).cpu().numpy()
        out = torch.hstack(net(x)).cpu().numpy()
        net.train()                                   # BatchNorm batch statistics
        outb = torch.hstack(net(x, train=False)).cpu().numpy()
    rel = np.sqrt(((mel - g["mel"]) ** 2).mean()) / np.abs(g["mel"]).max()
    check(rel, 1e-6, what=rel)
    # the detector runs at fp32 parity in EVERY session mode (Cnn14.forward, train=False -> ops.at_least_fp32_parity): the
    # bf16 session has the bf16x3 bounds
    np.testing.assert_allclose(out, g["out_eval"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(outb, g["out_bnbatch"], rtol=5e-3, atol=5e-4)
    # bit-exact detected-effect labels (models.py:61-64), all modes, no mask around the threshold
    assert np.array_equal(out > 0.5, g["out_eval"] > 0.5)
    assert np.array_equal(outb > 0.5, g["out_bnbatch"] > 0.5)


_FULL = {}


def _full_length_case(centre="gap"):
    """16 full-length (262144-sample) clips of different character + the CPU oracle's probabilities (computed once per session).
    centre = "gap": each head's threshold sits in the widest gap among the central clips (the most comfortable place);
    "median": at the median logit (midway between the 8th and 9th clip, whatever the gap there is)."""
    if centre not in _FULL:
        from oracle import ref_cnn14
        g = torch.Generator().manual_seed(41)
        t = torch.arange(262144) / 48000.0
        clips = []
        for i in range(16):
            kind = i % 4
            if kind == 0:
                c = torch.randn(262144, generator=g) * (0.02 + 0.05 * i)
            elif kind == 1:
                c = 0.3 * torch.sin(2 * torch.pi * (100.0 + 300.0 * i + 500.0 * i * t) * t)
            elif kind == 2:
                c = torch.randn(262144, generator=g).cumsum(0)
                c = 0.2 * c / c.abs().max()
            else:
                env = torch.exp(-((t * (3 + i)) % 1.0) * 8.0)
                c = env * torch.sin(2 * torch.pi * 220.0 * (1 + i / 4) * t) * 0.4 + 0.01 * torch.randn(262144, generator=g)
            clips.append(c)
        x = torch.stack(clips)[:, None, :]
        sd = _cnn14_sd()
        torch.set_num_threads(min(32, os.cpu_count() or 8))
        with torch.no_grad():
            p0 = torch.hstack(ref_cnn14.cnn14_forward(x, sd, bn_train=False)).double()
            # a randomly initialised Cnn14 in eval mode barely depends on its input (logit spread ~1e-2 across clips):
            # re-centre and stretch each head (an affine map of its weight / bias) so that the 16 clips fall on BOTH sides
            # of the threshold with logits spread over about +-2 -- the fp32 rounding noise is stretched with them, which
            # makes this a far harsher label test than the raw heads would be
            z = torch.log(p0 / (1 - p0))
            for k in range(5):
                a = float(2.0 / z[:, k].std())
                zs = z[:, k].sort().values
                if centre == "gap":                            # threshold in the widest gap among the central clips
                    j = int((zs[5:12] - zs[4:11]).argmax()) + 4
                else:                                          # at the median logit
                    j = 7
                c = float((zs[j] + zs[j + 1]) / 2)
                sd[f"heads.{k}.bias"] = (sd[f"heads.{k}.bias"].double() - c) * a
                sd[f"heads.{k}.weight"] = sd[f"heads.{k}.weight"].double() * a
                sd[f"heads.{k}.bias"], sd[f"heads.{k}.weight"] = sd[f"heads.{k}.bias"].float(), sd[f"heads.{k}.weight"].float()
            ref = torch.hstack(ref_cnn14.cnn14_forward(x, sd, bn_train=False)).numpy()
        _FULL[centre] = (x, ref, sd)
    return _FULL[centre]


def test_detector_labels_full_length_bit_exact():
    """BASELINE config 5's detector on 16 FULL-LENGTH clips (262144 samples, 48 kHz, n_fft 2048 / hop 512 / 128 mels =
    cfg/model/cls_panns_48k*.yaml): thresholded labels (reference models.py:60-64) equal the CPU oracle's bit for bit in
    every session mode, through FXClassifier.forward as RemFXChainInference calls it."""
    from remfx_amd.classifier import Cnn14
    from remfx_amd.models import FXClassifier
    x, ref, sd = _full_length_case()
    net = Cnn14(num_classes=5, sample_rate=48000, model_sample_rate=48000, n_fft=2048, hop_length=512, n_mels=128)
    net.load_state_dict(sd, strict=False)
    cls = FXClassifier(3e-4, 1e-3, 48000, net).to(DEV).eval()
    with torch.no_grad():
        out = torch.hstack(cls(x.to(DEV))).cpu().numpy()
    margin = np.abs(ref - 0.5).min()
    err = np.abs(out - ref).max()
    print(f"detector labels, threshold in the widest central gap: margin {margin:.3e} / max|out - ref| {err:.3e} = {margin / err:.1f}")
    assert (ref > 0.5).any() and (ref <= 0.5).any(), "degenerate case: all labels equal"
    # probabilities: the stretched heads amplify the fp32 feature noise by ~2 / (logit spread) ~ 100-400x
    # (measured: 6e-3 with the 2^-17 products of bf16x3, which is what a bf16 session's detector runs in)
    assert err < tol(5e-3, bf16x3=3e-2, bf16=3e-2), err
    assert np.array_equal(out > 0.5, ref > 0.5), (margin, err)


def test_detector_labels_threshold_at_median_logit():
    """Same 16 clips with every head's threshold at the MEDIAN logit (not where the gap is widest): the margin is whatever the
    clips give.  Labels must agree wherever the oracle's probability is further from 0.5 than the measured error; the ratio
    margin / max|out - ref| is printed (pytest -s) and, when it exceeds 1, the labels must be equal bit for bit."""
    from remfx_amd.classifier import Cnn14
    from remfx_amd.models import FXClassifier
    x, ref, sd = _full_length_case("median")
    net = Cnn14(num_classes=5, sample_rate=48000, model_sample_rate=48000, n_fft=2048, hop_length=512, n_mels=128)
    net.load_state_dict(sd, strict=False)
    cls = FXClassifier(3e-4, 1e-3, 48000, net).to(DEV).eval()
    with torch.no_grad():
        out = torch.hstack(cls(x.to(DEV))).cpu().numpy()
    margin = np.abs(ref - 0.5).min()
    err = np.abs(out - ref).max()
    print(f"detector labels, threshold at the median logit: margin {margin:.3e} / max|out - ref| {err:.3e} = {margin / err:.2f}")
    assert (ref > 0.5).sum(0).min() >= 7 and (ref <= 0.5).sum(0).min() >= 7          # 8 / 8 per head, up to ties
    assert err < tol(5e-3, bf16x3=3e-2, bf16=3e-2), err
    decided = np.abs(ref - 0.5) > err
    assert np.array_equal((out > 0.5)[decided], (ref > 0.5)[decided])
    if margin > err:
        assert np.array_equal(out > 0.5, ref > 0.5), (margin, err)

def test_cnn14_train_step_vs_oracle():
    """FXClassifier loss + a few gradients (train-mode BN) vs autograd over the CPU oracle."""
    from oracle import ref_cnn14
    from remfx_amd.classifier import Cnn14
    from remfx_amd.models import FXClassifier
    sd = _cnn14_sd()
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 1, 24000, generator=g) * 0.1
    lab = (torch.rand(3, 5, generator=g) > 0.5).float()
    sdr = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone())
           for k, v in sd.items()}
    outs = ref_cnn14.cnn14_forward(x, sdr, bn_train=True)
    lref = sum(torch.nn.functional.binary_cross_entropy(o.squeeze(-1), lab[:, k]) for k, o in enumerate(outs))
    lref.backward()
    net = Cnn14(5, 48000, 48000, 2048, 512, 128)
    net.load_state_dict(sd, strict=False)
    model = FXClassifier(3e-4, 1e-3, 48000, net).to(DEV)
    model.train()
    # dropout off (p applies only with train=True): compare through the valid path on a train-mode net
    loss = model.common_step((x.to(DEV), None, None, lab.to(DEV)), 0, mode="valid")
    loss.backward()
    check(abs(float(loss) - float(lref)), 2e-4, max(1.0, abs(float(lref))))
    for k in ("conv_block1.conv1.weight", "conv_block3.bn1.weight", "conv_block6.conv2.weight", "fc1.weight",
              "heads.2.weight"):
        got, ref = dict(net.named_parameters())[k].grad.cpu(), sdr[k].grad
        scale = max(1e-6, float(ref.abs().max()))
        # train-mode BatchNorm gradients are ill-conditioned: the exact-fp32 mode itself sits 3e-3 from the CPU oracle
        # (summation order); the 2^-17 product rounding of bf16x3 measures 8.5e-3 -> bound 3 x the fp32 one
        check(float(((got - ref) ** 2).mean().sqrt()), 5e-3, scale, bf16x3=1.5e-2, what=k)


def test_remfx_step_and_chain_flow(golden_dir):
    from oracle import ref_tcn
    from remfx_amd import models
    g = np.load(os.path.join(golden_dir, "flow.npz"))
    net = models.TCNModel(sample_rate=48000, num_bins=1025, ninputs=1, noutputs=1, nblocks=3, channel_width=8,
                          kernel_size=7, stack_size=10, dilation_growth=2, causal=False)
    net.model.load_state_dict(ref_tcn.tcn_init_state_dict(1, 1, 3, 8, 7, seed=21))
    model = models.RemFX(1e-4, 0.95, 0.999, 1e-6, 1e-3, 48000, net).to(DEV)
    x, y = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["y"]).to(DEV)
    loss = model.training_step((x, y, None, None), 0)
    check(abs(float(loss) - float(g["loss"])), 1e-4, abs(float(g["loss"])))
    assert sorted(model.logged) == g["log_names"].tolist()
    for k, v in zip(g["log_names"].tolist(), g["log_vals"]):
        check(abs(float(model.logged[k]) - float(v)), 2e-3, max(1.0, abs(float(v))), what=k)

    class Tag(nn.Module):
        def __init__(self, mul, add):
            super().__init__(); self.mul, self.add = mul, add
        def sample(self, z):
            return z * self.mul + self.add
    class Holder(nn.Module):
        def __init__(self, m):
            super().__init__(); self.model = m
    names = models.ALL_EFFECT_NAMES
    mods = {n: Holder(Tag(1.0 + 0.1 * (i + 1), 0.01 * (i + 1))) for i, n in enumerate(names)}
    probs = torch.from_numpy(g["probs"]).to(DEV)
    class FakeCls(nn.Module):
        def forward(self, z):
            return [probs[:, k:k + 1] for k in range(5)]
    order = ["RandomPedalboardDistortion", "RandomPedalboardCompressor", "RandomPedalboardReverb",
             "RandomPedalboardChorus", "RandomPedalboardDelay"]
    chain = models.RemFXChainInference(mods, 48000, 1025, order, classifier=FakeCls())
    xc, yc = torch.from_numpy(g["xc"]).to(DEV), torch.from_numpy(g["yc"]).to(DEV)
    closs, cout = chain.forward((xc, yc, None, None), 0)
    # strict > 0.5 threshold: 0.5 is NOT detected, 0.51 is
    assert chain.last_labels.cpu().tolist() == [[1, 0, 0, 1, 0], [0, 1, 0, 0, 1], [0, 0, 0, 0, 0]]
    np.testing.assert_allclose(cout.cpu().numpy(), g["chain_out"], rtol=1e-6, atol=1e-6)
    check(abs(float(closs) - float(g["chain_loss"])), 1e-4, abs(float(g["chain_loss"])))
    chain.test_step((xc, yc, None, None), 0)
    assert sorted(chain.logged) == g["chain_log_names"].tolist()
    for k, v in zip(g["chain_log_names"].tolist(), g["chain_log_vals"]):
        check(abs(float(chain.logged[k]) - float(v)), 2e-3, max(1.0, abs(float(v))), what=k)


class _ToyEmbedder(nn.Module):
    """Stand-in for a HEAR scene-embedding model (hearbaseline / wav2clip_hear / panns_hear are not in the image): frame
    energies through a fixed random projection -> (B, dim)."""

    def __init__(self, dim, frame=400):
        super().__init__()
        g = torch.Generator().manual_seed(dim)
        self.frame = frame
        self.w = nn.Parameter(torch.randn(64, dim, generator=g) / 8.0)

    def get_scene_embeddings(self, audio):
        B, T = audio.shape
        n = T // self.frame
        e = audio[:, :n * self.frame].reshape(B, n, self.frame).pow(2).mean(-1).add(1e-8).log()          # (B, n)
        feats = torch.nn.functional.adaptive_avg_pool1d(e.unsqueeze(1), 64).squeeze(1)
        return feats @ self.w


@pytest.mark.parametrize("cls_name,dim,rate", [("PANNs", 2048, 32000), ("VGGish", 128, 16000), ("Wav2CLIP", 512, 16000),
                                               ("wav2vec2", 1024, 16000)])
def test_hear_embedding_classifiers(cls_name, dim, rate):
    """classifier.py:16-128 + the non-Cnn14 branch of FXClassifier (models.py:457-476, 503-506, 540-570): device-side
    resampling to the embedding model's rate, frozen embedder, MLP head on the HIP GEMMs, cross-entropy against the wet-label
    vector, multilabel F1 -- against the same computation in plain torch; one optimiser step moves the head only."""
    from remfx_amd import classifier, models
    from remfx_amd.resample import resample
    torch.manual_seed(3)
    emb = _ToyEmbedder(dim)
    net = getattr(classifier, cls_name)(num_classes=5, sample_rate=48000, embedder=emb)
    assert sorted(k for k in net.state_dict() if k.startswith("proj")) == sorted(
        f"proj.{i}.{n}" for i in (0, 2, 4) for n in ("weight", "bias"))
    assert "resample.kernel" in net.state_dict()                 # torchaudio.transforms.Resample keeps its filter bank as a buffer
    model = models.FXClassifier(3e-4, 1e-3, 48000, net, label_smoothing=0.1).to(DEV)
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(6, 1, 24000, generator=g) * torch.logspace(-2, 0, 6).view(6, 1, 1)).to(DEV)
    lab = (torch.rand(6, 5, generator=g) > 0.5).float().to(DEV)
    logits = model(x)
    assert logits.shape == (6, 5)
    with torch.no_grad():
        e = emb.get_scene_embeddings(resample(x, 48000, rate).reshape(6, -1))
        ref = net.proj(e)
    check(float((logits - ref).abs().max()), 2e-4, max(1.0, float(ref.abs().max())), what=cls_name)
    opt = model.configure_optimizers()
    assert len(opt.flat.params) == 6                         # the head's three Linear layers; the embedder is frozen
    opt.zero_grad()
    loss = model.training_step((x, None, None, lab), 0)
    lref = torch.nn.functional.cross_entropy(ref, lab, label_smoothing=0.1)
    check(abs(float(loss) - float(lref)), 2e-4, max(1.0, float(lref)))
    probs = torch.sigmoid(logits.detach())               # the metric is defined on the model's own outputs (a bf16 session may flip a borderline one)
    for k, name in enumerate(model.effects):
        pred, t = (probs[:, k] > 0.5), lab[:, k] > 0.5
        tp, fp, fn = float((pred & t).sum()), float((pred & ~t).sum()), float((~pred & t).sum())
        f1 = 2 * tp / (2 * tp + fp + fn) if (2 * tp + fp + fn) else 0.0
        assert abs(model.logged[f"train_f1_{name}"] - f1) < 1e-6
    assert "train_avg_acc" in model.logged and "train_loss" in model.logged
    w_emb, w_head = emb.w.detach().clone(), net.proj[4].weight.detach().clone()
    loss.backward()
    opt.step()
    assert torch.equal(emb.w, w_emb) and not torch.equal(net.proj[4].weight, w_head)


def test_hear_classifier_without_package_raises():
    from remfx_amd import classifier
    with pytest.raises(ImportError, match="panns_hear"):
        classifier.PANNs(num_classes=5, sample_rate=48000)


@pytest.mark.one_mode
def test_mixup_branch_on_device(golden_dir):
    """The mixup training branch with the batch resident on the GPU (how `FXClassifier.training_step` sees it): same draws, partners,
    OR-ed labels, loss and logged scalars as the fixture recorded from the reference's own `mixup` / `FXClassifier`
    (remfx/models.py:393-420, 491-500; tests/golden/mixup.npz)."""
    from oracle.gen_golden import tiny_heads_forward, tiny_heads_state
    from remfx_amd.classifier import Cnn14
    from remfx_amd.models import FXClassifier, mixup
    g = np.load(os.path.join(golden_dir, "mixup.npz"))
    x, y = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["y"]).to(DEV)
    for s in g["seeds"]:
        np.random.seed(int(s))
        torch.manual_seed(int(s))
        mx, my, lam = mixup(x, y)
        assert mx.device.type == "cuda" and torch.equal(my.cpu(), torch.from_numpy(g[f"my{s}"]))
        assert torch.allclose(mx.cpu(), torch.from_numpy(g[f"mx{s}"]), rtol=0, atol=1e-7)

    class Tiny(Cnn14):
        def __init__(self, st):
            nn.Module.__init__(self)
            self.w, self.b = nn.Parameter(st["w"].clone()), nn.Parameter(st["b"].clone())

        def forward(self, z, train=False):
            return tiny_heads_forward(z, self.w, self.b)
    for s in (1, 4):
        cls = FXClassifier(3e-4, 1e-3, 48000, Tiny(tiny_heads_state()), mixup=True).to(DEV)
        np.random.seed(s)
        torch.manual_seed(s)
        loss = cls.training_step((x, None, None, y), 0)
        assert abs(float(loss) - float(g[f"cls_loss{s}"])) < 1e-5 * max(1.0, abs(float(loss)))
        names = sorted(cls.logged)
        assert names == list(g[f"cls_log_names{s}"])
        assert np.allclose([float(cls.logged[k]) for k in names], g[f"cls_log_vals{s}"], rtol=1e-4, atol=1e-5)

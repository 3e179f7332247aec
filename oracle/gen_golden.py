"""Golden-vector generator (test infrastructure; runs ONLY in the build container).

Imports the in-tree reference modules from /root/reference through import shims
(the reference's own third-party imports -- pytorch_lightning, omegaconf,
torchaudio, hearbaseline, ... -- are absent from this image) and
records small input/output fixtures under tests/golden/.  Weights are produced
by the oracle's own seeded generators and loaded into the reference modules with
load_state_dict, so the fixtures hold inputs + expected outputs only.

Nothing here travels to the GPU box except the .npz files it writes.
Usage:  PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py
"""
import os
import sys
import types

sys.dont_write_bytecode = True          # never drop __pycache__ into /root/reference
import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import ref_cnn14, ref_hdemucs, ref_losses, ref_tcn  # noqa: E402


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_shims():
    """Stub the third-party packages the reference imports at module import time."""
    ident = lambda f=None, *a, **k: f
    pl = _mod("pytorch_lightning", LightningModule=_Lightning, LightningDataModule=object,
              Callback=object, Trainer=object, seed_everything=lambda s: torch.manual_seed(s))
    pl.utilities = _mod("pytorch_lightning.utilities", rank_zero_only=ident)
    pl.loggers = _mod("pytorch_lightning.loggers", WandbLogger=object, CSVLogger=object)
    pl.loggers.logger = _mod("pytorch_lightning.loggers.logger", Logger=object)
    pl.callbacks = _mod("pytorch_lightning.callbacks", ModelCheckpoint=object)
    _mod("omegaconf", DictConfig=dict)
    ta = _mod("torchaudio")
    ta.functional = _mod("torchaudio.functional", resample=None)
    ta.transforms = _mod("torchaudio.transforms", MelSpectrogram=_MelStandIn, Resample=nn.Identity,
                         FrequencyMasking=lambda *a, **k: nn.Identity(),
                         TimeMasking=lambda *a, **k: nn.Identity())
    ta.models = _mod("torchaudio.models", HDemucs=ref_hdemucs.HDemucs)
    for n in ("hearbaseline", "hearbaseline.vggish", "hearbaseline.wav2vec2", "wav2clip_hear",
              "panns_hear", "pedalboard", "pyloudnorm", "torchvision", "torchvision.transforms",
              "asteroid", "asteroid.models", "umx", "umx.openunmix", "wandb"):
        _mod(n)
    sys.modules["hearbaseline"].vggish = sys.modules["hearbaseline.vggish"]
    sys.modules["hearbaseline"].wav2vec2 = sys.modules["hearbaseline.wav2vec2"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["asteroid"].models = sys.modules["asteroid.models"]
    _mod("umx.openunmix.model", OpenUnmix=object, Separator=object)
    tm = _mod("torchmetrics")
    tm.classification = _mod("torchmetrics.classification", Accuracy=lambda **k: _BinaryAccuracy(),
                             MultilabelF1Score=lambda *a, **k: nn.Identity())
    al = _mod("auraloss")
    al.time = _mod("auraloss.time", SISDRLoss=_SISDR)
    al.freq = _mod("auraloss.freq", MultiResolutionSTFTLoss=_MRSTFT)
    if REF not in sys.path:
        sys.path.insert(0, REF)


class _Lightning(nn.Module):
    """pl.LightningModule stand-in: an nn.Module that records self.log calls."""
    def __init__(self):
        super().__init__()
        self.logged = {}

    def log(self, name, value, **kw):
        self.logged[name] = float(value)


class _MelStandIn(nn.Module):
    """torchaudio.transforms.MelSpectrogram stand-in backed by the oracle mel."""
    def __init__(self, sample_rate, n_fft, hop_length=None, n_mels=128):
        super().__init__()
        self.a = (sample_rate, n_fft, hop_length, n_mels)

    def forward(self, x):
        return ref_cnn14.mel_spectrogram(x, *self.a)


class _BinaryAccuracy(nn.Module):
    """torchmetrics.classification.Accuracy(task="binary") on one batch of probabilities: mean((p > 0.5) == target)."""
    def forward(self, preds, target):
        return ((preds > 0.5).float() == target.float()).float().mean()


class _MRSTFT(nn.Module):
    def __init__(self, **kw):
        super().__init__()

    def forward(self, x, y):
        return ref_losses.mrstft_loss(x, y)


class _SISDR(nn.Module):
    def forward(self, x, y):
        return ref_losses.sisdr_loss(x, y)


def gen_utils():
    import remfx.utils as ru
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 1, 8192, generator=g)
    S = ru.spectrogram(x, torch.hann_window(512), 512, 128, 0.3)
    a = torch.arange(40.0).reshape(2, 20)
    np.savez_compressed(os.path.join(OUT, "utils_small.npz"), x=x.numpy(), spec=S.numpy(),
                        crop_in=a.numpy(), center7=ru.center_crop(a, 7).numpy(),
                        causal7=ru.causal_crop(a, 7).numpy())
    print("utils_small", tuple(S.shape))


def gen_tcn():
    from remfx.tcn import TCN
    cases = {
        "tcn_small": dict(cfg=dict(ninputs=1, noutputs=1, nblocks=4, channel_width=8, kernel_size=7,
                                   stack_size=10, dilation_growth=2, causal=False), B=2, T=2048, seed=3),
        "tcn_mid": dict(cfg=dict(ninputs=1, noutputs=1, nblocks=12, channel_width=32, kernel_size=7,
                                 stack_size=10, dilation_growth=2, causal=False), B=1, T=8192, seed=4),
        "tcn_causal": dict(cfg=dict(ninputs=2, noutputs=2, nblocks=3, channel_width=16, kernel_size=5,
                                    stack_size=2, dilation_growth=3, causal=True), B=2, T=1024, seed=5),
    }
    for name, c in cases.items():
        cfg = c["cfg"]
        net = TCN(**cfg).eval()
        sd = ref_tcn.tcn_init_state_dict(cfg["ninputs"], cfg["noutputs"], cfg["nblocks"],
                                         cfg["channel_width"], cfg["kernel_size"], seed=c["seed"])
        for k in [k for k in sd if k.endswith("relu.weight")]:   # non-trivial PReLU slopes
            sd[k] = torch.linspace(0.05, 0.45, sd[k].numel())
        net.load_state_dict(sd)
        g = torch.Generator().manual_seed(100 + c["seed"])
        x = torch.randn(c["B"], cfg["ninputs"], c["T"], generator=g)
        with torch.no_grad():
            y = net(x)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), x=x.numpy(), y=y.numpy(),
                            rf=np.int64(net.receptive_field), seed=np.int64(c["seed"]),
                            **{"cfg_" + k: np.asarray(v) for k, v in cfg.items()})
        print(name, tuple(y.shape), "rf", net.receptive_field)
    # full config: receptive field, parameter count, key list (cheap known answers)
    import yaml
    full = yaml.safe_load(open(os.path.join(REF, "cfg/model/tcn.yaml")))["model"]["network"]
    for k in ("_target_", "sample_rate", "num_bins"):
        full.pop(k)
    net = TCN(**full)
    np.savez_compressed(os.path.join(OUT, "tcn_full_kat.npz"), rf=np.int64(net.receptive_field),
                        nparams=np.int64(sum(p.numel() for p in net.parameters())),
                        keys=np.array(list(net.state_dict().keys())))
    print("tcn_full rf", net.receptive_field, sum(p.numel() for p in net.parameters()))


def gen_tcn_full():
    """The FULL cfg/model/tcn.yaml network (20 blocks x 256 channels, k = 7; 9 974 017 parameters) of the imported
    reference (remfx/tcn.py:62-138) on one 32768-sample clip: forward output (all 20491 samples), and the gradient of
    sum(y * r) for strided slices of eight parameters spread over the depth -- pins BASELINE config 2 at full width
    against the reference itself (forward AND autograd backward).  Weights: oracle's seeded generator (40 MB, not stored)."""
    import yaml
    from remfx.tcn import TCN
    full = yaml.safe_load(open(os.path.join(REF, "cfg/model/tcn.yaml")))["model"]["network"]
    for k in ("_target_", "sample_rate", "num_bins"):
        full.pop(k)
    net = TCN(**full).eval()
    sd = ref_tcn.tcn_init_state_dict(1, 1, 20, 256, 7, seed=31)
    for i, k in enumerate([k for k in sd if k.endswith("relu.weight")]):   # non-trivial PReLU slopes, different per block
        sd[k] = torch.linspace(0.05, 0.45, sd[k].numel()).roll(7 * i)
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(131)
    T = 32768
    x = torch.randn(1, 1, T, generator=g) * 0.5
    y = net(x)
    r = torch.randn(y.shape, generator=g)
    (y * r).sum().backward()
    params = dict(net.named_parameters())
    names = ["process_blocks.0.conv1.weight", "process_blocks.0.res.weight", "process_blocks.3.conv1.weight",
             "process_blocks.9.conv1.bias", "process_blocks.10.relu.weight", "process_blocks.12.res.weight",
             "process_blocks.19.conv1.weight", "output.weight"]
    rec = {}
    for n in names:
        gr = params[n].grad.reshape(-1)
        rec["gnorm_" + n] = np.float64(gr.double().norm())
        rec["gslice_" + n] = gr[:: max(1, gr.numel() // 512)][:512].numpy().copy()
    gtot = torch.sqrt(sum(p.grad.double().pow(2).sum() for p in net.parameters()))
    np.savez_compressed(os.path.join(OUT, "tcn_full_fwd_bwd.npz"), seed=np.int64(31), x_seed=np.int64(131), T=np.int64(T),
                        y=y.detach().numpy(), grad_names=np.array(names), grad_total_norm=np.float64(gtot), **rec)
    print("tcn_full_fwd_bwd", tuple(y.shape), "y rms", float(y.pow(2).mean().sqrt()), "gnorm", float(gtot))


def gen_tcn_full_length():
    """BASELINE config 2 at its REAL shape: the full cfg/model/tcn.yaml network of the imported reference (remfx/tcn.py:62-138) on one
    262144-sample clip, forward only (the reference takes ~16 s per clip on the build container's cores): strided output slices and
    the output norm -> tests/golden/tcn_full_length_fwd.npz.  Same seeded weights as gen_tcn_full."""
    import yaml
    from remfx.tcn import TCN
    full = yaml.safe_load(open(os.path.join(REF, "cfg/model/tcn.yaml")))["model"]["network"]
    for k in ("_target_", "sample_rate", "num_bins"):
        full.pop(k)
    net = TCN(**full).eval()
    sd = ref_tcn.tcn_init_state_dict(1, 1, 20, 256, 7, seed=31)
    for i, k in enumerate([k for k in sd if k.endswith("relu.weight")]):
        sd[k] = torch.linspace(0.05, 0.45, sd[k].numel()).roll(7 * i)
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(132)
    T = 262144
    x = torch.randn(1, 1, T, generator=g) * 0.5
    with torch.no_grad():
        y = net(x)
    yf = y.reshape(-1)
    np.savez_compressed(os.path.join(OUT, "tcn_full_length_fwd.npz"), seed=np.int64(31), x_seed=np.int64(132), T=np.int64(T),
                        y_len=np.int64(yf.numel()), y_stride=np.int64(61), y_slice=yf[::61].numpy().copy(),
                        y_head=yf[:2048].numpy().copy(), y_tail=yf[-2048:].numpy().copy(), y_norm=np.float64(yf.double().norm()))
    print("tcn_full_length_fwd", tuple(y.shape), "y rms", float(y.pow(2).mean().sqrt()))


def gen_cnn14():
    from remfx.classifier import Cnn14
    net = Cnn14(num_classes=5, sample_rate=48000, model_sample_rate=48000, n_fft=2048,
                hop_length=512, n_mels=128).eval()
    sd = ref_cnn14.cnn14_init_state_dict(seed=7)
    # BN with non-trivial affine + running stats so eval mode is exercised
    g = torch.Generator().manual_seed(8)
    for k in list(sd):
        if k.endswith("running_mean"):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.05
        elif k.endswith("running_var"):
            sd[k] = torch.rand(sd[k].shape, generator=g) * 0.5 + 0.75
        elif ".bn" in k and k.endswith("weight"):
            sd[k] = torch.rand(sd[k].shape, generator=g) * 0.4 + 0.8
        elif ".bn" in k and k.endswith("bias"):
            sd[k] = torch.randn(sd[k].shape, generator=g) * 0.05
    missing = net.load_state_dict(sd, strict=False)
    assert not missing.unexpected_keys, missing
    for k in list(sd):                       # spread the head logits away from 0
        if k.startswith("heads.") and k.endswith("weight"):
            sd[k] = sd[k] * 40.0
    net.load_state_dict(sd, strict=False)
    t = torch.arange(32768) / 48000.0
    x = torch.stack([torch.randn(32768, generator=g) * 0.1,
                     0.3 * torch.sin(2 * torch.pi * (200.0 + 4000.0 * t) * t) +
                     0.1 * torch.sin(2 * torch.pi * 9000.0 * t)])[:, None, :]
    taps = {}
    hooks = [net.conv_block2.register_forward_hook(lambda m, i, o: taps.__setitem__("cb2", o.detach().clone())),
             net.fc1.register_forward_hook(lambda m, i, o: taps.__setitem__("fc1", o.detach().clone()))]
    with torch.no_grad():
        mel = net.melspec(x)
        out_eval = torch.hstack(net(x))
        cb2_eval, fc1_eval = taps["cb2"], taps["fc1"]
        net.train()     # BatchNorm batch statistics; dropout off because train=False arg
        out_bnbatch = torch.hstack(net(x, train=False))
    for h in hooks:
        h.remove()
    np.savez_compressed(os.path.join(OUT, "cnn14_full.npz"), x=x.numpy(), mel=mel.numpy(),
                        out_eval=out_eval.numpy(), out_bnbatch=out_bnbatch.numpy(),
                        cb2_eval=cb2_eval[:, ::16, ::4, ::8].numpy(), fc1_eval=fc1_eval[:, ::8].numpy(),
                        nparams=np.int64(sum(p.numel() for p in net.parameters())),
                        keys=np.array(list(net.state_dict().keys())))
    print("cnn14_full", out_eval.numpy().round(4), out_bnbatch.numpy().round(4))


def gen_flow():
    """Control flow of the reference wrappers themselves (remfx/models.py run verbatim over the
    oracle's loss restatements): RemFX.common_step logging + crops, RemFXChainInference threshold /
    ordering / per-clip chains."""
    sys.modules["remfx.effects"] = _mod("remfx.effects", Pedalboard_Effects=[
        type(n, (), {}) for n in ("RandomPedalboardReverb", "RandomPedalboardChorus", "RandomPedalboardDelay",
                                  "RandomPedalboardDistortion", "RandomPedalboardCompressor")])
    import remfx.models as rm
    # 1. one training step of RemFX(TCNModel) -- logged names / values and the loss
    cfg = dict(ninputs=1, noutputs=1, nblocks=3, channel_width=8, kernel_size=7, stack_size=10,
               dilation_growth=2, causal=False)
    net = rm.TCNModel(sample_rate=48000, num_bins=1025, **cfg)
    sd = ref_tcn.tcn_init_state_dict(1, 1, 3, 8, 7, seed=21)
    net.model.load_state_dict(sd)
    model = rm.RemFX(1e-4, 0.95, 0.999, 1e-6, 1e-3, 48000, net)
    g = torch.Generator().manual_seed(22)
    x = torch.randn(2, 1, 12000, generator=g) * 0.3
    y = x + 0.05 * torch.randn(2, 1, 12000, generator=g)
    loss = model.training_step((x, y, None, None), 0)
    logged = dict(model.logged)
    # 2. chain inference with a fake classifier and tagged "removal models"
    class Tag(nn.Module):
        def __init__(self, mul, add):
            super().__init__(); self.mul, self.add = mul, add
        def sample(self, z):
            return z * self.mul + self.add
    class Holder(nn.Module):
        def __init__(self, m):
            super().__init__(); self.model = m
    names = ["RandomPedalboardReverb", "RandomPedalboardChorus", "RandomPedalboardDelay",
             "RandomPedalboardDistortion", "RandomPedalboardCompressor"]
    models = {n: Holder(Tag(1.0 + 0.1 * (i + 1), 0.01 * (i + 1))) for i, n in enumerate(names)}
    probs = torch.tensor([[0.9, 0.5, 0.2, 0.7, 0.3], [0.1, 0.6, 0.2, 0.4, 0.51], [0.0, 0.0, 0.0, 0.0, 0.0]])
    class FakeCls(nn.Module):
        def forward(self, z):
            return [probs[:, k:k + 1] for k in range(5)]
    order = ["RandomPedalboardDistortion", "RandomPedalboardCompressor", "RandomPedalboardReverb",
             "RandomPedalboardChorus", "RandomPedalboardDelay"]       # cfg/exp/remfx_detect.yaml:80-85
    chain = rm.RemFXChainInference(models, 48000, 1025, order, classifier=FakeCls())
    xc = torch.randn(3, 1, 9000, generator=g) * 0.2
    yc = torch.randn(3, 1, 9000, generator=g) * 0.2
    closs, cout = chain.forward((xc, yc, None, None), 0)
    chain.test_step((xc, yc, None, None), 0)
    np.savez_compressed(os.path.join(OUT, "flow.npz"), x=x.numpy(), y=y.numpy(), loss=np.float32(loss.item()),
                        log_names=np.array(sorted(logged)), log_vals=np.array([logged[k] for k in sorted(logged)], dtype=np.float32),
                        probs=probs.numpy(), xc=xc.numpy(), yc=yc.numpy(), chain_out=cout.numpy(),
                        chain_loss=np.float32(closs.item()), chain_log_names=np.array(sorted(chain.logged)),
                        chain_log_vals=np.array([chain.logged[k] for k in sorted(chain.logged)], dtype=np.float32))
    print("flow", sorted(logged), float(loss), sorted(chain.logged))


def tiny_heads_state(seed=5):
    """Weights of the 5-head stand-in network of the mixup fixture (three waveform statistics -> sigmoid(linear) per head)."""
    g = torch.Generator().manual_seed(seed)
    return {"w": torch.randn(5, 3, generator=g), "b": 0.1 * torch.randn(5, generator=g)}


def tiny_heads_forward(x, w, b):
    """x (B, 1, T) -> list of 5 (B, 1) probabilities; shared by the fixture generator and tests/ (deterministic, no dropout)."""
    f = torch.stack([x.abs().mean((1, 2)), x.pow(2).mean((1, 2)).sqrt(), x.amax((1, 2))], 1) * 4.0      # (B, 3)
    z = f @ w.t() + b
    return [torch.sigmoid(z[:, k:k + 1]) for k in range(5)]


def gen_mixup():
    """`mixup` and the mixup branch of FXClassifier.common_step (remfx/models.py:393-420, 491-500) of the imported reference,
    seeded: numpy draws lambda ~ U(0.25, 0.75) per item, then ONE numpy uniform decides whether to mix, then torch.randperm picks
    the partners; labels are OR-ed.  Several seeds so that both branches occur.  The network is a deterministic 5-head stand-in
    subclassing the reference's Cnn14 (the reference tests isinstance(network, Cnn14) to pick BCELoss + per-effect accuracy)."""
    sys.modules.setdefault("remfx.effects", _mod("remfx.effects", Pedalboard_Effects=[
        type(n, (), {}) for n in ("RandomPedalboardReverb", "RandomPedalboardChorus", "RandomPedalboardDelay",
                                  "RandomPedalboardDistortion", "RandomPedalboardCompressor")]))
    import remfx.models as rm
    import remfx.classifier as rc
    rec = {}
    g = torch.Generator().manual_seed(31)
    x = torch.randn(6, 1, 256, generator=g) * 0.3
    y = (torch.rand(6, 5, generator=g) > 0.6).float()
    rec["x"], rec["y"] = x.numpy(), y.numpy()
    mixed_any = plain_any = False
    seeds = list(range(8))
    for s in seeds:
        np.random.seed(s)
        torch.manual_seed(s)
        mx, my, lam = rm.mixup(x, y)
        did = not (mx is x)
        mixed_any |= did
        plain_any |= not did
        rec[f"mx{s}"], rec[f"my{s}"], rec[f"lam{s}"], rec[f"did{s}"] = mx.numpy(), my.numpy(), lam.numpy(), np.bool_(did)
    assert mixed_any and plain_any
    rec["seeds"] = np.array(seeds)

    class Tiny(rc.Cnn14):
        def __init__(self, st):
            nn.Module.__init__(self)
            self.w, self.b = nn.Parameter(st["w"].clone()), nn.Parameter(st["b"].clone())

        def forward(self, z, train=False):
            return tiny_heads_forward(z, self.w, self.b)
    st = tiny_heads_state()
    for s in (1, 4):                                  # one seed of each branch (asserted below)
        net = Tiny(st)
        cls = rm.FXClassifier(3e-4, 1e-3, 48000, net, mixup=True)
        np.random.seed(s)
        torch.manual_seed(s)
        loss = cls.training_step((x, None, None, y), 0)
        loss.backward()
        rec[f"cls_loss{s}"] = np.float32(loss.item())
        rec[f"cls_gw{s}"], rec[f"cls_gb{s}"] = net.w.grad.numpy(), net.b.grad.numpy()
        names = sorted(cls.logged)
        rec[f"cls_log_names{s}"] = np.array(names)
        rec[f"cls_log_vals{s}"] = np.array([cls.logged[k] for k in names], dtype=np.float32)
    assert bool(rec["did1"]) != bool(rec["did4"]), (rec["did1"], rec["did4"])
    np.savez_compressed(os.path.join(OUT, "mixup.npz"), **rec)
    print("mixup", [bool(rec[f"did{s}"]) for s in seeds], float(rec["cls_loss1"]), float(rec["cls_loss4"]))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    install_shims()
    only = sys.argv[1:]                      # e.g. `python oracle/gen_golden.py tcn_full` regenerates one fixture
    for name, fn in (("utils", gen_utils), ("tcn", gen_tcn), ("tcn_full", gen_tcn_full), ("tcn_full_length", gen_tcn_full_length),
                     ("cnn14", gen_cnn14),
                     ("flow", gen_flow), ("mixup", gen_mixup)):
        if not only or name in only:
            fn()

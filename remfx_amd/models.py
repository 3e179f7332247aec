"""Host-side mirror of remfx.models for the effect-removal hot path.

Same class names, constructor arguments, step semantics, logged metric names and
state_dict layout as reference remfx/models.py (RemFX :152-256, TCNModel :370-390,
DemucsModel :307-324, RemFXChainInference :22-149), so the Hydra ``_target_``
strings of cfg/model/*.yaml resolve to these classes.  The
arithmetic (networks, losses, metrics) runs on the HIP kernels; this file is
orchestration only.  pytorch_lightning is optional: without it the classes are
plain nn.Modules driven by remfx_amd.trainer.
"""
import os
import random

import torch
from torch import Tensor, nn

from .losses import L1Loss, MultiResolutionSTFTLoss, SISDRLoss, stft_memo
from .tcn import TCN
from .utils import causal_crop

try:  # the reference derives from pl.LightningModule; keep that when lightning is installed
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # not installed in this image
    pl = None

    class _Base(nn.Module):
        """Minimal LightningModule surface used by the reference code paths."""

        def __init__(self):
            super().__init__()
            self.logged = {}
            self.trainer = None

        def log(self, name, value, **kwargs):
            self.logged[name] = value.detach() if torch.is_tensor(value) else value


# On by default: A/B 0 = the Input_* metrics after the network on the step's stream.  (It was off for the last hours of round 5, until
# the GroupNorm statistics of the forward pass stopped depending on a zero fill + atomics: DESIGN.md 4.10.)
METRIC_STREAM = os.environ.get("RFX_METRIC_STREAM", "1") != "0"
_METRIC_STREAMS = {}


def _metric_stream(device):
    idx = device.index if device.index is not None else torch.cuda.current_device()
    st = _METRIC_STREAMS.get(idx)
    if st is None:
        st = torch.cuda.Stream(device=idx)           # default priority: fills the gaps of the (high-priority) compute stream
        _METRIC_STREAMS[idx] = st
    return st


# label order of effects.Pedalboard_Effects (effects.py:699-707); class NAMES are the dict keys
ALL_EFFECT_NAMES = ["RandomPedalboardReverb", "RandomPedalboardChorus", "RandomPedalboardDelay",
                    "RandomPedalboardDistortion", "RandomPedalboardCompressor"]


class RemFX(_Base):
    def __init__(self, lr: float, lr_beta1: float, lr_beta2: float, lr_eps: float, lr_weight_decay: float,
                 sample_rate: float, network: nn.Module):
        super().__init__()
        self.lr, self.lr_beta1, self.lr_beta2 = lr, lr_beta1, lr_beta2
        self.lr_eps, self.lr_weight_decay, self.sample_rate = lr_eps, lr_weight_decay, sample_rate
        self.model = network
        self.metrics = nn.ModuleDict({"SISDR": SISDRLoss(), "STFT": MultiResolutionSTFTLoss()})
        self.log_train_audio = True
        self.output_str = "IN_SISDR,OUT_SISDR,IN_STFT,OUT_STFT\n"

    @property
    def device(self):
        return next(self.model.parameters()).device

    def configure_optimizers(self):
        """models.py:185-206.  Returns the same structure; the optimiser is the flat HIP AdamW
        when the parameters live on the GPU."""
        from .optim import FlatAdamW, FlatParams, MultiStepLR
        # memory order = forward-use order where the network knows one that differs from its registration order (Hybrid Demucs): the
        # gradient exchange then overlaps backward bucket by bucket (optim.FlatParams, ddp.GradSync)
        inner = getattr(self.model, "model", None)
        layout = inner.forward_use_order() if hasattr(inner, "forward_use_order") else None
        flat = FlatParams(list(self.model.parameters()), layout=layout)
        optimizer = FlatAdamW(flat, lr=self.lr, betas=(self.lr_beta1, self.lr_beta2), eps=self.lr_eps,
                              weight_decay=self.lr_weight_decay)
        max_steps = self.trainer.max_steps if self.trainer is not None else 50000
        sched = MultiStepLR(optimizer, [0.8 * max_steps, 0.95 * max_steps], gamma=0.1)
        return {"optimizer": optimizer,
                "lr_scheduler": {"scheduler": sched, "monitor": "val_loss", "interval": "step", "frequency": 1}}

    def training_step(self, batch, batch_idx):
        return self.common_step(batch, batch_idx, mode="train")

    def validation_step(self, batch, batch_idx):
        return self.common_step(batch, batch_idx, mode="valid")

    def test_step(self, batch, batch_idx):
        return self.common_step(batch, batch_idx, mode="test")

    def common_step(self, batch, batch_idx, mode: str = "train"):
        x, y, _, _ = batch                                     # (B, C, T) each
        with stft_memo():                                      # the loss and both STFT metrics share their spectra
            # The Input_* metrics depend on the batch only: on the GPU they are enqueued on their own stream BEFORE the network, so
            # their analyses (three MRSTFT resolutions + the SI-SDR sums, ~1.4 ms at 64 clips, HBM-class) run beside the forward pass
            # instead of after it.  Values and logging order are unchanged.
            early = None
            if METRIC_STREAM and x.is_cuda and len(self.metrics):
                main_s, met_s = torch.cuda.current_stream(), _metric_stream(x.device)
                met_s.wait_stream(main_s)
                with torch.cuda.stream(met_s), torch.no_grad():
                    early = {m: (-1 if m == "SISDR" else 1) * self.metrics[m](x, y) for m in self.metrics}
            loss, output = self.model((x, y))
            target = y
            if output.shape[-1] < y.shape[-1]:                 # models.py:222-224
                target = causal_crop(y, output.shape[-1])
            self.log(f"{mode}_loss", loss)
            if early is not None:
                main_s.wait_stream(met_s)
                for v in early.values():
                    v.record_stream(main_s)
            with torch.no_grad():
                for metric in self.metrics:
                    negate = -1 if metric == "SISDR" else 1    # SISDR loss is -SI-SDR
                    self.log(f"{mode}_{metric}", negate * self.metrics[metric](output.detach(), target),
                             on_step=False, on_epoch=True, logger=True, prog_bar=True, sync_dist=True)
                    self.log(f"Input_{metric}", early[metric] if early is not None else negate * self.metrics[metric](x, y),
                             on_step=False, on_epoch=True, logger=True, prog_bar=True, sync_dist=True)
        return loss


class _RemovalWrapper(nn.Module):
    """forward((x, target)) -> (loss, output); sample(x) -> output; loss = MRSTFT + 100 * L1."""

    def _loss(self, output, target):
        return self.mrstftloss(output, target) + self.l1loss(output, target) * 100


class TCNModel(_RemovalWrapper):
    def __init__(self, sample_rate, num_bins, **kwargs):
        super().__init__()
        self.model = TCN(**kwargs)
        self.mrstftloss = MultiResolutionSTFTLoss(n_bins=num_bins, sample_rate=sample_rate)
        self.l1loss = L1Loss()

    def forward(self, batch):
        x, target = batch
        output = self.model(x)                                 # B x 1 x T'
        if output.shape[-1] < target.shape[-1]:                # models.py:383-384
            target = causal_crop(target, output.shape[-1])
        return self._loss(output, target), output

    def sample(self, x: Tensor) -> Tensor:
        return self.model(x)


class DemucsModel(_RemovalWrapper):
    def __init__(self, sample_rate, **kwargs) -> None:
        super().__init__()
        from .hdemucs import HDemucs
        self.model = HDemucs(**kwargs)
        self.num_bins = kwargs["nfft"] // 2 + 1
        self.mrstftloss = MultiResolutionSTFTLoss(n_bins=self.num_bins, sample_rate=sample_rate)
        self.l1loss = L1Loss()

    def forward(self, batch):
        x, target = batch
        output = self.model(x).squeeze(1)
        return self._loss(output, target), output

    def sample(self, x: Tensor) -> Tensor:
        return self.model(x).squeeze(1)


class OpenUnmixModel(_RemovalWrapper):
    def __init__(self, n_fft: int = 2048, hop_length: int = 512, n_channels: int = 1, alpha: float = 0.3,
                 sample_rate: int = 22050):
        super().__init__()
        from .umx import OpenUnmix, Separator
        from .utils import spectrogram
        self._spectrogram = spectrogram
        self.n_channels, self.n_fft, self.hop_length, self.alpha = n_channels, n_fft, hop_length, alpha
        self.register_buffer("window", torch.hann_window(n_fft))
        self.num_bins = n_fft // 2 + 1
        self.sample_rate = sample_rate
        self.model = OpenUnmix(nb_channels=n_channels, nb_bins=self.num_bins)
        self.separator = Separator(target_models={"other": self.model}, nb_channels=n_channels,
                                   sample_rate=sample_rate, n_fft=n_fft, n_hop=hop_length)
        self.mrstftloss = MultiResolutionSTFTLoss(n_bins=self.num_bins, sample_rate=sample_rate)
        self.l1loss = L1Loss()

    def forward(self, batch):
        x, target = batch
        X = self._spectrogram(x, self.window, self.n_fft, self.hop_length, self.alpha)
        Y = self.model(X)  # noqa: F841  dead value kept: it updates BN running stats / draws dropout (Q3)
        sep_out = self.separator(x).squeeze(1)
        return self._loss(sep_out, target), sep_out

    def sample(self, x: Tensor) -> Tensor:
        return self.separator(x).squeeze(1)


class DCUNetModel(_RemovalWrapper):
    def __init__(self, sample_rate, num_bins, **kwargs):
        super().__init__()
        from .dcunet import DCUNet
        self.model = DCUNet(**kwargs)          # asteroid keeps its own sample_rate default (wrapper swallows it)
        self.mrstftloss = MultiResolutionSTFTLoss(n_bins=num_bins, sample_rate=sample_rate)
        self.l1loss = L1Loss()

    def forward(self, batch):
        x, target = batch
        output = self.model(x.squeeze(1))                      # (B, 1, T)
        if output.shape[-1] < target.shape[-1]:                # models.py:360-361
            target = causal_crop(target, output.shape[-1])
        return self._loss(output, target), output

    def sample(self, x: Tensor) -> Tensor:
        return self.model(x.squeeze(1))


class DPTNetModel(_RemovalWrapper):
    """models.py:327-344: asteroid DPTNet (cfg/model/dptnet.yaml) + MRSTFT + 100 L1; no target crop (the network pads / crops
    its output to the input length)."""

    def __init__(self, sample_rate, num_bins, **kwargs):
        super().__init__()
        from .dptnet import DPTNet
        self.model = DPTNet(**kwargs)
        self.num_bins = num_bins
        self.mrstftloss = MultiResolutionSTFTLoss(n_bins=num_bins, sample_rate=sample_rate)
        self.l1loss = L1Loss()

    def forward(self, batch):
        x, target = batch
        output = self.model(x.squeeze(1))                      # (B, 1, T)
        return self._loss(output, target), output

    def sample(self, x: Tensor) -> Tensor:
        return self.model(x.squeeze(1))


def mixup(x: torch.Tensor, y: torch.Tensor, alpha: float = 1.0):
    """Mixup for time-domain clips, reference `remfx/models.py:393-420`.  The RANDOM-DRAW ORDER is the contract (the fixture
    `tests/golden/mixup.npz` replays it from the imported reference): (1) numpy: one weight per clip, lambda ~ U(0.25, 0.75)
    (only when alpha > 0; otherwise lambda = 1); (2) numpy: one uniform, mix iff it exceeds 0.5; (3) torch (CPU generator):
    `randperm(B)` picks each clip's partner -- drawn only on the mixing path.  A mixed clip is `lambda x + (1 - lambda) x[partner]`,
    its label vector the logical OR of the two label vectors (not a lambda blend).  Returns
    `(clips, labels, lambda)`; on the non-mixing path the inputs come back as they are."""
    import numpy as np
    n = x.size(0)
    weights = 1
    if alpha > 0:
        weights = torch.from_numpy(np.random.uniform(0.25, 0.75, n)).float().to(x.device).view(n, 1, 1)
    if not np.random.rand() > 0.5:
        return x, y, weights
    partner = torch.randperm(n).to(x.device)
    blended = weights * x + (1 - weights) * x[partner, :]
    either = torch.logical_or(y, y[partner, :]).float()
    return blended, either, weights


def _multilabel_f1(probs, target, threshold=0.5):
    """torchmetrics MultilabelF1Score(num_labels, threshold, average="none", multidim_average="global") on one batch:
    per-label 2 TP / (2 TP + FP + FN), 0 where the denominator is 0 (models.py:457-476 metrics)."""
    pred = (probs > threshold).long()
    tgt = target.long()
    tp = (pred * tgt).sum(0).float()
    fp = (pred * (1 - tgt)).sum(0).float()
    fn = ((1 - pred) * tgt).sum(0).float()
    den = 2 * tp + fp + fn
    return torch.where(den > 0, 2 * tp / den.clamp_min(1.0), torch.zeros_like(den))


class FXClassifier(_Base):
    """models.py:423-592.  Cnn14 network: sum over the 5 heads of BCELoss, per-effect binary accuracy.  Any other network
    (the HEAR-embedding classifiers, classifier.py:16-128: one (B, 5) logit tensor): CrossEntropyLoss(label_smoothing) against
    the wet-label vector and per-effect / macro multilabel F1 of the sigmoid outputs (models.py:457-476, 503-506, 540-570)."""

    def __init__(self, lr: float, lr_weight_decay: float, sample_rate: float, network: nn.Module,
                 mixup: bool = False, label_smoothing: float = 0.0):
        super().__init__()
        from .classifier import Cnn14
        self.lr, self.lr_weight_decay, self.sample_rate = lr, lr_weight_decay, sample_rate
        self.network = network
        self.effects = ["Reverb", "Chorus", "Delay", "Distortion", "Compressor"]
        self.mixup, self.label_smoothing = mixup, label_smoothing     # label_smoothing unused for Cnn14 (Q12)
        self.multihead = isinstance(network, Cnn14)
        self.loss_fn = torch.nn.BCELoss() if self.multihead else torch.nn.CrossEntropyLoss(label_smoothing=label_smoothing)

    def forward(self, x: torch.Tensor, train: bool = False):
        return self.network(x, train=train)

    def common_step(self, batch, batch_idx, mode: str = "train"):
        train = mode == "train"
        x, y, dry_label, wet_label = batch
        labels = wet_label
        mixed = train and self.mixup
        if mixed:
            x, labels, _ = mixup(x, wet_label)
        outputs = self(x, train)
        if self.multihead or mixed:                 # models.py:496-500 iterates `outputs` in the mixup branch whatever the network
            loss = 0
            for idx, output in enumerate(outputs):
                loss = loss + self.loss_fn(output.squeeze(-1), labels[..., idx])
        else:
            loss = self.loss_fn(outputs, labels)    # (B, 5) logits against the (B, 5) float label vector (class probabilities)
        self.log(f"{mode}_loss", loss, on_step=True, on_epoch=True, prog_bar=True, logger=True, sync_dist=True)
        with torch.no_grad():
            if self.multihead:
                accs = []
                for idx, name in enumerate(self.effects[:len(outputs)]):
                    acc = ((outputs[idx].squeeze(-1) > 0.5).float() == wet_label[..., idx]).float().mean()
                    self.log(f"{mode}_{name}_acc", acc, on_step=True, on_epoch=True, prog_bar=True, logger=True,
                             sync_dist=True)
                    accs.append(acc)
                self.log(f"{mode}_avg_acc", torch.mean(torch.stack(accs)), on_step=True, on_epoch=True,
                         prog_bar=True, logger=True, sync_dist=True)
            else:
                f1 = _multilabel_f1(torch.sigmoid(outputs), wet_label)
                for idx, name in enumerate(self.effects):
                    self.log(f"{mode}_f1_{name}", f1[idx], on_step=True, on_epoch=True, prog_bar=True, logger=True,
                             sync_dist=True)
                self.log(f"{mode}_avg_acc", f1.mean(), on_step=True, on_epoch=True, prog_bar=True, logger=True,
                         sync_dist=True)              # macro F1 under the reference's name
        return loss

    def training_step(self, batch, batch_idx):
        return self.common_step(batch, batch_idx, mode="train")

    def validation_step(self, batch, batch_idx):
        return self.common_step(batch, batch_idx, mode="valid")

    def test_step(self, batch, batch_idx):
        return self.common_step(batch, batch_idx, mode="test")

    def configure_optimizers(self):
        """models.py:586-592: AdamW(lr, weight_decay), default betas / eps, no scheduler."""
        from .optim import FlatAdamW, FlatParams
        return FlatAdamW(FlatParams(list(self.network.parameters())), lr=self.lr, betas=(0.9, 0.999), eps=1e-8,
                         weight_decay=self.lr_weight_decay)


class RemFXChainInference(_Base):
    """Classifier -> threshold -> per-clip ordered chain of effect-removal models
    (models.py:22-149).  Clips that share a detected-effect signature are batched
    together per removal model (numerically identical for GroupNorm-only networks;
    SURVEY 3.3), instead of the reference's batch-1 python loop."""

    def __init__(self, models, sample_rate, num_bins, effect_order, classifier=None,
                 shuffle_effect_order=False, use_all_effect_models=False):
        super().__init__()
        self.model = models                                     # plain dict, as upstream (Q6)
        self.mrstftloss = MultiResolutionSTFTLoss(n_bins=num_bins, sample_rate=sample_rate)
        self.l1loss = L1Loss()
        self.metrics = nn.ModuleDict({"SISDR": SISDRLoss(), "STFT": MultiResolutionSTFTLoss()})
        self.sample_rate = sample_rate
        self.effect_order = effect_order
        self.classifier = classifier
        self.shuffle_effect_order = shuffle_effect_order
        self.output_str = "IN_SISDR,OUT_SISDR,IN_STFT,OUT_STFT\n"
        self.use_all_effect_models = use_all_effect_models
        self.last_labels = None

    def forward(self, batch, batch_idx, order=None, verbose=False):
        x, y, _, rem_fx_labels = batch
        effects_order = order if order else self.effect_order
        if self.classifier:
            with torch.no_grad():
                labels = torch.hstack(self.classifier(x))
                rem_fx_labels = torch.where(labels > 0.5, 1.0, 0.0)    # strict >, models.py:61-64
        self.last_labels = rem_fx_labels
        lab = rem_fx_labels.detach().cpu()
        if self.use_all_effect_models:
            present = [list(ALL_EFFECT_NAMES) for _ in range(x.shape[0])]
        else:
            present = [[ALL_EFFECT_NAMES[i] for i, e in enumerate(row) if float(e) == 1.0] for row in lab]
        if verbose and present:
            print("Detected effects:", present[0])
            print("Removing effects...")
        chains = [tuple(e for e in effects_order if e in names) for names in present]
        output = x.clone()
        with torch.no_grad():
            # group clips by their remaining chain; run each removal model on sub-batches
            todo = {i: list(c) for i, c in enumerate(chains)}
            while any(todo.values()):
                groups = {}
                for i, c in todo.items():
                    if c:
                        groups.setdefault(c[0], []).append(i)
                for effect, idxs in groups.items():
                    sel = torch.tensor(idxs, device=x.device)
                    res = self.model[effect].model.sample(output.index_select(0, sel))
                    output.index_copy_(0, sel, res)
                    for i in idxs:
                        todo[i].pop(0)
        loss = self.mrstftloss(output, y) + self.l1loss(output, y) * 100
        return loss, output

    def test_step(self, batch, batch_idx):
        x, y, _, _ = batch
        if self.shuffle_effect_order:
            random.shuffle(self.effect_order)                    # in place, as upstream (Q7)
        loss, output = self.forward(batch, batch_idx, order=self.effect_order)
        if output.shape[-1] < y.shape[-1]:
            y = causal_crop(y, output.shape[-1])
        self.log("test_loss", loss)
        with torch.no_grad():
            for metric in self.metrics:
                negate = -1 if metric == "SISDR" else 1
                self.log(f"test_{metric}", negate * self.metrics[metric](output, y), on_step=False,
                         on_epoch=True, logger=True, prog_bar=True, sync_dist=True)
                self.log(f"Input_{metric}", negate * self.metrics[metric](x, y), on_step=False,
                         on_epoch=True, logger=True, prog_bar=True, sync_dist=True)
        return loss

    def sample(self, batch):
        return self.forward(batch, 0)[1]

"""Data module of the removal path: the reference's loader surface (remfx/datasets.py:333-673) over
  * rendered clips on disk in the reference's layout (datasets.py:370-380, 445-468):
        {render_root}/processed/{effects_string}/{mode}/{idx}/{input.wav, target.wav, dry_effects.pt, wet_effects.pt}
  * or seeded white noise when no corpus / rendered set is reachable (BASELINE.json's configs).

What the hot path consumes is the batch tuple  (x_wet, y_dry, dry_labels, wet_labels)  with shapes
(B,1,T), (B,1,T), (B,5), (B,5) (datasets.py:461-468).  RENDERING clips (pedalboard / sox / pyloudnorm on the
CPU, datasets.py:399-452) is SURVEY 8(f) rank 3 and is not done here: a dataset that would have to render says so.

The classes take the reference's constructor arguments, so ``cfg/config.yaml``'s datamodule node instantiates
unchanged.  Multi-GPU: the loaders shard by rank with a DistributedSampler (what Lightning injects for the
reference), see ``EffectDatamodule._dl``.
"""
import os
import warnings
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

NUM_EFFECTS = 5      # len(effects.Pedalboard_Effects)


def load_wav(path):
    """(channels, samples) float32 in [-1, 1) and the sample rate: what torchaudio.load returns for the PCM16 /
    PCM32 / float32 WAV files the reference writes with torchaudio.save (datasets.py:447-448)."""
    from scipy.io import wavfile
    sr, a = wavfile.read(str(path))
    if a.ndim == 1:
        a = a[:, None]
    if a.dtype == np.int16:
        x = a.astype(np.float32) / 32768.0
    elif a.dtype == np.int32:
        x = a.astype(np.float32) / 2147483648.0
    elif a.dtype == np.uint8:
        x = (a.astype(np.float32) - 128.0) / 128.0
    else:
        x = a.astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(x.T)), int(sr)


def save_wav(path, x, sample_rate):
    """float32 WAV of a (channels, samples) tensor (torchaudio.save's default for float tensors)."""
    from scipy.io import wavfile
    wavfile.write(str(path), int(sample_rate), x.detach().cpu().to(torch.float32).numpy().T.copy())


def _usable_dir(p):
    return p is not None and "<unset env" not in str(p) and os.path.isdir(str(p))


class SyntheticEffectDataset(Dataset):
    """Seeded white noise at the level the dataset normalises to (about -20 dB), random wet labels."""

    def __init__(self, total_chunks=8, chunk_size=262144, seed=12345, num_classes=NUM_EFFECTS, level=0.1, **_):
        self.n, self.t, self.seed, self.k, self.level = total_chunks, chunk_size, seed, num_classes, level

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed + idx)
        x = torch.randn(1, self.t, generator=g) * self.level
        y = torch.randn(1, self.t, generator=g) * self.level
        wet = (torch.rand(self.k, generator=g) > 0.5).float()
        return x, y, torch.zeros(self.k), wet


class EffectDataset(Dataset):
    """remfx.datasets.EffectDataset (datasets.py:333-468), consumer side.

    * rendered chunks under ``proc_root`` -> served from disk exactly like the reference's __getitem__;
    * no corpus (``root`` unset or missing: ``${oc.env:DATASET_ROOT}`` without the variable) -> the synthetic
      white-noise items of BASELINE.json's configs, ``total_chunks`` of them, with a one-time warning;
    * a corpus that would have to be RENDERED -> NotImplementedError."""

    _SEEDS = {"train": 12345, "val": 22345, "test": 32345}

    def __init__(self, root=None, sample_rate=48000, chunk_size=262144, total_chunks=1000, effect_modules=None,
                 effects_to_keep=None, effects_to_remove=None, num_kept_effects=(1, 5), num_removed_effects=(1, 5),
                 shuffle_kept_effects=True, shuffle_removed_effects=False, render_files=True, render_root=None,
                 mode="train", parallel=False):
        super().__init__()
        self.root, self.sample_rate, self.chunk_size, self.total_chunks = root, sample_rate, chunk_size, total_chunks
        self.mode, self.effects = mode, effect_modules or {}
        self.effects_to_keep = [] if effects_to_keep is None else list(effects_to_keep)
        self.effects_to_remove = [] if effects_to_remove is None else list(effects_to_remove)
        self.num_kept_effects, self.num_removed_effects = list(num_kept_effects), list(num_removed_effects)
        self.shuffle_kept_effects, self.shuffle_removed_effects = shuffle_kept_effects, shuffle_removed_effects
        self.validate_effect_input()
        effects_string = "_".join(self.effects_to_keep + ["_"] + self.effects_to_remove + ["_"]
                                  + [str(x) for x in self.num_kept_effects] + ["_"]
                                  + [str(x) for x in self.num_removed_effects])          # datasets.py:370-379
        self.proc_root = (Path(str(render_root)) / "processed" / effects_string / mode
                          if render_root is not None and "<unset env" not in str(render_root) else None)
        rendered = self._rendered_chunks()
        self.synthetic = None
        if rendered:
            self.total_chunks = rendered                       # datasets.py:451 (render_files=False branch)
        elif _usable_dir(root) and render_files:
            raise NotImplementedError(
                f"EffectDataset(mode={mode!r}): rendering {total_chunks} chunks from {root} needs the reference's "
                "pedalboard / pyloudnorm pipeline (datasets.py:399-452; SURVEY 8(f) rank 3).  Render once with the "
                "reference's scripts/generate_dataset.py, then point render_root at the result with render_files=False.")
        else:
            warnings.warn("EffectDataset: no corpus (DATASET_ROOT) and no rendered chunks: serving seeded white-noise "
                          "clips (BASELINE.json synthetic inputs)", stacklevel=2)
            self.synthetic = SyntheticEffectDataset(total_chunks=total_chunks, chunk_size=chunk_size,
                                                    seed=self._SEEDS.get(mode, 42345))

    def _rendered_chunks(self):
        if self.proc_root is None or not self.proc_root.is_dir():
            return 0
        return sum(1 for p in self.proc_root.iterdir() if (p / "input.wav").exists())

    def validate_effect_input(self):
        """The three checks of datasets.py:470-505: effect objects are Pedalboard_Effects members, every name to keep /
        remove is a key of effect_modules, and the [min, max] counts are ordered."""
        from .effects import Pedalboard_Effects
        for effect in self.effects.values():
            if type(effect) not in Pedalboard_Effects:
                raise ValueError(f"Effect {effect} not found in ALL_EFFECTS. Please choose from {Pedalboard_Effects}")
        for name in self.effects_to_keep + self.effects_to_remove:
            if name not in self.effects:
                raise ValueError(f"Effect {name} not found in self.effects. Please choose from {list(self.effects)}")
        for lo_hi, what in ((self.num_kept_effects, "kept"), (self.num_removed_effects, "removed")):
            if lo_hi[0] > lo_hi[1]:
                raise ValueError(f"num_{what}_effects must be a tuple of (min, max). Got {lo_hi}")

    def __len__(self):
        return self.total_chunks

    def __getitem__(self, idx):
        if self.synthetic is not None:
            return self.synthetic[idx]
        d = self.proc_root / str(idx)
        dry_effect_names = torch.load(d / "dry_effects.pt")
        wet_effect_names = torch.load(d / "wet_effects.pt")
        inp, _ = load_wav(d / "input.wav")
        tgt, _ = load_wav(d / "target.wav")
        return inp, tgt, dry_effect_names, wet_effect_names


class InferenceDataset(Dataset):
    """remfx.datasets.InferenceDataset (datasets.py:587-620): paired clean/ and effected/ WAV folders.
    Resample -> sum to mono -> pad / trim `effected` to `clean`; labels dry = 0, wet = 1.  The resampler is the
    device-side polyphase kernel (remfx_amd.resample) when ``device`` is a GPU, so set ``device`` and use
    ``num_workers=0``; files already at ``sample_rate`` need no resampling and stay on the host."""

    def __init__(self, root: str, sample_rate: int, device=None, **kwargs):
        self.root, self.sample_rate, self.device = Path(root), sample_rate, device
        self.clean_paths = sorted(self.root.glob("clean/*.wav"))
        self.effected_paths = sorted(self.root.glob("effected/*.wav"))

    def __len__(self):
        return len(self.clean_paths)

    def _load(self, path):
        audio, sr = load_wav(path)
        if self.device is not None:
            audio = audio.to(self.device)
        if sr != self.sample_rate:
            from .resample import resample
            audio = resample(audio, sr, self.sample_rate)
        return audio

    def __getitem__(self, idx):
        clean = self._load(self.clean_paths[idx]).sum(0, keepdim=True)
        effected = self._load(self.effected_paths[idx]).sum(0, keepdim=True)
        if effected.shape[1] > clean.shape[1]:
            effected = effected[:, :clean.shape[1]]
        elif effected.shape[1] < clean.shape[1]:
            effected = torch.nn.functional.pad(effected, (0, clean.shape[1] - effected.shape[1]))
        return effected, clean, torch.zeros(NUM_EFFECTS, device=clean.device), torch.ones(NUM_EFFECTS, device=clean.device)


class EffectDatamodule:
    """remfx.datasets.EffectDatamodule (datasets.py:623-673): same constructor, same three loaders (val uses
    train_batch_size); unknown kwargs are swallowed like upstream (datasets.py:634).
    Datasets may be objects (the reference's usage) or kwargs dicts for SyntheticEffectDataset.

    With torch.distributed initialised every loader shards its dataset over the ranks with a DistributedSampler
    (Lightning does this for the reference when devices > 1); call ``set_epoch`` once per epoch."""

    def __init__(self, train_dataset=None, val_dataset=None, test_dataset=None, *, train_batch_size=16,
                 test_batch_size=1, num_workers=0, pin_memory=False, **kwargs):
        mk = lambda d, seed: d if isinstance(d, Dataset) else SyntheticEffectDataset(seed=seed, **(d or {}))
        self.train_dataset, self.val_dataset, self.test_dataset = (mk(train_dataset, 12345), mk(val_dataset, 22345),
                                                                   mk(test_dataset, 32345))
        self.train_batch_size, self.test_batch_size = train_batch_size, test_batch_size
        self.num_workers, self.pin_memory = num_workers, pin_memory
        self.epoch = 0

    def setup(self, stage=None):
        pass

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def _dl(self, ds, bs, shuffle):
        import torch.distributed as dist
        sampler = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(ds, shuffle=shuffle)
            sampler.set_epoch(self.epoch)
            shuffle = False
        on_device = getattr(ds, "device", None) is not None        # device-side data prep: main process only
        return DataLoader(ds, batch_size=bs, shuffle=shuffle, sampler=sampler,
                          num_workers=0 if on_device else self.num_workers,
                          pin_memory=self.pin_memory and not on_device)

    def train_dataloader(self):
        return self._dl(self.train_dataset, self.train_batch_size, True)

    def val_dataloader(self):
        return self._dl(self.val_dataset, self.train_batch_size, False)

    def test_dataloader(self):
        return self._dl(self.test_dataset, self.test_batch_size, False)


SyntheticEffectDatamodule = EffectDatamodule      # round-1 name

"""Data module of the removal path: the reference's loader surface (remfx/datasets.py:333-673) over
  * rendered clips on disk in the reference's layout (datasets.py:370-380, 445-468):
        {render_root}/processed/{effects_string}/{mode}/{idx}/{input.wav, target.wav, dry_effects.pt, wet_effects.pt}
  * or seeded white noise when no corpus / rendered set is reachable (BASELINE.json's configs).

What the hot path consumes is the batch tuple  (x_wet, y_dry, dry_labels, wet_labels)  with shapes
(B,1,T), (B,1,T), (B,5), (B,5) (datasets.py:461-468).  RENDERING clips (pedalboard / pyloudnorm on the CPU in the
reference, datasets.py:109-202, 267-318, 399-452) runs ON THE DEVICE here: `process_effects` applies the randomly chosen
kept / removed effects and the in-between loudness normalisation through remfx_amd.effects (csrc/fx.hip), for
`EffectDataset(render_files=True)` (writes the reference's layout) and `DynamicEffectDataset` (on-the-fly augmentation).

The classes take the reference's constructor arguments, so ``cfg/config.yaml``'s datamodule node instantiates
unchanged.  Multi-GPU: the loaders shard by rank with a DistributedSampler (what Lightning injects for the
reference), see ``EffectDatamodule._dl``.
"""
import os
import sys
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


# split tables of the four corpora (datasets.py:20-50)
vocalset_splits = {"train": ["male1", "male2", "male3", "male4", "male5", "male6", "male7", "male8", "male9", "female1",
                             "female2", "female3", "female4", "female5", "female6", "female7"],
                   "val": ["male10", "female8"], "test": ["male11", "female9"]}
guitarset_splits = {"train": ["00", "01", "02", "03"], "val": ["04"], "test": ["05"]}
dsd_100_splits = {"train": ["train"], "val": ["val"], "test": ["test"]}
idmt_drums_splits = {"train": ["WaveDrum02", "TechnoDrum01"], "val": ["RealDrum01"], "test": ["TechnoDrum02", "WaveDrum01"]}


def locate_files(root, mode):
    """datasets.py:53-106: one sorted file list per corpus found under `root` (VocalSet1-2, audio_mono-mic = GuitarSet,
    DSD100/DSD100, IDMT-SMT-DRUMS-V2), restricted to the split of `mode`."""
    import glob
    root, file_list = str(root), []
    d = os.path.join(root, "VocalSet1-2")
    if os.path.isdir(d):
        files = []
        for sd in glob.glob(os.path.join(d, "data_by_singer", "*")):
            if os.path.basename(sd) in vocalset_splits[mode]:
                files += glob.glob(os.path.join(sd, "**", "**", "*.wav"))
        print(f"Found {len(files)} files in VocalSet {mode}.")
        file_list.append(sorted(files))
    d = os.path.join(root, "audio_mono-mic")
    if os.path.isdir(d):
        files = [f for f in glob.glob(os.path.join(d, "*.wav")) if os.path.basename(f).split("_")[0] in guitarset_splits[mode]]
        print(f"Found {len(files)} files in GuitarSet {mode}.")
        file_list.append(sorted(files))
    d = os.path.join(root, "DSD100/DSD100")
    if os.path.isdir(d):
        files = glob.glob(os.path.join(d, mode, "**", "*.wav"), recursive=True)
        file_list.append(sorted(files))
        print(f"Found {len(files)} files in DSD100 {mode}.")
    d = os.path.join(root, "IDMT-SMT-DRUMS-V2")
    if os.path.isdir(d):
        files = [f for f in glob.glob(os.path.join(d, "audio", "*.wav")) if os.path.basename(f).split("_")[0] in idmt_drums_splits[mode]]
        file_list.append(sorted(files))
        print(f"Found {len(files)} files in IDMT-SMT-Drums {mode}.")
    return file_list


def select_random_chunk(audio_file, chunk_size, sample_rate, device=None):
    """utils.py:120-135: a random chunk of `chunk_size` samples AT `sample_rate` from a file (None when the file is too
    short or the chunk is nearly silent); the resampling runs on the device (remfx_amd.resample)."""
    audio, sr = load_wav(audio_file)
    new_chunk_size = int(chunk_size * (sr / sample_rate))
    if new_chunk_size >= audio.shape[-1]:
        return None
    max_len = audio.shape[-1] - new_chunk_size
    random_start = torch.randint(0, max_len, (1,)).item()
    chunk = audio[:, random_start:random_start + new_chunk_size]
    if torch.mean(torch.abs(chunk)) < 1e-4:                     # skip if energy too low
        return None
    if device is not None:
        chunk = chunk.to(device)
    if sr != sample_rate:
        from .resample import resample
        chunk = resample(chunk, sr, sample_rate)
    return chunk


def process_effects(dry, effects, effects_to_keep, effects_to_remove, num_kept_effects, num_removed_effects,
                    shuffle_kept_effects, shuffle_removed_effects, normalize):
    """datasets.py:267-318 (= 139-191 of parallel_process_effects): random subset of the effects to keep applied to the dry
    clip, random subset of the effects to remove applied on top for the wet clip, loudness normalisation after every effect
    and at the end; the same random calls in the same order as the reference.  dry: (1, T) on the device.
    Returns (normalized_dry, normalized_wet, dry_labels (5,), wet_labels (5,))."""
    from .effects import Pedalboard_Effects as ALL_EFFECTS
    idx = torch.randperm(len(effects_to_keep)) if shuffle_kept_effects else torch.arange(len(effects_to_keep))
    r1, r2 = num_kept_effects[0], num_kept_effects[1]
    n = torch.round((r1 - r2) * torch.rand(1) + r2).int()
    dry_labels = []
    for effect in [effects[effects_to_keep[i]] for i in idx[:n]]:
        dry = normalize(effect(dry))                              # normalise in-between effects
        dry_labels.append(ALL_EFFECTS.index(type(effect)))
    idx = torch.randperm(len(effects_to_remove)) if shuffle_removed_effects else torch.arange(len(effects_to_remove))
    wet = torch.clone(dry)
    r1, r2 = num_removed_effects[0], num_removed_effects[1]
    n = torch.round((r1 - r2) * torch.rand(1) + r2).int()
    wet_labels = []
    for effect in [effects[effects_to_remove[i]] for i in idx[:n]]:
        wet = normalize(effect(wet))
        wet_labels.append(ALL_EFFECTS.index(type(effect)))
    wet_labels_tensor, dry_labels_tensor = torch.zeros(len(ALL_EFFECTS)), torch.zeros(len(ALL_EFFECTS))
    for i in wet_labels:
        wet_labels_tensor[i] = 1.0
    for i in dry_labels:
        dry_labels_tensor[i] = 1.0
    return normalize(dry), normalize(wet), dry_labels_tensor, wet_labels_tensor


def _random_chunk(files, chunk_size, sample_rate, device):
    import random
    chunk = None
    corpus = random.choice(files)
    while chunk is None:
        chunk = select_random_chunk(random.choice(corpus), chunk_size, sample_rate, device)
    if chunk.shape[0] > 1:                                       # sum to mono
        chunk = chunk.sum(0, keepdim=True)
    return chunk


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
    * a corpus + ``render_files=True`` -> the chunks are RENDERED on the device (datasets.py:399-452: random chunk of a
      random file, mono, `process_effects`, written as input.wav / target.wav / dry_effects.pt / wet_effects.pt) by RANK 0
      only (the other ranks wait at a barrier); an existing rendered set is replaced only after upstream's y/n question on a
      terminal or with REMFX_OVERWRITE_RENDERED=1, and kept (with a warning) otherwise."""

    _SEEDS = {"train": 12345, "val": 22345, "test": 32345}

    def __init__(self, root=None, sample_rate=48000, chunk_size=262144, total_chunks=1000, effect_modules=None,
                 effects_to_keep=None, effects_to_remove=None, num_kept_effects=(1, 5), num_removed_effects=(1, 5),
                 shuffle_kept_effects=True, shuffle_removed_effects=False, render_files=True, render_root=None,
                 mode="train", parallel=False, device=None):
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
        self.device = device
        if _usable_dir(root) and render_files and self.proc_root is not None:
            self._render(rendered)
        elif rendered:
            if render_files and rendered != total_chunks:
                warnings.warn(f"EffectDataset(mode={mode!r}): render_files=True but no corpus to render from; serving the "
                              f"{rendered} chunks already under {self.proc_root} (asked for {total_chunks})", stacklevel=2)
            self.total_chunks = rendered                       # datasets.py:451 (render_files=False branch)
        else:
            warnings.warn("EffectDataset: no corpus (DATASET_ROOT) and no rendered chunks: serving seeded white-noise "
                          "clips (BASELINE.json synthetic inputs)", stacklevel=2)
            self.synthetic = SyntheticEffectDataset(total_chunks=total_chunks, chunk_size=chunk_size,
                                                    seed=self._SEEDS.get(mode, 42345))

    def _render(self, rendered):
        """datasets.py:381-452 on the device.  Rank 0 renders, the other ranks wait at a barrier and then list the same
        `proc_root` (every rank constructs the dataset under DDP; concurrent renders into one directory would tear the
        input / target / label files of a chunk apart).  An existing non-empty rendered set is never deleted implicitly:
        upstream asks y/n on stdin (datasets.py:385-395) -- so does this on a terminal; without one it takes
        REMFX_OVERWRITE_RENDERED=1 as the "y" and otherwise KEEPS the set with a warning."""
        import shutil
        from .effects import LoudnessNormalize
        if not torch.cuda.is_available():
            raise RuntimeError("EffectDataset(render_files=True) renders on the GPU (remfx_amd.effects has no CPU path)")
        # scripts/train.py instantiates the datamodule BEFORE the Trainer brings the process group up: under a launcher (WORLD_SIZE > 1
        # in the environment) join the group here, or every rank would think it is rank 0 of 1 and render into the same directory
        if int(os.environ.get("WORLD_SIZE", "1")) > 1 and torch.distributed.is_available() and not torch.distributed.is_initialized():
            from . import ddp
            ddp.init_from_env()
        dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()
        rank = torch.distributed.get_rank() if dist_on else 0
        try:
            if rank == 0:
                self._render_rank0(rendered, shutil, LoudnessNormalize)
        finally:
            if dist_on:
                torch.distributed.barrier()                  # the other ranks list proc_root only after rank 0 is done
        self.total_chunks = self._rendered_chunks() or self.total_chunks

    def _render_rank0(self, rendered, shutil, LoudnessNormalize):
        dev = torch.device(self.device) if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        files = locate_files(self.root, self.mode)
        if not files or not any(files):
            raise ValueError(f"EffectDataset: no audio files of the known corpora under {self.root} for mode {self.mode!r}")
        if rendered:
            if os.environ.get("REMFX_OVERWRITE_RENDERED", "") == "1":
                answer = "y"
            elif sys.stdin is not None and sys.stdin.isatty():
                answer = input("WARNING: By default, will re-render files.\nSet render_files=False to skip re-rendering.\n"
                               "Are you sure you want to re-render? (y/n): ")
            else:
                answer = "n"
            if answer != "y":
                warnings.warn(f"EffectDataset: keeping the {rendered} chunks already under {self.proc_root} "
                              "(render_files=True, but an existing set is only replaced after a 'y' on the terminal or with "
                              "REMFX_OVERWRITE_RENDERED=1)", stacklevel=3)
                return
            shutil.rmtree(self.proc_root)
        self.proc_root.mkdir(parents=True, exist_ok=True)
        normalize = LoudnessNormalize(self.sample_rate, target_lufs_db=-20)
        for num_chunk in range(self.total_chunks):
            chunk = _random_chunk(files, self.chunk_size, self.sample_rate, dev)
            dry, wet, dry_effects, wet_effects = process_effects(
                chunk, self.effects, self.effects_to_keep, self.effects_to_remove, self.num_kept_effects,
                self.num_removed_effects, self.shuffle_kept_effects, self.shuffle_removed_effects, normalize)
            d = self.proc_root / str(num_chunk)
            d.mkdir(exist_ok=True)
            save_wav(d / "input.wav", wet, self.sample_rate)
            save_wav(d / "target.wav", dry, self.sample_rate)
            torch.save(dry_effects, d / "dry_effects.pt")
            torch.save(wet_effects, d / "wet_effects.pt")

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


class DynamicEffectDataset(Dataset):
    """remfx.datasets.DynamicEffectDataset (datasets.py:205-330; cfg/exp/5-5_full_cls_dynamic.yaml): every item is a fresh
    random chunk with freshly drawn effects -- rendered on the device (`device`, default the current GPU), so use
    ``num_workers=0``.  Without a corpus (``root`` unset / missing) the source chunks are seeded white noise at about
    -20 dB and the effects are still drawn and rendered: on-the-fly augmentation stays exercisable offline."""

    def __init__(self, root=None, sample_rate=48000, chunk_size=262144, total_chunks=1000, effect_modules=None,
                 effects_to_keep=None, effects_to_remove=None, num_kept_effects=(1, 5), num_removed_effects=(1, 5),
                 shuffle_kept_effects=True, shuffle_removed_effects=False, render_files=True, render_root=None,
                 mode="train", parallel=False, device=None):
        super().__init__()
        from .effects import LoudnessNormalize
        self.root, self.sample_rate, self.chunk_size, self.total_chunks = root, sample_rate, chunk_size, total_chunks
        self.mode, self.effects = mode, effect_modules or {}
        self.effects_to_keep = [] if effects_to_keep is None else list(effects_to_keep)
        self.effects_to_remove = [] if effects_to_remove is None else list(effects_to_remove)
        self.num_kept_effects, self.num_removed_effects = list(num_kept_effects), list(num_removed_effects)
        self.shuffle_kept_effects, self.shuffle_removed_effects = shuffle_kept_effects, shuffle_removed_effects
        self.normalize = LoudnessNormalize(sample_rate, target_lufs_db=-20)
        self.device, self.renders_on_device = device, True
        self.files = locate_files(root, mode) if _usable_dir(root) else []
        if not any(self.files):
            warnings.warn("DynamicEffectDataset: no corpus (DATASET_ROOT): effects are rendered over seeded white-noise chunks",
                          stacklevel=2)
            self.files = []
        self._noise = torch.Generator().manual_seed(EffectDataset._SEEDS.get(mode, 42345))

    def process_effects(self, dry):
        return process_effects(dry, self.effects, self.effects_to_keep, self.effects_to_remove, self.num_kept_effects,
                               self.num_removed_effects, self.shuffle_kept_effects, self.shuffle_removed_effects, self.normalize)

    def __len__(self):
        return self.total_chunks

    def __getitem__(self, _):
        if not torch.cuda.is_available():
            raise RuntimeError("DynamicEffectDataset renders on the GPU (remfx_amd.effects has no CPU path)")
        dev = torch.device(self.device) if self.device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.files:
            chunk = _random_chunk(self.files, self.chunk_size, self.sample_rate, dev)
        else:
            chunk = (torch.randn(1, self.chunk_size, generator=self._noise) * 0.1).to(dev)
        dry, wet, dry_effects, wet_effects = self.process_effects(chunk)
        return wet, dry, dry_effects, wet_effects


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
        # eval.sh passes `datamodule.train_dataset=None` (the string): a split that is not used stays empty
        mk = lambda d, seed: (d if isinstance(d, Dataset) else None if isinstance(d, str) else
                              SyntheticEffectDataset(seed=seed, **(d or {})))
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
        # device-side data prep (resampling / effect rendering): main process only
        on_device = getattr(ds, "device", None) is not None or getattr(ds, "renders_on_device", False)
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

"""Synthetic stand-in for the reference's EffectDatamodule (remfx/datasets.py:624-673).

The reference renders effected audio with pedalboard / sox from corpora on disk: that
CPU/offline data path is outside the hot path.  What the hot path consumes
is only the batch tuple  (x_wet, y_dry, dry_labels, wet_labels)  with shapes
(B,1,T), (B,1,T), (B,5), (B,5) (datasets.py:461-468) -- produced here from seeded white
noise at the level the dataset normalises to (about -20 dB), as BASELINE.json's configs ask.
"""
import torch


class SyntheticEffectDataset(torch.utils.data.Dataset):
    def __init__(self, total_chunks=8, chunk_size=262144, seed=12345, num_classes=5, level=0.1, **_):
        self.n, self.t, self.seed, self.k, self.level = total_chunks, chunk_size, seed, num_classes, level

    def __len__(self):
        return self.n

    def __getitem__(self, idx):
        g = torch.Generator().manual_seed(self.seed + idx)
        x = torch.randn(1, self.t, generator=g) * self.level
        y = torch.randn(1, self.t, generator=g) * self.level
        wet = (torch.rand(5, generator=g) > 0.5).float()
        return x, y, torch.zeros(5), wet


class SyntheticEffectDatamodule:
    """Same loader surface as EffectDatamodule: {train,val,test}_dataloader(); unknown kwargs are
    swallowed like upstream (datasets.py:634); val uses train_batch_size (Q15)."""

    def __init__(self, train_dataset=None, val_dataset=None, test_dataset=None, *, train_batch_size=16,
                 test_batch_size=1, num_workers=0, pin_memory=False, **kwargs):
        mk = lambda d, seed: d if isinstance(d, torch.utils.data.Dataset) else SyntheticEffectDataset(seed=seed, **(d or {}))
        self.train_dataset, self.val_dataset, self.test_dataset = mk(train_dataset, 12345), mk(val_dataset, 22345), mk(test_dataset, 32345)
        self.train_batch_size, self.test_batch_size = train_batch_size, test_batch_size
        self.num_workers, self.pin_memory = num_workers, pin_memory

    def _dl(self, ds, bs, shuffle):
        return torch.utils.data.DataLoader(ds, batch_size=bs, shuffle=shuffle, num_workers=self.num_workers,
                                           pin_memory=self.pin_memory)

    def train_dataloader(self):
        return self._dl(self.train_dataset, self.train_batch_size, True)

    def val_dataloader(self):
        return self._dl(self.val_dataset, self.train_batch_size, False)

    def test_dataloader(self):
        return self._dl(self.test_dataset, self.test_batch_size, False)

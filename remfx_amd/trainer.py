"""Minimal stand-in for pytorch_lightning.Trainer, used when lightning is not installed
(it is not in this image).  Implements exactly what scripts/train.py / chain_inference.py use
(reference scripts/train.py:40-55, cfg/config.yaml:110-120): fit() over train batches with
gradient-norm clipping, optimiser + per-step LR schedule, per-epoch validation, a
{"state_dict": ...} checkpoint (the layout chain_inference.py:24-25 loads), and test().
Multi-GPU: one process per GPU (torch.distributed / RCCL); gradients of the flat buffer are
all-reduced by remfx_amd.ddp.GradSync, logged scalars are mean-reduced (sync_dist=True).
"""
import os

import torch

from . import ddp


class CSVLogger:
    def __init__(self, save_dir="./logs", name="", version=None, **_):
        self.path = os.path.join(save_dir, name or "", "metrics.csv")
        self.rows = []

    def log(self, step, metrics):
        self.rows.append(dict(step=step, **{k: float(v) for k, v in metrics.items()}))

    def save(self):
        if not self.rows:
            return
        os.makedirs(os.path.dirname(self.path), exist_ok=True)
        keys = sorted({k for r in self.rows for k in r})
        with open(self.path, "w") as f:
            f.write(",".join(keys) + "\n")
            for r in self.rows:
                f.write(",".join(str(r.get(k, "")) for k in keys) + "\n")


class Trainer:
    def __init__(self, max_steps=50000, max_epochs=-1, min_epochs=0, gradient_clip_val=None, accelerator=None,
                 devices=1, precision=32, log_every_n_steps=1, accumulate_grad_batches=1, callbacks=None,
                 logger=None, limit_val_batches=None, limit_test_batches=None, **kwargs):
        self.max_steps, self.max_epochs = max_steps, max_epochs
        self.gradient_clip_val = gradient_clip_val
        self.accelerator, self.devices, self.precision = accelerator, devices, precision
        self.callbacks, self.logger = callbacks or [], logger
        self.limit_val_batches, self.limit_test_batches = limit_val_batches, limit_test_batches
        self.global_step = 0
        self.rank, self.local_rank, self.world = ddp.init_from_env()
        if str(precision) not in ("32", "32-true"):
            raise NotImplementedError("precision: the HIP path computes in fp32 this round (DESIGN.md)")
        self.device = torch.device("cuda", self.local_rank) if accelerator in ("gpu", "cuda") or (
            accelerator is None and torch.cuda.is_available()) else torch.device("cpu")

    def _to(self, batch):
        return tuple(t.to(self.device, non_blocking=True) if torch.is_tensor(t) else t for t in batch)

    def _log(self, model, prefix_filter=None):
        vals = {k: ddp.all_reduce_mean_scalar(torch.as_tensor(v, dtype=torch.float32, device=self.device))
                for k, v in getattr(model, "logged", {}).items()}
        if self.logger is not None and self.rank == 0:
            self.logger.log(self.global_step, vals)
        return vals

    def fit(self, model, datamodule=None, ckpt_dir=None):
        model.trainer = self
        model.to(self.device)
        cfg = model.configure_optimizers()
        opt = cfg["optimizer"] if isinstance(cfg, dict) else cfg
        sched = cfg["lr_scheduler"]["scheduler"] if isinstance(cfg, dict) and "lr_scheduler" in cfg else None
        ddp.broadcast_parameters(opt.flat.data)
        sync = ddp.GradSync(opt.flat)
        epoch = 0
        last = {}
        while self.global_step < self.max_steps and (self.max_epochs < 0 or epoch < self.max_epochs):
            model.train()
            for i, batch in enumerate(datamodule.train_dataloader()):
                if self.global_step >= self.max_steps:
                    break
                opt.zero_grad()
                loss = model.training_step(self._to(batch), i)
                loss.backward()
                pre = sync.finish()
                opt.step(clip_norm=self.gradient_clip_val, grad_prescale=pre)
                if sched is not None:
                    sched.step()
                self.global_step += 1
                last = self._log(model)
            epoch += 1
            from . import lstm as _lstm
            if _lstm.error_flag():              # checked once per epoch (device sync): never train on a timed-out exchange
                raise RuntimeError("LSTM recurrence kernel reported a spin time-out")
            if hasattr(datamodule, "val_dataloader"):
                model.eval()
                with torch.no_grad():
                    for i, batch in enumerate(datamodule.val_dataloader()):
                        if self.limit_val_batches is not None and i >= self.limit_val_batches:
                            break
                        model.validation_step(self._to(batch), i)
                last.update(self._log(model))
        if ckpt_dir and self.rank == 0:
            os.makedirs(ckpt_dir, exist_ok=True)
            torch.save({"state_dict": model.state_dict(), "global_step": self.global_step, "epoch": epoch},
                       os.path.join(ckpt_dir, "last.ckpt"))
        if self.logger is not None and self.rank == 0:
            self.logger.save()
        self.logged_metrics = last
        return last

    def test(self, model, datamodule=None, ckpt_path=None):
        model.trainer = self
        model.to(self.device)
        if ckpt_path and os.path.exists(str(ckpt_path)):
            model.load_state_dict(torch.load(ckpt_path, map_location=self.device)["state_dict"])
        model.eval()           # registered sub-modules only: removal models kept in a plain dict stay as they are (Q6)
        sums, n = {}, 0
        with torch.no_grad():
            for i, batch in enumerate(datamodule.test_dataloader()):
                if self.limit_test_batches is not None and i >= self.limit_test_batches:
                    break
                model.test_step(self._to(batch), i)
                for k, v in model.logged.items():
                    sums[k] = sums.get(k, 0.0) + float(v)
                n += 1
        out = {k: float(ddp.all_reduce_mean_scalar(torch.tensor(v / max(n, 1), device=self.device))) for k, v in sums.items()}
        if self.logger is not None and self.rank == 0:
            self.logger.log(self.global_step, out)
            self.logger.save()
        return [out]

"""Minimal stand-in for pytorch_lightning.Trainer, used when lightning is not installed
(it is not in this image).  Implements exactly what scripts/train.py / chain_inference.py use
(reference scripts/train.py:40-55, cfg/config.yaml:110-120): fit() over train batches with
gradient-norm clipping, optimiser + per-step LR schedule, per-epoch validation, checkpoints in
Lightning's on-disk layout (state_dict, optimizer_states, lr_schedulers, epoch, global_step: what
chain_inference.py:24-25 / test.py:20-23 read), resume, and test().
Multi-GPU: one process per GPU (torch.distributed / RCCL); gradients of the flat buffer are
all-reduced by remfx_amd.ddp.GradSync, logged scalars are mean-reduced (sync_dist=True).
"""
import os

import torch

from . import ddp


class CSVLogger:
    """pytorch_lightning.loggers.CSVLogger as cfg/logger/csv.yaml configures it (`save_dir: "."`, `version: ${now:...}`):
    rows go to `<save_dir>/<name>/<version>/metrics.csv`, `name` defaulting to Lightning's "lightning_logs", an integer or missing
    version spelled `version_<n>` (the next free n when missing); `name=""` drops that directory level, as in Lightning."""

    def __init__(self, save_dir="./logs", name="lightning_logs", version=None, **_):
        base = os.path.join(str(save_dir), name or "")
        if version is None:
            taken = [int(d[8:]) for d in (os.listdir(base) if os.path.isdir(base) else []) if d.startswith("version_") and d[8:].isdigit()]
            version = max(taken, default=-1) + 1
        self.log_dir = os.path.join(base, version if isinstance(version, str) else f"version_{version}")
        self.path = os.path.join(self.log_dir, "metrics.csv")
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


PRECISIONS = {"32": "f32", "32-true": "f32", "bf16-mixed": "bf16", "bf16": "bf16"}


def _check_lstm(where):
    """Raise on EVERY rank when any rank's recurrence kernel timed out (a rank raising alone would leave the others hung in
    the next collective): the flag is max-reduced over the process group first."""
    from . import lstm as _lstm
    bad = _lstm.error_flag()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor([1.0 if bad else 0.0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        bad = bool(t.item() > 0)
    if bad:
        raise RuntimeError(f"LSTM recurrence kernel reported a spin time-out ({where}): the step's outputs are invalid")


def load_checkpoint_file(path, map_location="cpu"):
    """torch.load of a Lightning / remfx_amd .ckpt; a missing file is an error, as in the reference (test.py:20).
    RFX_ALLOW_RANDOM_INIT=1 turns it into a warning + None (throughput runs without released checkpoints)."""
    if not path or not os.path.exists(str(path)):
        if os.environ.get("RFX_ALLOW_RANDOM_INIT") == "1":
            import warnings
            warnings.warn(f"checkpoint {path!r} not found: keeping the seeded random initialisation (RFX_ALLOW_RANDOM_INIT=1)")
            return None
        raise FileNotFoundError(f"checkpoint {path!r} not found (set RFX_ALLOW_RANDOM_INIT=1 to run on random weights)")
    return torch.load(path, map_location=map_location, weights_only=False)


class Trainer:
    def __init__(self, max_steps=50000, max_epochs=-1, min_epochs=0, gradient_clip_val=None, accelerator=None,
                 devices=1, precision=32, log_every_n_steps=1, accumulate_grad_batches=1, callbacks=None,
                 logger=None, limit_val_batches=None, limit_test_batches=None, ckpt_every_n_steps=None, **kwargs):
        self.max_steps, self.max_epochs = max_steps, max_epochs
        self.gradient_clip_val = gradient_clip_val
        self.accelerator, self.devices, self.precision = accelerator, devices, precision
        self.callbacks, self.logger = callbacks or [], logger
        self.limit_val_batches, self.limit_test_batches = limit_val_batches, limit_test_batches
        self.ckpt_every_n_steps = ckpt_every_n_steps
        self.global_step, self.current_epoch = 0, 0
        self.rank, self.local_rank, self.world = ddp.init_from_env()
        if str(precision) not in PRECISIONS:
            raise ValueError(f"precision={precision!r}: supported {sorted(PRECISIONS)} (32 = exact fp32 MFMA; bf16-mixed = "
                             "bf16 MFMA operands, fp32 accumulation / master weights / norms / FFT / losses)")
        # "32" leaves the library's arithmetic mode alone (RFX_GEMM_PREC / ops.set_gemm_precision: f32 or bf16x3)
        self.gemm_mode = "bf16" if PRECISIONS[str(precision)] == "bf16" else None
        self.device = torch.device("cuda", self.local_rank) if accelerator in ("gpu", "cuda") or (
            accelerator is None and torch.cuda.is_available()) else torch.device("cpu")

    _MODE_BEFORE_BF16 = None        # the library mode a bf16-mixed trainer replaced (restored by the next precision=32 trainer)

    def _apply_precision(self):
        from . import ops
        if self.gemm_mode is not None:
            if ops.gemm_precision() != self.gemm_mode and Trainer._MODE_BEFORE_BF16 is None:
                Trainer._MODE_BEFORE_BF16 = ops.gemm_precision()
            ops.set_gemm_precision(self.gemm_mode)
        elif Trainer._MODE_BEFORE_BF16 is not None:      # precision=32 after a bf16-mixed trainer in the same process
            ops.set_gemm_precision(Trainer._MODE_BEFORE_BF16)
            Trainer._MODE_BEFORE_BF16 = None

    def _to(self, batch):
        return tuple(t.to(self.device, non_blocking=True) if torch.is_tensor(t) else t for t in batch)

    def _log(self, model, prefix_filter=None):
        vals = {k: ddp.all_reduce_mean_scalar(torch.as_tensor(v, dtype=torch.float32, device=self.device))
                for k, v in getattr(model, "logged", {}).items()}
        if self.logger is not None and self.rank == 0:
            self.logger.log(self.global_step, vals)
        return vals

    # ---- checkpoints (Lightning's dict layout) ----------------------------------------------------
    def checkpoint_dict(self, model, opt=None, sched=None):
        # callbacks: Lightning keeps ModelCheckpoint's state (best_model_score, best_model_path) here; restored by _resume so that
        # the first validation after a resume does not overwrite a better earlier best.ckpt
        best = getattr(self, "best_score", None)
        ck = {"epoch": self.current_epoch, "global_step": self.global_step, "pytorch-lightning_version": "2.0.0",
              "state_dict": model.state_dict(), "loops": {},
              "callbacks": {"ModelCheckpoint": {"monitor": "valid_loss", "mode": "min", "best_model_score": best,
                                                "best_model_path": getattr(self, "best_path", None) if best is not None else ""}},
              "optimizer_states": [opt.state_dict()] if opt is not None else [],
              "lr_schedulers": [sched.state_dict()] if sched is not None else []}
        return ck

    def save_checkpoint(self, path, model, opt=None, sched=None):
        if self.rank != 0:
            return
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        tmp = str(path) + ".tmp"
        torch.save(self.checkpoint_dict(model, opt, sched), tmp)
        os.replace(tmp, path)                                   # never leave a torn file behind a crash

    def _resume(self, ckpt_path, model, opt, sched):
        ck = load_checkpoint_file(ckpt_path, map_location=self.device)
        if ck is None:
            return
        model.load_state_dict(ck["state_dict"])
        if ck.get("optimizer_states"):
            opt.load_state_dict(ck["optimizer_states"][0])
        if sched is not None and ck.get("lr_schedulers"):
            sched.load_state_dict(ck["lr_schedulers"][0])
        self.global_step, self.current_epoch = int(ck.get("global_step", 0)), int(ck.get("epoch", 0))
        # Lightning keys the callback state by the callback's state_key ("ModelCheckpoint{'monitor': ...}"): take any key of that family
        cbs = ck.get("callbacks") or {}
        cb = next((v for k, v in cbs.items() if str(k).startswith("ModelCheckpoint") and isinstance(v, dict)), {})
        if cb.get("best_model_score") is not None:
            self._resumed_best = float(cb["best_model_score"])
            self._resumed_best_path = cb.get("best_model_path") or None

    def fit(self, model, datamodule=None, ckpt_dir=None, ckpt_path=None):
        """ckpt_dir: where last.ckpt is written (every ckpt_every_n_steps steps when set, every epoch, and at the end);
        ckpt_path: checkpoint to resume from (weights, AdamW moments, scheduler position, step / epoch counters)."""
        self._apply_precision()
        model.trainer = self
        model.to(self.device)
        from . import ops
        ops.enter_compute_stream(self.device)                    # high-priority compute stream; weight gradients ride a normal one
        cfg = model.configure_optimizers()
        opt = cfg["optimizer"] if isinstance(cfg, dict) else cfg
        sched = cfg["lr_scheduler"]["scheduler"] if isinstance(cfg, dict) and "lr_scheduler" in cfg else None
        if ckpt_path:
            self._resume(ckpt_path, model, opt, sched)
        ddp.broadcast_parameters(opt.flat.data)
        ddp.broadcast_buffers(model)                             # BatchNorm running statistics etc., as DDP does
        sync = ddp.GradSync(opt.flat)
        last = {}
        last_path = os.path.join(ckpt_dir, "last.ckpt") if ckpt_dir else None
        # ModelCheckpoint(monitor="valid_loss", mode="min") of the reference's cfg/config.yaml callbacks: best.ckpt next to last.ckpt
        # best_score: restored from the checkpoint this fit resumed from (Lightning's ModelCheckpoint state); _best_written: whether
        # THIS fit wrote best.ckpt -- test(ckpt_path="best") never picks up a stale file an earlier run left in ckpt_dir
        self.best_path, self._best_state = (os.path.join(ckpt_dir, "best.ckpt") if ckpt_dir else None), None
        self.best_score, self._best_written = getattr(self, "_resumed_best", None), False
        # a resumed fit that never beats the restored score must still find the best weights: the restored best_model_path is valid when
        # the file is there and carries exactly the restored score
        self._restored_best_file = None
        rb = getattr(self, "_resumed_best_path", None)
        if self.best_score is not None and rb and os.path.exists(rb):
            try:
                cbr = (load_checkpoint_file(rb, map_location="cpu").get("callbacks") or {})
                sc = next((v.get("best_model_score") for k, v in cbr.items() if str(k).startswith("ModelCheckpoint") and isinstance(v, dict)), None)
                if sc is not None and float(sc) == float(self.best_score):
                    self._restored_best_file = rb
            except Exception:
                pass
        self._resumed_best = self._resumed_best_path = None
        from . import ops as _ops
        step_gc = _ops.StepGC().__enter__()                      # no cyclic-GC pauses inside the enqueue loop (ops.StepGC)
        try:
            while self.global_step < self.max_steps and (self.max_epochs < 0 or self.current_epoch < self.max_epochs):
                model.train()
                if hasattr(datamodule, "set_epoch"):
                    datamodule.set_epoch(self.current_epoch)         # reshuffles the per-rank shards
                for i, batch in enumerate(datamodule.train_dataloader()):
                    if self.global_step >= self.max_steps:
                        break
                    opt.zero_grad()
                    loss = model.training_step(self._to(batch), i)
                    loss.backward()
                    pre = sync.finish()
                    _check_lstm("before the optimiser step")         # never apply gradients of a timed-out exchange
                    opt.step(clip_norm=self.gradient_clip_val, grad_prescale=pre)
                    if sched is not None:
                        sched.step()
                    self.global_step += 1
                    step_gc.tick()
                    last = self._log(model)
                    if last_path and self.ckpt_every_n_steps and self.global_step % self.ckpt_every_n_steps == 0:
                        self.save_checkpoint(last_path, model, opt, sched)
                self.current_epoch += 1
                if hasattr(datamodule, "val_dataloader"):
                    model.eval()
                    with torch.no_grad():
                        for i, batch in enumerate(datamodule.val_dataloader()):
                            if self.limit_val_batches is not None and i >= self.limit_val_batches:
                                break
                            model.validation_step(self._to(batch), i)
                    _check_lstm("validation")
                    last.update(self._log(model))
                    score = last.get("valid_loss")
                    if score is not None and (self.best_score is None or float(score) < self.best_score):
                        self.best_score = float(score)               # `score` is the all-reduced mean: every rank takes the same branch
                        if self.best_path:
                            self.save_checkpoint(self.best_path, model, opt, sched)
                            self._best_written = True
                        else:                                        # no checkpoint directory: keep the weights in memory
                            self._best_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
                if last_path:
                    self.save_checkpoint(last_path, model, opt, sched)
        finally:
            step_gc.__exit__(None, None, None)          # the collector comes back on whatever ends the loop
        if last_path:
            self.save_checkpoint(last_path, model, opt, sched)
        ddp.barrier()                                            # rank 0's files are complete before any rank reads them (test(ckpt_path="best"))
        if self.logger is not None and self.rank == 0:
            self.logger.save()
        self.logged_metrics = last
        return last

    def test(self, model, datamodule=None, ckpt_path=None):
        self._apply_precision()
        model.trainer = self
        model.to(self.device)
        if ckpt_path == "best":           # the best-by-validation-loss weights of the fit that just ended (Lightning's ckpt_path="best")
            best = getattr(self, "best_path", None)
            ddp.barrier()                                        # only rank 0 writes checkpoints
            if best and getattr(self, "_best_written", False) and os.path.exists(best):
                model.load_state_dict(load_checkpoint_file(best, map_location=self.device)["state_dict"])
            elif getattr(self, "_best_state", None) is not None:
                model.load_state_dict(self._best_state)
            elif getattr(self, "_restored_best_file", None):
                model.load_state_dict(load_checkpoint_file(self._restored_best_file, map_location=self.device)["state_dict"])
            elif getattr(self, "best_score", None) is not None:
                import warnings
                warnings.warn("Trainer.test(ckpt_path='best'): the restored best score has no matching checkpoint file; evaluating the "
                              "last-step weights", stacklevel=2)
            # no validation ran: the last-step weights, as Lightning falls back to
        elif ckpt_path:
            ck = load_checkpoint_file(ckpt_path, map_location=self.device)
            if ck is not None:
                model.load_state_dict(ck["state_dict"])
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
        _check_lstm("test")
        out = {k: float(ddp.all_reduce_mean_scalar(torch.tensor(v / max(n, 1), device=self.device))) for k, v in sums.items()}
        if self.logger is not None and self.rank == 0:
            self.logger.log(self.global_step, out)
            self.logger.save()
        return [out]

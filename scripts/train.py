"""Training entry point, same command line as the reference scripts/train.py:9-55:
    python scripts/train.py +exp=chorus_aug model=demucs datamodule.train_batch_size=64 trainer.max_steps=100
Uses hydra + pytorch_lightning when they are installed, otherwise the built-in composer /
trainer (remfx_amd.config, remfx_amd.trainer)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from remfx_amd import config as rcfg  # noqa: E402


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    cfg_dir = os.environ.get("REMFX_CFG_DIR", os.path.join(ROOT, "cfg"))
    cfg = rcfg.compose(cfg_dir, "config.yaml", argv)
    if cfg.get("seed"):
        torch.manual_seed(cfg["seed"])
    datamodule = rcfg.instantiate(cfg["datamodule"])
    model = rcfg.instantiate(cfg["model"])
    logger = rcfg.instantiate(cfg["logger"]) if "logger" in cfg else None
    trainer = rcfg.instantiate(cfg["trainer"], callbacks=[], logger=logger)
    # `+ckpt_path=...` resumes (weights, AdamW moments, scheduler, counters); the reference's own ckpt_path branch
    # (scripts/train.py:19-30) discards what it loads
    metrics = trainer.fit(model=model, datamodule=datamodule, ckpt_path=cfg.get("ckpt_path"),
                          ckpt_dir=os.path.join(cfg.get("logs_dir", "./logs"), "ckpts"))
    if trainer.rank == 0:
        print({k: round(float(v), 5) for k, v in metrics.items()})
    if cfg.get("test_after_fit", True):                     # scripts/train.py:55: trainer.test(..., ckpt_path="best")
        out = trainer.test(model=model, datamodule=datamodule, ckpt_path="best")
        if trainer.rank == 0:
            print({k: round(float(v), 5) for k, v in out[0].items()})
    return metrics


if __name__ == "__main__":
    main()

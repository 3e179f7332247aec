"""Evaluation entry point, same command line as the reference scripts/test.py:10-47 (`eval.sh:38,45`):
    python scripts/test.py +exp=chorus_aug model=demucs +ckpt_path=ckpts/demucs_chorus_aug.ckpt render_files=False
The model named by the config is instantiated, the checkpoint's {"state_dict": ...} is loaded STRICTLY (a missing file
is an error, as upstream; RFX_ALLOW_RANDOM_INIT=1 keeps the seeded initialisation for plumbing runs without released
checkpoints) and `trainer.test` runs over the datamodule's test split, logging test_loss / test_SISDR / test_STFT /
Input_SISDR / Input_STFT (reference remfx/models.py:213-256)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from remfx_amd import config as rcfg  # noqa: E402
from remfx_amd.trainer import load_checkpoint_file  # noqa: E402


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    cfg = rcfg.compose(os.environ.get("REMFX_CFG_DIR", os.path.join(ROOT, "cfg")), "config.yaml", argv)
    if cfg.get("seed"):
        torch.manual_seed(cfg["seed"])
    datamodule = rcfg.instantiate(cfg["datamodule"])
    model = rcfg.instantiate(cfg["model"])
    logger = rcfg.instantiate(cfg["logger"]) if "logger" in cfg else None
    trainer = rcfg.instantiate(cfg["trainer"], callbacks=[], logger=logger)
    ck = load_checkpoint_file(cfg.get("ckpt_path"), map_location=trainer.device)       # test.py:19-23
    if ck is not None:
        model.load_state_dict(ck["state_dict"])                                        # strict, as upstream
    out = trainer.test(model=model, datamodule=datamodule)
    if trainer.rank == 0:
        print({k: round(float(v), 5) for k, v in out[0].items()})
    return out


if __name__ == "__main__":
    main()

"""Chain-inference entry point, same command line as the reference scripts/chain_inference.py:11-73:
    python scripts/chain_inference.py +exp=remfx_detect
Effect-specific removal models + the Cnn14 detector are instantiated from cfg.ckpts / cfg.classifier;
checkpoints ({"state_dict": ...}, strict load) must exist, as upstream; RFX_ALLOW_RANDOM_INIT=1 keeps the seeded
random initialisation instead, with a warning (no released checkpoint is reachable offline: throughput runs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from remfx_amd import config as rcfg  # noqa: E402
from remfx_amd.models import RemFXChainInference  # noqa: E402
from remfx_amd.trainer import load_checkpoint_file  # noqa: E402


def build(cfg, device):
    models = {}
    for effect, node in cfg["ckpts"].items():
        model = rcfg.instantiate(node["model"])
        ck = load_checkpoint_file(node.get("ckpt_path"), map_location=device)
        if ck is not None:
            model.load_state_dict(ck["state_dict"])                                          # strict, as upstream
        models[effect] = model.to(device)
    classifier = None
    if "classifier" in cfg:
        classifier = rcfg.instantiate(cfg["classifier"])
        ck = load_checkpoint_file(cfg.get("classifier_ckpt"), map_location=device)
        if ck is not None:
            classifier.load_state_dict(ck["state_dict"])
        classifier.to(device)
    return RemFXChainInference(models, sample_rate=cfg["sample_rate"], num_bins=cfg["num_bins"],
                               effect_order=list(cfg["inference_effects_ordering"]), classifier=classifier,
                               shuffle_effect_order=cfg["inference_effects_shuffle"],
                               use_all_effect_models=cfg["inference_use_all_effect_models"])


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    cfg = rcfg.compose(os.environ.get("REMFX_CFG_DIR", os.path.join(ROOT, "cfg")), "config.yaml", argv)
    if cfg.get("seed"):
        torch.manual_seed(cfg["seed"])
    datamodule = rcfg.instantiate(cfg["datamodule"])
    logger = rcfg.instantiate(cfg["logger"]) if "logger" in cfg else None
    trainer = rcfg.instantiate(cfg["trainer"], callbacks=[], logger=logger)
    inference_model = build(cfg, trainer.device)
    out = trainer.test(model=inference_model, datamodule=datamodule)
    if trainer.rank == 0:
        print(out)
    return out


if __name__ == "__main__":
    main()

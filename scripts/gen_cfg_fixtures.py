"""Compose the reference's OWN config tree (/root/reference/cfg, build container only) for the command lines of
BASELINE.md section 5 and a few more, and commit the composed dictionaries as a fixture:

    python scripts/gen_cfg_fixtures.py  ->  tests/golden/cfg_composed.json

The fixture is data (the merged key/value tree each command line resolves to).  tests/test_host_cpu.py instantiates
datamodule / model / trainer / classifier / per-effect networks from it on every box, and -- where the reference tree is
present -- re-composes it and compares, so both the composer and the constructor surface are checked against the
reference's real configs (cfg/exp/*.yaml, cfg/model/*.yaml, cfg/config.yaml)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from remfx_amd import config as rcfg  # noqa: E402

REF_CFG = "/root/reference/cfg"
COMMANDS = {
    "config1_umx": ["+exp=distortion", "model=umx", "accelerator=null", "datamodule.train_batch_size=4"],
    "config2_tcn": ["+exp=reverb", "model=tcn", "datamodule.train_batch_size=32"],
    "config3_demucs_bf16": ["+exp=chorus_aug", "model=demucs", "datamodule.train_batch_size=64",
                            "trainer.precision=bf16-mixed", "trainer.devices=8"],
    "config4_dcunet": ["+exp=5-5_full", "model=dcunet", "datamodule.train_batch_size=32", "trainer.devices=8"],
    "config5_remfx_detect": ["+exp=remfx_detect"],
    "cls_5-5_full_cls": ["+exp=5-5_full_cls"],
    "cls_mixup": ["+exp=5-5_full_cls", "model=cls_panns_48k_mixup"],
    "dptnet": ["+exp=distortion", "model=dptnet"],                            # asteroid DPTNet removal network
    "cls_vggish": ["+exp=5-5_full_cls", "model=cls_vggish"],                  # HEAR-embedding classifier head
    "cls_dynamic": ["+exp=5-5_full_cls_dynamic", "model=cls_panns_48k"],      # DynamicEffectDataset: on-the-fly effect rendering
    "cls_16k": ["+exp=5-5_full_cls", "model=cls_panns_16k"],
    "remfx_all": ["+exp=remfx_all"],
    "chain_inference_aug": ["+exp=chain_inference_aug"],
}


def _plain(o):
    if isinstance(o, dict):
        return {str(k): _plain(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_plain(v) for v in o]
    return o


def digest(cfg):
    """sha256 of the composed tree as sorted JSON with `${now:...}` timestamps masked (tests/test_host_cpu.py recomputes it)."""
    import hashlib
    import re
    text = re.sub(r"\d{4}-\d\d-\d\d-\d\d-\d\d-\d\d", "<now>", json.dumps(_plain(cfg), sort_keys=True))
    return hashlib.sha256(text.encode()).hexdigest()


def main():
    os.environ.pop("DATASET_ROOT", None)
    os.environ.pop("WANDB_PROJECT", None)
    os.environ.pop("WANDB_ENTITY", None)
    out = {}
    for name, argv in COMMANDS.items():
        cfg = rcfg.compose(REF_CFG, "config.yaml", argv)
        out[name] = {"argv": argv, "cfg": _plain(cfg)}
    path = os.path.join(ROOT, "tests", "golden", "cfg_composed.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote", path, os.path.getsize(path), "bytes")
    # every experiment and every model file of the reference tree, as digests of the composed dictionary (timestamps masked):
    # pins the files the 13 command lines above do not touch without committing 400 KB of near-identical trees
    dig = {}
    for f in sorted(os.listdir(os.path.join(REF_CFG, "exp"))):
        argv = ["+exp=" + f[:-5]]
        dig[" ".join(argv)] = digest(rcfg.compose(REF_CFG, "config.yaml", argv))
    for f in sorted(os.listdir(os.path.join(REF_CFG, "model"))):
        argv = ["+exp=5-5_full", "model=" + f[:-5]]
        dig[" ".join(argv)] = digest(rcfg.compose(REF_CFG, "config.yaml", argv))
    dig["+exp=default logger=wandb"] = digest(rcfg.compose(REF_CFG, "config.yaml", ["+exp=default", "logger=wandb"]))
    path = os.path.join(ROOT, "tests", "golden", "cfg_digests.json")
    with open(path, "w") as f:
        json.dump(dig, f, indent=1, sort_keys=True)
    print("wrote", path, len(dig), "digests")


if __name__ == "__main__":
    main()

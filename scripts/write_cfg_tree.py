"""Author of the repo's own Hydra config tree (cfg/): the experiment / model / effects / logger schema of the reference
(`cfg/config.yaml`, `cfg/exp/*.yaml`, `cfg/model/*.yaml`, `cfg/effects/all.yaml`, `cfg/logger/*.yaml` -- SURVEY 2.1 row 13,
"keep verbatim-compatible") restated as tables, so that every command line of BASELINE.md section 5 composes from
`ROOT/cfg` without `REMFX_CFG_DIR`.

    python scripts/write_cfg_tree.py          # (re)writes cfg/

The experiment files of the reference repeat one body with three or four fields changed; here the body is one function and
the experiments are rows.  `tests/test_host_cpu.py::test_own_cfg_tree_matches_reference_composition` composes every command
of `scripts/gen_cfg_fixtures.py` from this tree and compares the result with `tests/golden/cfg_composed.json` (made from the
reference's tree in the build container)."""
import os

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "cfg")

FX = ["distortion", "compressor", "reverb", "chorus", "delay"]          # label / removal order (remfx/effects.py:699-707)
SR, NCLS = "${sample_rate}", "${num_classes}"


class _D(yaml.SafeDumper):
    def ignore_aliases(self, data):
        return True


def _emit(rel, body, header="", package_global=True):
    path = os.path.join(CFG, rel)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    text = yaml.dump(body, Dumper=_D, sort_keys=False, default_flow_style=None, width=110)
    with open(path, "w") as f:
        if package_global:
            f.write("# @package _global_\n")
        if header:
            f.write("".join(f"# {ln}\n" for ln in header.splitlines()))
        f.write(text)


# ---- datamodule ------------------------------------------------------------------------------------------------------------------
def dataset(mode, chunks, cls="EffectDataset", parallel=False):
    """One split of remfx.datasets.{EffectDataset, DynamicEffectDataset} (datasets.py:205-330, 333-470): everything but the size,
    the split name and the class refers back to the top-level keys, so an experiment only edits those."""
    d = {"_target_": f"remfx.datasets.{cls}", "total_chunks": chunks, "sample_rate": SR, "root": "${oc.env:DATASET_ROOT}",
         "chunk_size": "${chunk_size}", "mode": mode, "effect_modules": "${effects}"}
    for k in ("effects_to_keep", "effects_to_remove", "num_kept_effects", "num_removed_effects", "shuffle_kept_effects",
              "shuffle_removed_effects", "render_files", "render_root"):
        d[k] = "${%s}" % k
    if parallel is not None:
        d["parallel"] = parallel
    return d


def checkpoint_callbacks(monitor, mode, verbose, audio):
    cb = {"model_checkpoint": {"_target_": "pytorch_lightning.callbacks.ModelCheckpoint", "monitor": monitor, "save_top_k": 1,
                               "save_last": True, "mode": mode, "verbose": verbose,
                               "dirpath": "${logs_dir}/ckpts/${now:%Y-%m-%d-%H-%M-%S}",
                               "filename": "{epoch:02d}-{%s:.3f}" % monitor},
          "learning_rate_monitor": {"_target_": "pytorch_lightning.callbacks.LearningRateMonitor", "logging_interval": "step"}}
    if audio:
        cb["audio_logging"] = {"_target_": "remfx.callbacks.AudioCallback", "sample_rate": SR, "log_audio": "${log_audio}"}
    return cb


def trainer(max_epochs, max_steps):
    return {"_target_": "pytorch_lightning.Trainer", "precision": 32, "min_epochs": 0, "max_epochs": max_epochs,
            "log_every_n_steps": 1, "accumulate_grad_batches": 1, "accelerator": "${accelerator}", "devices": 1,
            "gradient_clip_val": 10.0, "max_steps": max_steps}


def write_primary():
    body = {"defaults": ["_self_", {"model": None}, {"effects": "all"}, {"logger": "csv"}],
            "seed": 12345, "train": True, "sample_rate": 48000, "chunk_size": 262144, "logs_dir": "./logs",
            "render_files": True, "render_root": "./data", "accelerator": None, "log_audio": True,
            "num_kept_effects": [2, 2], "num_removed_effects": [2, 2], "shuffle_kept_effects": True,
            "shuffle_removed_effects": False, "num_classes": 5,
            "effects_to_keep": ["reverb", "chorus", "delay"], "effects_to_remove": ["compressor", "distortion"],
            "callbacks": checkpoint_callbacks("valid_loss", "min", False, True),
            "datamodule": {"_target_": "remfx.datasets.EffectDatamodule",
                           "train_dataset": dataset("train", 8000), "val_dataset": dataset("val", 1000),
                           "test_dataset": dataset("test", 1000),
                           "train_batch_size": 16, "test_batch_size": 1, "num_workers": 8, "pin_memory": True,
                           "persistent_workers": True},
            "trainer": trainer(-1, 50000)}
    _emit("config.yaml", body, package_global=False,
          header="Primary config (scripts/train.py, test.py, chain_inference.py, remfx_detect.py): same keys, defaults and\n"
                 "interpolations as the reference's cfg/config.yaml, so its command lines and overrides apply unchanged.\n"
                 "chunk_size 262144 = 5.5 s @ 48 kHz.  Without DATASET_ROOT the datasets serve seeded white-noise clips.\n"
                 "Written by scripts/write_cfg_tree.py.")


# ---- models ----------------------------------------------------------------------------------------------------------------------
def remfx(network):
    return {"_target_": "remfx.models.RemFX", "lr": 1e-4, "lr_beta1": 0.95, "lr_beta2": 0.999, "lr_eps": 1e-6,
            "lr_weight_decay": 1e-3, "sample_rate": SR, "network": network}


DCUNET_NET = {"_target_": "remfx.models.DCUNetModel", "architecture": "Large-DCUNet-20", "stft_kernel_size": 512,
              "fix_length_mode": "pad", "sample_rate": SR, "num_bins": 1025}
REMOVAL = {
    "demucs": {"_target_": "remfx.models.DemucsModel", "sources": ["mixture"], "audio_channels": 1, "nfft": 4096,
               "sample_rate": SR, "channels": 48},
    "dcunet": DCUNET_NET,
    "umx": {"_target_": "remfx.models.OpenUnmixModel", "n_fft": 2048, "hop_length": 512, "n_channels": 1, "alpha": 0.3,
            "sample_rate": SR},
    "tcn": {"_target_": "remfx.models.TCNModel", "ninputs": 1, "noutputs": 1, "nblocks": 20, "channel_growth": 0,
            "channel_width": 256, "kernel_size": 7, "stack_size": 10, "dilation_growth": 2, "condition": False, "latent_dim": 2,
            "norm_type": "identity", "causal": False, "estimate_loudness": False, "sample_rate": SR, "num_bins": 1025},
    "dptnet": {"_target_": "remfx.models.DPTNetModel", "n_src": 1, "in_chan": 64, "out_chan": 64, "chunk_size": 100,
               "n_repeats": 2, "fb_name": "free", "kernel_size": 16, "n_filters": 64, "stride": 8, "sample_rate": SR,
               "num_bins": 1025},
}


def cnn14(model_sr=SR, n_mels=128, specaugment=None):
    n = {"_target_": "remfx.classifier.Cnn14", "num_classes": NCLS, "n_fft": 2048, "hop_length": 512, "n_mels": n_mels,
         "sample_rate": SR, "model_sample_rate": model_sr}
    if specaugment is not None:
        n["specaugment"] = specaugment
    return n


def classifier(network, mixup=None, label_smoothing=None):
    m = {"_target_": "remfx.models.FXClassifier", "lr": 3e-4, "lr_weight_decay": 1e-3, "sample_rate": SR}
    if mixup is not None:
        m["mixup"] = mixup
    if label_smoothing is not None:
        m["label_smoothing"] = label_smoothing
    m["network"] = network
    return m


def hear(cls):                                   # pretrained-embedding heads (remfx/classifier.py:16-128)
    return {"_target_": f"remfx.classifier.{cls}", "num_classes": NCLS, "sample_rate": SR}


CLASSIFIERS = {   # file -> FXClassifier node
    "cls_panns_16k": classifier(cnn14(model_sr=16000)),
    "cls_panns_44k_label_smoothing": classifier(cnn14(specaugment=False), mixup=True, label_smoothing=0.1),
    "cls_panns_48k": classifier(cnn14(specaugment=False), mixup=False),
    "cls_panns_48k_64": classifier(cnn14(n_mels=64, specaugment=False), mixup=False),
    "cls_panns_48k_mixup": classifier(cnn14(specaugment=False), mixup=True),
    "cls_panns_48k_specaugment": classifier(cnn14(specaugment=True), mixup=False),
    "cls_panns_48k_specaugment_label_smoothing": classifier(cnn14(specaugment=True), mixup=False, label_smoothing=0.15),
    "cls_panns_pt": classifier(hear("PANNs"), mixup=False),
    "cls_vggish": classifier(hear("VGGish")),
    "cls_wav2clip": classifier(hear("Wav2CLIP")),
    "cls_wav2vec2": classifier(hear("wav2vec2")),
}


def write_models():
    for name, net in REMOVAL.items():
        _emit(f"model/{name}.yaml", {"model": remfx(net)},
              header=f"model={name}: RemFX wrapper (AdamW 1e-4, betas 0.95 / 0.999, eps 1e-6, wd 1e-3) around {net['_target_']}")
    for name, node in CLASSIFIERS.items():
        _emit(f"model/{name}.yaml", {"model": node},
              header=f"model={name}: FXClassifier (AdamW 3e-4, wd 1e-3) around {node['network']['_target_']}")


# ---- effects / logger --------------------------------------------------------------------------------------------------------------
def write_effects_logger():
    fx = lambda cls, **kw: dict({"_target_": f"remfx.effects.RandomPedalboard{cls}", "sample_rate": SR}, **kw)
    body = {"effects": {
        "chorus": fx("Chorus", min_rate_hz=0.25, max_rate_hz=1.5, min_feedback=0.1, max_feedback=0.4, min_depth=0.2, max_depth=0.6,
                     min_mix=0.15, max_mix=0.4),
        "distortion": fx("Distortion", min_drive_db=8, max_drive_db=25),
        "compressor": fx("Compressor", min_threshold_db=-42.0, max_threshold_db=-20.0, min_ratio=1.5, max_ratio=6.0),
        "reverb": fx("Reverb", min_room_size=0.3, max_room_size=1.0, min_damping=0.2, max_damping=1.0, min_wet_dry=0.2,
                     max_wet_dry=0.6, min_width=0.2, max_width=1.0),
        # `max_delay_sconds` is the reference's own spelling of the keyword (remfx/effects.py RandomPedalboardDelay): API, kept
        "delay": fx("Delay", min_delay_seconds=0.1, max_delay_sconds=1.0, min_feedback=0.05, max_feedback=0.3, min_mix=0.1,
                    max_mix=0.35)}}
    _emit("effects/all.yaml", body, header="effects=all: parameter ranges of the five pedalboard effects the experiments draw from")
    _emit("logger/csv.yaml", {"logger": {"_target_": "pytorch_lightning.loggers.CSVLogger", "save_dir": ".",
                                         "version": "${now:%Y-%m-%d-%H-%M-%S}"}})
    _emit("logger/wandb.yaml", {"logger": {"_target_": "pytorch_lightning.loggers.WandbLogger", "project": "${oc.env:WANDB_PROJECT}",
                                           "entity": "${oc.env:WANDB_ENTITY}", "job_type": "train", "group": "", "save_dir": ".",
                                           "log_model": True}},
          header="logger=wandb: resolves only where W&B is installed (observability: outside the hot path, SURVEY 2.1)")


# ---- experiments -----------------------------------------------------------------------------------------------------------------
def experiment(model, removed, n_removed, kept=None, n_kept=(0, 0), shuffle_removed=True, num_classes=5, accelerator="gpu",
               log_audio=True, render_files=True, render_root=None, dm=None):
    body = {"defaults": [{"override /model": model}, {"override /effects": "all"}],
            "seed": 12345, "sample_rate": 48000, "chunk_size": 262144, "logs_dir": "./logs"}
    if render_files is not None:
        body["render_files"] = render_files
    if render_root is not None:
        body["render_root"] = render_root
    body.update({"accelerator": accelerator, "log_audio": log_audio,
                 "num_kept_effects": list(n_kept), "num_removed_effects": list(n_removed),
                 "shuffle_kept_effects": True, "shuffle_removed_effects": shuffle_removed, "num_classes": num_classes,
                 "effects_to_keep": kept, "effects_to_remove": list(removed),
                 "datamodule": dm or {"train_batch_size": 16, "test_batch_size": 1, "num_workers": 8}})
    return body


def chain(ckpt_suffix, shuffle, use_all, with_classifier, dm=None):
    """Chain-inference experiments (scripts/chain_inference.py, remfx_detect.py; remfx/models.py:22-149): one removal model per
    effect -- Hybrid Demucs (= ${model}) for distortion / compressor, DCUNet for reverb / chorus / delay -- and optionally the
    Cnn14 detector."""
    body = experiment("demucs", FX, (0, 5), render_files=None, dm=dm)
    body["dcunet"] = remfx(DCUNET_NET)
    if with_classifier:
        body["classifier"] = classifier(cnn14(specaugment=True), mixup=False)
        body["classifier_ckpt"] = "ckpts/classifier.ckpt"
    arch = {"Distortion": "demucs", "Compressor": "demucs", "Reverb": "dcunet", "Chorus": "dcunet", "Delay": "dcunet"}
    body["ckpts"] = {f"RandomPedalboard{e}": {"model": "${model}" if a == "demucs" else "${dcunet}",
                                               "ckpt_path": f"ckpts/{a}_{e.lower()}{ckpt_suffix}.ckpt"} for e, a in arch.items()}
    body["inference_effects_ordering"] = [f"RandomPedalboard{e}" for e in arch]
    body.update({"num_bins": 1025, "inference_effects_shuffle": shuffle, "inference_use_all_effect_models": use_all})
    return body


def write_experiments():
    exps = {}
    # N effects applied, all of them removed (the monolithic-network experiments): "<pool>-<N>"
    for name, rng in {"0-0": (0, 0), "1-1": (1, 1), "2-2": (2, 2), "3-3": (3, 3), "4-4": (4, 4), "5-1": (1, 1), "5-5": (5, 5),
                      "5-5_full": (0, 5)}.items():
        exps[name] = experiment("demucs", FX, rng)
    # single-effect removal, clean input chain / with up to four distractor effects kept ("_aug"); default network per effect
    single = {"chorus": ("dcunet", "chorus", ["compressor", "distortion", "delay", "reverb"]),
              "compression": ("demucs", "compressor", ["distortion", "chorus", "delay", "reverb"]),
              "delay": ("dcunet", "delay", ["compressor", "distortion", "chorus", "reverb"]),
              "distortion": ("demucs", "distortion", ["compressor", "reverb", "chorus", "delay"]),
              "reverb": ("dcunet", "reverb", ["compressor", "distortion", "chorus", "delay"])}
    for name, (model, effect, distractors) in single.items():
        exps[name] = experiment(model, [effect], (1, 1), shuffle_removed=False, num_classes=1)
        exps[name + "_aug"] = experiment(model, [effect], (1, 1), kept=distractors, n_kept=(0, 4), shuffle_removed=False)
    exps["default"] = experiment("umx", ["compressor", "reverb", "chorus", "delay", "distortion"], (0, 5), shuffle_removed=False,
                                 accelerator=None, render_root="./data")
    # effect classifier
    cls_extra = {"callbacks": checkpoint_callbacks("valid_avg_acc_epoch", "max", True, False), "trainer": trainer(300, -1)}
    exps["5-5_full_cls"] = dict(experiment("cls_panns_48k_specaugment", FX, (0, 5), log_audio=False,
                                           dm={"train_batch_size": 64, "test_batch_size": 256, "num_workers": 8}), **cls_extra)
    dyn = {"_target_": "remfx.datasets.EffectDatamodule",
           "train_dataset": dataset("train", 8000, cls="DynamicEffectDataset", parallel=True),
           "val_dataset": dataset("val", 1000, parallel=None), "test_dataset": dataset("test", 1000, parallel=None),
           "train_batch_size": 32, "test_batch_size": 256, "num_workers": 12}
    exps["5-5_full_cls_dynamic"] = dict(experiment("demucs", FX, (0, 5), log_audio=False, dm=dyn), **cls_extra)
    # chain inference
    exps["chain_inference"] = chain("", False, False, False)
    exps["chain_inference_aug"] = chain("_aug", False, False, False)
    exps["chain_inference_aug_classifier"] = chain("_aug", False, False, True)
    exps["chain_inference_custom"] = chain("_aug", False, False, False, dm={
        "train_batch_size": 1, "test_batch_size": 1, "num_workers": 8, "train_dataset": "None", "val_dataset": "None",
        "test_dataset": {"_target_": "remfx.datasets.InferenceDataset", "root": "${oc.env:DATASET_ROOT}", "sample_rate": SR}})
    exps["remfx_oracle"] = chain("_aug", True, False, False)
    exps["remfx_detect"] = chain("_aug", True, False, True)
    exps["remfx_all"] = chain("_aug", True, True, True)
    for name, body in exps.items():
        _emit(f"exp/{name}.yaml", body, header=f"+exp={name}")
    return sorted(exps)


def main():
    write_primary()
    write_models()
    write_effects_logger()
    names = write_experiments()
    print("wrote cfg/: config.yaml,", len(REMOVAL) + len(CLASSIFIERS), "model files,", len(names), "experiments")


if __name__ == "__main__":
    main()

"""GPU: scripts/train.py end to end through the config composer, the mini trainer, the flat AdamW
kernel and checkpointing; AdamW + clip step vs torch.optim.AdamW on the same gradients."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def test_train_script_tcn_two_steps(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "train.py"), "+exp=reverb", "model=tcn",
                        "model.network.nblocks=3", "model.network.channel_width=16", "chunk_size=16384",
                        "datamodule.train_batch_size=2", "datamodule.train_dataset.total_chunks=4",
                        "datamodule.val_dataset.total_chunks=2", "datamodule.test_dataset.total_chunks=2", "trainer.max_steps=2",
                        f"logs_dir={tmp_path}", f"logger.save_dir={tmp_path}"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "train_loss" in r.stdout and "valid_loss" in r.stdout and "test_loss" in r.stdout
    ck = torch.load(os.path.join(tmp_path, "ckpts", "last.ckpt"), map_location="cpu", weights_only=False)
    assert "state_dict" in ck and "model.model.process_blocks.0.conv1.weight" in ck["state_dict"]
    import glob
    assert len(glob.glob(os.path.join(tmp_path, "lightning_logs", "*", "metrics.csv"))) == 1     # cfg/logger/csv.yaml: <save_dir>/lightning_logs/<stamp>/


def test_flat_adamw_matches_torch():
    from remfx_amd.optim import FlatAdamW, FlatParams, MultiStepLR
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(33, 17), torch.nn.Tanh(), torch.nn.Linear(17, 5)).to(DEV)
    ref = torch.nn.Sequential(torch.nn.Linear(33, 17), torch.nn.Tanh(), torch.nn.Linear(17, 5)).to(DEV)
    ref.load_state_dict(net.state_dict())
    opt = FlatAdamW(FlatParams(list(net.parameters())), lr=1e-2, betas=(0.95, 0.999), eps=1e-6, weight_decay=1e-3)
    sched = MultiStepLR(opt, [2, 3], gamma=0.1)
    topt = torch.optim.AdamW(ref.parameters(), lr=1e-2, betas=(0.95, 0.999), eps=1e-6, weight_decay=1e-3)
    tsched = torch.optim.lr_scheduler.MultiStepLR(topt, [2, 3], gamma=0.1)
    x = torch.randn(64, 33, device=DEV) * 3
    for step in range(4):
        opt.zero_grad(); topt.zero_grad()
        (net(x).pow(2).sum() * 10).backward()
        (ref(x).pow(2).sum() * 10).backward()
        total = torch.nn.utils.clip_grad_norm_(ref.parameters(), 10.0)
        opt.step(clip_norm=10.0); topt.step()
        sched.step(); tsched.step()
        assert abs(float(opt.last_grad_norm) - float(total)) < 1e-3 * float(total)
        assert abs(opt.param_groups[0]["lr"] - topt.param_groups[0]["lr"]) < 1e-12
        for p, q in zip(net.parameters(), ref.parameters()):
            assert float((p - q).abs().max()) < 2e-6, step


def test_chain_inference_script_small(tmp_path):
    """scripts/chain_inference.py +exp=remfx_detect on short clips (random init, no checkpoints offline):
    detector + 5 removal networks + metrics run end to end."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "chain_inference.py"), "+exp=remfx_detect",
                        "chunk_size=32768", "datamodule.test_batch_size=3", "datamodule.test_dataset.total_chunks=3",
                        "inference_use_all_effect_models=True", f"logs_dir={tmp_path}"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT,
                       env=dict(os.environ, RFX_ALLOW_RANDOM_INIT="1"))      # no released checkpoint is reachable offline
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    for k in ("test_loss", "test_SISDR", "test_STFT", "Input_SISDR", "Input_STFT"):
        assert k in r.stdout, k


@pytest.mark.one_mode
def test_train_script_umx_config1(tmp_path):
    """BASELINE config 1 (`+exp=distortion model=umx`, 4 clips, one train step) through scripts/train.py.  The reference
    runs it on the CPU (`accelerator=null`); this build has no CPU compute path by design, so the same command line runs
    with accelerator=gpu."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "train.py"), "+exp=distortion", "model=umx",
                        "accelerator=gpu", "chunk_size=32768", "datamodule.train_batch_size=4",
                        "datamodule.train_dataset.total_chunks=4", "datamodule.val_dataset.total_chunks=2",
                        "datamodule.test_dataset.total_chunks=2", "trainer.max_steps=1", f"logs_dir={tmp_path}"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "train_loss" in r.stdout and "valid_loss" in r.stdout
    ck = torch.load(os.path.join(tmp_path, "ckpts", "last.ckpt"), map_location="cpu", weights_only=False)
    assert ck["global_step"] == 1 and len(ck["optimizer_states"]) == 1 and len(ck["lr_schedulers"]) == 1
    assert any(k.startswith("model.separator.") for k in ck["state_dict"])          # duplicate registration, as upstream


@pytest.mark.one_mode
def test_train_script_demucs_bf16_mixed_config3(tmp_path):
    """BASELINE config 3's command line (`+exp=chorus_aug model=demucs trainer.precision=bf16-mixed`), reduced width and
    clip length: the trainer switches the GEMMs to bf16 operands, trains two steps, resumes from its own checkpoint."""
    base = [sys.executable, os.path.join(ROOT, "scripts", "train.py"), "+exp=chorus_aug", "model=demucs",
            "trainer.precision=bf16-mixed", "model.network.channels=8", "chunk_size=32768", "datamodule.train_batch_size=2",
            "datamodule.train_dataset.total_chunks=4", "datamodule.val_dataset.total_chunks=2",
            "datamodule.test_dataset.total_chunks=2", f"logs_dir={tmp_path}"]
    r = subprocess.run(base + ["trainer.max_steps=2"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "train_loss" in r.stdout
    ck = torch.load(os.path.join(tmp_path, "ckpts", "last.ckpt"), map_location="cpu", weights_only=False)
    assert ck["global_step"] == 2
    st = ck["optimizer_states"][0]["state"]
    assert float(st[0]["step"]) == 2.0 and st[0]["exp_avg"].abs().sum() > 0
    r = subprocess.run(base + ["trainer.max_steps=3", "+ckpt_path=" + os.path.join(tmp_path, "ckpts", "last.ckpt")],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    ck2 = torch.load(os.path.join(tmp_path, "ckpts", "last.ckpt"), map_location="cpu", weights_only=False)
    assert ck2["global_step"] == 3 and float(ck2["optimizer_states"][0]["state"][0]["step"]) == 3.0      # one more step, not three


def test_gradient_sink_matches_autograd_accumulation():
    """Parameter gradients written straight into the flat gradient buffer (ops.GradSink: conv weight / bias gradients on
    the side stream, GroupNorm / LayerScale gradients by their kernels) equal the gradients autograd's accumulation
    produces for the same network, input and upstream gradient -- on a small Hybrid Demucs, whose backward reaches every
    kind of sink (conv, fork-conv, GLU-conv, DConv GroupNorms, LSTM / attention projections)."""
    from remfx_amd import ops
    from remfx_amd.hdemucs import HDemucs
    from remfx_amd.optim import FlatParams
    torch.manual_seed(5)
    net = HDemucs(sources=["mixture"], audio_channels=1, nfft=4096, channels=8).to(DEV)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith(".scale"):
                p.fill_(0.3)
    g = torch.Generator().manual_seed(6)
    x = (torch.randn(2, 1, 20000, generator=g) * 0.5).to(DEV)
    gy = torch.randn(2, 1, 1, 20000, generator=g).to(DEV)
    net(x).backward(gy)                                           # plain autograd accumulation
    ref = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    flat = FlatParams(list(net.parameters()))
    assert flat.sink is not None
    for rep in range(2):                                          # twice: the buffer is re-zeroed, the sink re-armed
        flat.zero_grad()
        assert ops.SINK is flat.sink
        net(x).backward(gy)
        nsunk = sum(1 for w in flat.sink.writes if w)
        flat.join()
        assert ops.SINK is None
        torch.cuda.synchronize()
        assert nsunk > 100, nsunk                                 # most of the 397 tensors go through the sink
        num = den = 0.0
        for n, p in net.named_parameters():
            if n in ref:
                d = float((p.grad - ref[n]).double().pow(2).sum())
                num += d; den += float(ref[n].double().pow(2).sum())
                assert d ** 0.5 <= 1e-4 * max(1e-6, float(ref[n].abs().max())) * ref[n].numel() ** 0.5, n
        assert (num / den) ** 0.5 < 1e-5, (num / den) ** 0.5       # fp32 atomics order noise only


def test_test_script_loads_checkpoint_and_evaluates(tmp_path):
    """scripts/test.py (reference scripts/test.py:10-47): strict load of the checkpoint scripts/train.py wrote, then the test
    split -- same metrics as the post-fit test of the training run that produced the checkpoint (same weights, same data)."""
    common = ["+exp=reverb", "model=tcn", "model.network.nblocks=3", "model.network.channel_width=16", "chunk_size=16384",
              "datamodule.train_batch_size=2", "datamodule.train_dataset.total_chunks=4", "datamodule.val_dataset.total_chunks=2",
              "datamodule.test_dataset.total_chunks=2", f"logs_dir={tmp_path}"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "train.py")] + common + ["trainer.max_steps=2"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    fit_test = eval(r.stdout.strip().splitlines()[-1])
    ck = os.path.join(tmp_path, "ckpts", "last.ckpt")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "test.py")] + common + [f"+ckpt_path={ck}", "datamodule.train_dataset=None", "datamodule.val_dataset=None"],
                        capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-3000:]
    out = eval(r2.stdout.strip().splitlines()[-1])
    for k in ("test_loss", "test_SISDR", "test_STFT", "Input_SISDR", "Input_STFT"):
        assert abs(out[k] - fit_test[k]) <= 1e-4 * max(1.0, abs(fit_test[k])), (k, out[k], fit_test[k])
    # a missing checkpoint is an error, as upstream (torch.load raises)
    r3 = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "test.py")] + common + ["+ckpt_path=/nonexistent.ckpt"],
                        capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r3.returncode != 0 and "not found" in r3.stderr


@pytest.mark.one_mode
def test_remfx_detect_script_whole_file(tmp_path):
    """scripts/remfx_detect.py (reference scripts/remfx_detect.py:13-61) on a 44.1 kHz stereo file LONGER than one training clip
    (7 s -> 336000 samples at 48 kHz: more than 256 attention frames): device-side resampling, mono mix, detector + the five
    removal networks on the whole file, float32 WAV out at cfg.sample_rate with the input's resampled length."""
    import numpy as np
    from scipy.io import wavfile
    sr_in, secs = 44100, 7.0
    t = np.arange(int(sr_in * secs)) / sr_in
    rng = np.random.default_rng(0)
    a = np.stack([0.3 * np.sin(2 * np.pi * 220.0 * t) + 0.02 * rng.standard_normal(t.size),
                  0.2 * np.sin(2 * np.pi * 331.0 * t)], 1)
    src, dst = os.path.join(tmp_path, "in.wav"), os.path.join(tmp_path, "out.wav")
    wavfile.write(src, sr_in, (a * 32767).astype(np.int16))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "remfx_detect.py"), "+exp=remfx_detect",
                        f"+audio_input={src}", f"+output_path={dst}", "inference_use_all_effect_models=True"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, RFX_ALLOW_RANDOM_INIT="1"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "Loading models..." in r.stdout and "Saving output to" in r.stdout
    sr_out, y = wavfile.read(dst)
    assert sr_out == 48000 and y.ndim == 1 or y.shape[1] == 1
    n = int(np.ceil(a.shape[0] * 48000 / sr_in))
    assert abs(y.shape[0] - n) <= 1 and np.isfinite(y).all() and y.dtype == np.float32

"""GPU parity: HIP Open-Unmix + Separator vs the CPU oracle restatement (same state_dict)."""
import pytest
import torch

from tests.conftest import check, mode, tol

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rms(a, b):
    return float(((a - b) ** 2).mean().sqrt())


def _pair():
    from oracle import ref_umx
    from remfx_amd.umx import OpenUnmix
    torch.manual_seed(0)
    ref = ref_umx.OpenUnmix(nb_bins=1025, nb_channels=1)
    with torch.no_grad():
        g = torch.Generator().manual_seed(1)
        for n, p in ref.named_parameters():
            if n in ("input_mean", "input_scale", "output_scale", "output_mean"):
                p.add_(torch.randn(p.shape, generator=g) * 0.05)
        for n, b in ref.named_buffers():
            if n.endswith("running_mean"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.05)
            elif n.endswith("running_var"):
                b.copy_(torch.rand(b.shape, generator=g) * 0.5 + 0.75)
    net = OpenUnmix(nb_bins=1025, nb_channels=1)
    net.load_state_dict(ref.state_dict(), strict=True)
    return ref, net.to(DEV)


def test_umx_separator_eval():
    from oracle import ref_umx
    from remfx_amd.umx import Separator
    ref, net = _pair()
    ref.eval(); net.eval()
    x = torch.randn(2, 1, 30000, generator=torch.Generator().manual_seed(2)) * 0.3
    with torch.no_grad():
        y = ref_umx.separator(ref, x)
        sep = Separator(target_models={"other": net}, nb_channels=1, sample_rate=48000, n_fft=2048, n_hop=512).to(DEV)
        yd = sep(x.to(DEV)).cpu()
    assert yd.shape == y.shape == (2, 1, 1, 30000)
    check(_rms(yd, y), 1e-4, max(1.0, float(y.abs().max())), what=_rms(yd, y))


def test_umx_train_fwd_bwd():
    """train-mode BatchNorm statistics; LSTM inter-layer dropout disabled on both sides (random)."""
    from oracle import ref_umx
    from remfx_amd.umx import Separator
    ref, net = _pair()
    ref.train(); net.train()
    ref.lstm.dropout = 0.0; net.lstm.dropout = 0.0
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 1, 20000, generator=g) * 0.3
    y = ref_umx.separator(ref, x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    sep = Separator(target_models={"other": net}, nb_channels=1, sample_rate=48000, n_fft=2048, n_hop=512).to(DEV)
    yd = sep(x.to(DEV))
    check(_rms(yd.detach().cpu(), y.detach()), 1e-4, max(1.0, float(y.detach().abs().max())))
    yd.backward(gy.to(DEV))
    refg = dict(ref.named_parameters())
    num = den = 0.0
    for n, p in net.named_parameters():
        r = refg[n].grad
        d = p.grad.cpu() - r
        num += float((d ** 2).sum()); den += float((r ** 2).sum())
    check((num / den) ** 0.5, 2e-3, what=(num / den) ** 0.5)


def test_openunmix_model_wrapper():
    """OpenUnmixModel: duplicate registration (model + separator.target_models.other), dead Y pass, loss."""
    from remfx_amd.models import OpenUnmixModel
    torch.manual_seed(0)
    m = OpenUnmixModel(n_fft=2048, hop_length=512, n_channels=1, alpha=0.3, sample_rate=48000).to(DEV)
    keys = list(m.state_dict())
    assert "model.fc1.weight" in keys and "separator.target_models.other.fc1.weight" in keys and "window" in keys
    x = torch.randn(2, 1, 32768, device=DEV) * 0.1
    t = torch.randn(2, 1, 32768, device=DEV) * 0.1
    before = int(m.model.bn1.num_batches_tracked)
    loss, out = m((x, t))
    assert out.shape == (2, 1, 32768) and torch.isfinite(loss)
    assert int(m.model.bn1.num_batches_tracked) == before + 2          # Q3: BN stats updated twice per step
    loss.backward()
    assert m.model.fc1.weight.grad is not None and torch.isfinite(m.model.fc1.weight.grad).all()


def test_umx_full_length_golden(golden_dir):
    """BASELINE config 1 at its real length: 262144-sample clips = 513 STFT frames = the 513-step 3-layer BiLSTM, eval forward and
    train-mode forward + backward against tests/golden/umx_full.npz (oracle/gen_full_length_golden.py: the CPU oracle run once on
    the seeded weights of `_pair` and a seeded pair of clips)."""
    import os
    import numpy as np
    from remfx_amd.umx import Separator
    gd = np.load(os.path.join(golden_dir, "umx_full.npz"))
    g = torch.Generator().manual_seed(22)
    x = torch.randn(2, 1, 262144, generator=g) * 0.3
    _, net = _pair()
    net.eval()
    sep = Separator(target_models={"other": net}, nb_channels=1, sample_rate=48000, n_fft=2048, n_hop=512).to(DEV)
    with torch.no_grad():
        yd = sep(x.to(DEV)).cpu()
    assert yd.shape == (2, 1, 1, 262144)
    sl = lambda t, n=2048: t.detach().reshape(-1)[::max(1, t.numel() // n)][:n].numpy()
    e = float(np.sqrt(((sl(yd) - gd["eval_y_slice"]) ** 2).mean()))
    check(e, 1e-4, max(1.0, float(gd["eval_y_absmax"])), what=("eval", e))
    check(abs(float(yd.double().norm()) - float(gd["eval_y_norm"])), 1e-4, float(gd["eval_y_norm"]), what="eval norm")
    _, net = _pair()
    net.train()
    net.lstm.dropout = 0.0
    sep = Separator(target_models={"other": net}, nb_channels=1, sample_rate=48000, n_fft=2048, n_hop=512).to(DEV)
    gy = torch.randn(yd.shape, generator=g)
    yt = sep(x.to(DEV))
    e = float(np.sqrt(((sl(yt.cpu()) - gd["train_y_slice"]) ** 2).mean()))
    check(e, 1e-4, max(1.0, float(gd["train_y_absmax"])), what=("train", e))
    yt.backward(gy.to(DEV))
    params = dict(net.named_parameters())
    tot = sum(float(params[n].grad.double().pow(2).sum()) for n in gd["names"].tolist()) ** 0.5
    check(abs(tot - float(gd["grad_global_norm"])), 2e-3, float(gd["grad_global_norm"]), bf16=5e-2, what="global grad norm")
    num = den = 0.0
    for i, n in enumerate(gd["names"].tolist()):
        got, ref = sl(params[n].grad.cpu(), 512), gd[f"g{i}_slice"]
        num += float(((got - ref) ** 2).sum()); den += float((ref ** 2).sum())
        check(abs(float(params[n].grad.double().norm()) - float(gd[f"g{i}_norm"])), 1e-2, max(float(gd[f"g{i}_norm"]), 1e-3 * tot),
              bf16=0.1, what=(n, "norm"))
    rel = (num / den) ** 0.5
    print(f"Open-Unmix full-length gradients (513 recurrence steps): slice-wise global relative error {rel:.2e} [{mode()}]")
    check(rel, 2e-3, what=("grad slices", rel))

"""GPU parity: HIP DCUNet (Large-DCUNet-20) vs the CPU oracle restatement (same state_dict)."""
import pytest
import torch

from tests.conftest import check, mode, tol

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rms(a, b):
    return float(((a - b) ** 2).mean().sqrt())


def _pair(train):
    from oracle import ref_dcunet
    from remfx_amd.dcunet import DCUNet
    torch.manual_seed(0)
    ref = ref_dcunet.DCUNet(stft_kernel_size=512, fix_length_mode="pad")
    with torch.no_grad():      # non-trivial running statistics / biases so eval mode is exercised too
        g = torch.Generator().manual_seed(1)
        for n, b in ref.named_buffers():
            if n.endswith(("RMr", "RMi")):
                b.copy_(torch.randn(b.shape, generator=g) * 0.05)
            elif n.endswith(("RVrr", "RVii")):
                b.copy_(torch.rand(b.shape, generator=g) * 0.5 + 0.75)
            elif n.endswith("RVri"):
                b.copy_(torch.randn(b.shape, generator=g) * 0.05)
        for n, p in ref.named_parameters():
            if n.endswith((".Br", ".Bi")):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
    net = DCUNet(stft_kernel_size=512, fix_length_mode="pad")
    net.load_state_dict(ref.state_dict(), strict=True)
    ref.train(train); net.train(train)
    return ref, net.to(DEV)


def test_dcunet_eval_forward():
    ref, net = _pair(train=False)
    x = torch.randn(2, 20000, generator=torch.Generator().manual_seed(2)) * 0.3
    with torch.no_grad():
        y = ref(x)
        yd = net(x.to(DEV)).cpu()
    assert yd.shape == y.shape == (2, 1, 20000)
    check(_rms(yd, y), 1e-4, max(1.0, float(y.abs().max())), what=_rms(yd, y))
    # BASELINE config 4's quality figure: SI-SDR (auraloss definition) of the device output against the CPU restatement
    from oracle import ref_losses
    sisdr = -float(ref_losses.sisdr_loss(yd, y))
    print(f"DCUNet eval forward: SI-SDR(device output, CPU restatement) = {sisdr:.1f} dB [{mode()}]")
    assert sisdr > tol(80.0, bf16x3=70.0, bf16=25.0), sisdr


@pytest.mark.one_mode
def test_dcunet_zero_padded_filterbank_form():
    """The alternative reading of asteroid's STFTFB buffer shape (window zero-padded to n_filters): (1026, 1, 1024) buffers,
    strict state_dict round trip between oracle and product, same output parity."""
    from oracle import ref_dcunet
    from remfx_amd.dcunet import DCUNet
    torch.manual_seed(4)
    ref = ref_dcunet.DCUNet(stft_kernel_size=512, fix_length_mode="pad", stft_filter_form="zero_pad").eval()
    assert tuple(ref.state_dict()["encoder.filterbank._filters"].shape) == (1026, 1, 1024)
    net = DCUNet(stft_kernel_size=512, fix_length_mode="pad", stft_filter_form="zero_pad")
    net.load_state_dict(ref.state_dict(), strict=True)
    net = net.to(DEV).eval()
    with pytest.raises(RuntimeError):                       # the two forms are not interchangeable under a strict load
        DCUNet(stft_kernel_size=512, fix_length_mode="pad").load_state_dict(ref.state_dict(), strict=True)
    x = torch.randn(1, 20000, generator=torch.Generator().manual_seed(5)) * 0.3
    with torch.no_grad():
        y = ref(x)
        yd = net(x.to(DEV)).cpu()
    check(_rms(yd, y), 1e-4, max(1.0, float(y.abs().max())))


_F64 = {}


def _train_case():
    """Train-mode (batch-statistic complex BatchNorm) forward + backward of the CPU oracle in fp32 AND fp64 on the same
    weights / input (computed once per session; the fp64 pass is the ground truth both fp32 paths are measured against)."""
    if not _F64:
        import copy
        ref, net = _pair(train=True)
        g = torch.Generator().manual_seed(3)
        x = torch.randn(2, 50000, generator=g) * 0.3
        ref64 = copy.deepcopy(ref).double()
        y = ref(x)
        gy = torch.randn(y.shape, generator=g)
        y.backward(gy)
        y64 = ref64(x.double())
        y64.backward(gy.double())
        _F64.update(ref=ref, sd={k: v.clone() for k, v in net.state_dict().items()}, x=x, gy=gy, y=y.detach(), y64=y64.detach(),
                    g32={n: p.grad for n, p in ref.named_parameters()}, g64={n: p.grad for n, p in ref64.named_parameters()})
    return _F64


def _global_rel(got, truth):
    num = sum(float((got[n].double().cpu() - truth[n]).pow(2).sum()) for n in truth)
    den = sum(float(truth[n].pow(2).sum()) for n in truth)
    return (num / den) ** 0.5


def test_dcunet_train_fwd_bwd():
    """Train-mode gradients.  The batch-statistic 2x2 whitening of 3 x W maps is ill-conditioned, so fp32 itself is noisy
    here: the CPU oracle in fp32 sits 4.3e-3 (global relative) from its own fp64 run.  The parity statement is therefore
    made against the fp64 ground truth: the exact-fp32 HIP path must be no further from it than 2 x the CPU fp32 path is."""
    from remfx_amd.dcunet import DCUNet
    c = _train_case()
    ref, x, gy, y = c["ref"], c["x"], c["gy"], c["y"]
    net = DCUNet(stft_kernel_size=512, fix_length_mode="pad")
    net.load_state_dict(c["sd"], strict=True)
    net = net.to(DEV).train()
    yd = net(x.to(DEV))
    check(_rms(yd.detach().cpu(), y), 1e-4, max(1.0, float(y.abs().max())))
    check(_rms(yd.detach().cpu().double(), c["y64"]), 1e-4, max(1.0, float(y.abs().max())), what="fwd vs fp64")
    yd.backward(gy.to(DEV))
    got = {n: p.grad for n, p in net.named_parameters()}
    for n, p in net.named_parameters():
        r = c["g64"][n]
        check(_rms(p.grad.cpu().double(), r), 5e-2, max(1e-6, float(r.abs().max())), what=n)
    e_cpu = _global_rel(c["g32"], c["g64"])                 # CPU oracle fp32 vs fp64: the conditioning of the problem
    e_hip = _global_rel(got, c["g64"])
    e_pair = _global_rel(got, {n: v.double() for n, v in c["g32"].items()})
    print(f"DCUNet train-mode gradient, global relative error vs the fp64 oracle: CPU fp32 oracle {e_cpu:.2e}, HIP [{mode()}] "
          f"{e_hip:.2e}; HIP vs CPU fp32 oracle {e_pair:.2e}")
    assert 1e-4 < e_cpu < 2e-2, e_cpu                     # the premise: fp32 is this noisy on the CPU too
    if mode() == "f32":
        assert e_hip < 2.0 * e_cpu, (e_hip, e_cpu)
    else:
        # bf16x3: 2^-17 product rounding through the same conditioning (measured 1.2e-2); bf16: 2^-9 operands (0.22)
        check(e_hip, 5e-3, bf16x3=2e-2, bf16=0.4, what=e_hip)
    # running statistics updated identically (momentum 0.1 lerp)
    rb = dict(ref.named_buffers())
    for n, b in net.named_buffers():
        if n.endswith(("RMr", "RVrr", "RVri")):
            check(_rms(b.cpu(), rb[n]), 1e-5, max(1.0, float(rb[n].abs().max())), what=n)


def test_dcunet_model_wrapper_full_length():
    """DCUNetModel on a full 262144-sample clip: output (B, 1, T), finite loss with gradient."""
    from remfx_amd.models import DCUNetModel
    torch.manual_seed(0)
    m = DCUNetModel(48000, 1025, architecture="Large-DCUNet-20", stft_kernel_size=512, fix_length_mode="pad").to(DEV)
    x = torch.randn(1, 1, 262144, device=DEV) * 0.1
    t = torch.randn(1, 1, 262144, device=DEV) * 0.1
    loss, out = m((x, t))
    assert out.shape == (1, 1, 262144) and torch.isfinite(loss)
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())


def test_dcunet_full_length_golden(golden_dir):
    """BASELINE config 4 at its real length: one 262144-sample clip = 1023 STFT frames through Large-DCUNet-20, eval forward and
    train-mode forward + backward against tests/golden/dcunet_full.npz (oracle/gen_full_length_golden.py).  The train-mode
    expectations come from an fp64 run of the oracle; `cpu_fp32_vs_fp64_global_rel` is how far its own fp32 run sits from it."""
    import os
    import numpy as np
    gd = np.load(os.path.join(golden_dir, "dcunet_full.npz"))
    g = torch.Generator().manual_seed(21)
    x = torch.randn(1, 262144, generator=g) * 0.3
    sl = lambda t, n=2048: t.detach().reshape(-1)[::max(1, t.numel() // n)][:n].numpy()
    _, net = _pair(train=False)
    with torch.no_grad():
        yd = net(x.to(DEV)).cpu()
    assert yd.shape == (1, 1, 262144)
    e = float(np.sqrt(((sl(yd) - gd["eval_y_slice"]) ** 2).mean()))
    check(e, 1e-4, max(1.0, float(gd["eval_y_absmax"])), what=("eval", e))
    check(abs(float(yd.double().norm()) - float(gd["eval_y_norm"])), 1e-4, float(gd["eval_y_norm"]), bf16=2e-2, what="eval norm")
    _, net = _pair(train=True)
    gy = torch.randn(yd.shape, generator=g)
    yt = net(x.to(DEV))
    e = float(np.sqrt(((sl(yt.cpu()) - gd["train_y_slice"]) ** 2).mean()))
    check(e, 1e-4, max(1.0, float(gd["train_y_absmax"])), what=("train", e))
    yt.backward(gy.to(DEV))
    params = dict(net.named_parameters())
    tot = sum(float(p.grad.double().pow(2).sum()) for p in params.values()) ** 0.5
    e_cpu = float(gd["cpu_fp32_vs_fp64_global_rel"])
    check(abs(tot - float(gd["grad_global_norm"])), max(2e-3, 2 * e_cpu), float(gd["grad_global_norm"]), bf16x3=2e-2, bf16=0.4,
          what="global grad norm")
    num = den = 0.0
    for i, n in enumerate(gd["names"].tolist()):
        got, ref = sl(params[n].grad.cpu(), 512), gd[f"g{i}_slice"]
        num += float(((got - ref) ** 2).sum()); den += float((ref ** 2).sum())
    rel = (num / den) ** 0.5
    print(f"DCUNet full-length train-mode gradients vs the fp64 oracle: slice-wise global relative error {rel:.2e} [{mode()}]; "
          f"the CPU fp32 oracle itself: {e_cpu:.2e}")
    if mode() == "f32":
        assert rel < max(2.0 * e_cpu, 5e-3), (rel, e_cpu)
    else:
        check(rel, 5e-3, bf16x3=max(2e-2, 4 * e_cpu), bf16=0.4, what=("grad slices", rel))
    rb = dict(net.named_buffers())
    for k, n in enumerate(gd["rm_names"].tolist()):
        check(_rms(rb[n].cpu(), torch.from_numpy(gd[f"rm{k}"])), 1e-5, max(1.0, float(np.abs(gd[f"rm{k}"]).max())), bf16x3=1e-4, what=n)

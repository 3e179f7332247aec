"""GPU: the arithmetic modes of the gather-GEMM vs fp32 references on layer shapes with >= 8 input channels (the
tap-major bf16x3 / bf16 kernels) and fewer (channel-major exact-fp32 kernel in every mode).  bf16x3 error budget:
~2^-16 relative per product -> outputs within ~3e-5 of their scale.  Modes: tests/conftest.py."""
import numpy as np
import os
import pytest
import torch

from tests.conftest import check, mode, tol
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rms(a, b):
    return float(((a - b) ** 2).mean().sqrt())


CASES = [
    (16, 48, (1, 3000), (1, 7), (1, 1), (0, 0), (1, 16), 2),
    (64, 256, (1, 2100), (1, 7), (1, 1), (0, 0), (1, 2), 1),
    (48, 96, (32, 40), (8, 1), (4, 1), (2, 0), (1, 1), 2),
    (24, 33, (17, 23), (3, 3), (1, 1), (1, 1), (1, 1), 2),
    (8, 45, (40, 30), (7, 5), (2, 2), (3, 2), (1, 1), 2),
    (3, 20, (1, 500), (1, 7), (1, 1), (0, 0), (1, 4), 2),      # K = 21: padded K step
    # unit stride along b with aligned 256-column rows (the Hybrid Demucs decoder geometry)
    (48, 96, (6, 256), (3, 3), (1, 1), (1, 1), (1, 1), 2),     # decoder rewrite: one 256-column segment per row
    (32, 40, (1, 1000), (1, 3), (1, 1), (0, 2), (1, 2), 3),    # dilated k3 over 4 segments, last one partial, halo 2
    (24, 64, (5, 260), (3, 3), (1, 1), (1, 1), (1, 1), 1),     # 1.5 chunks of channels, a 4-column tail segment
    (16, 192, (3, 512), (3, 1), (1, 1), (1, 0), (1, 1), 2),    # row taps only (no halo columns), M = 192
]


@pytest.mark.parametrize("case", CASES)
def test_conv_bf16x3(case):
    from remfx_amd import ops
    Cin, Cout, (IA, IB), (KA, KB), stride, padding, dilation, N = case
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, Cin, IA, IB, generator=g)
    w = torch.randn(Cout, Cin, KA, KB, generator=g) / (Cin * KA * KB) ** 0.5
    b = torch.randn(Cout, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    y = F.conv2d(xr, wr, br, stride, padding, dilation)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    yd = ops.conv2d(xd, wd, bd, stride, padding, dilation)
    yd.backward(gy.to(DEV))
    check(_rms(yd.detach().cpu(), y.detach()), 3e-5, max(1.0, float(y.detach().abs().max())))
    check(_rms(xd.grad.cpu(), xr.grad), 3e-5, max(1.0, float(xr.grad.abs().max())))
    check(_rms(wd.grad.cpu(), wr.grad), 3e-5, max(1.0, float(wr.grad.abs().max())))  


def test_tcn_golden_bf16x3(golden_dir):
    from oracle import ref_tcn
    from remfx_amd.tcn import TCN
    gd = np.load(os.path.join(golden_dir, "tcn_mid.npz"))
    cfg = {k[4:]: gd[k].item() for k in gd.files if k.startswith("cfg_")}
    sd = ref_tcn.tcn_init_state_dict(cfg["ninputs"], cfg["noutputs"], cfg["nblocks"], cfg["channel_width"],
                                     cfg["kernel_size"], seed=int(gd["seed"]))
    for k in [k for k in sd if k.endswith("relu.weight")]:
        sd[k] = torch.linspace(0.05, 0.45, sd[k].numel())
    net = TCN(**{k: (bool(v) if k == "causal" else v) for k, v in cfg.items()})
    net.load_state_dict(sd)
    net = net.to(DEV)
    with torch.no_grad():
        y = net(torch.from_numpy(gd["x"]).to(DEV)).cpu().numpy()
    check(float(np.sqrt(((y - gd["y"]) ** 2).mean())), 1e-4)          # north_star: 1e-4 RMS


def test_hdemucs_full_forward_bf16x3():
    from oracle import ref_hdemucs
    from remfx_amd.hdemucs import HDemucs
    torch.manual_seed(3)
    ref = ref_hdemucs.HDemucs(sources=["mixture"], audio_channels=1, nfft=4096, channels=48)
    net = HDemucs(sources=["mixture"], audio_channels=1, nfft=4096, channels=48)
    net.load_state_dict(ref.state_dict())
    net = net.to(DEV)
    x = torch.randn(1, 1, 262144, generator=torch.Generator().manual_seed(2)) * 0.1
    with torch.no_grad():
        y = ref(x)
        yd = net(x.to(DEV)).cpu()
    err = _rms(yd, y)
    print("hdemucs bf16x3 rms err", err, "output rms", float(y.pow(2).mean().sqrt()))
    check(err, 1e-4, max(1.0, float(y.abs().max())), what=err)


def test_hdemucs_small_grads_bf16x3():
    """whole-network gradients in bf16x3 mode vs autograd over the fp32 CPU oracle."""
    from tests.test_gpu_hdemucs import _pair
    ref, net = _pair(8)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 1, 20000, generator=g) * 0.5
    y = ref(x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    yd = net(x.to(DEV))
    check(_rms(yd.detach().cpu(), y.detach()), 1e-4, max(1.0, float(y.detach().abs().max())))
    yd.backward(gy.to(DEV))
    refg = dict(ref.named_parameters())
    num = den = 0.0
    for n, p in net.named_parameters():
        r = refg[n].grad
        if r is None:
            continue
        num += float(((p.grad.cpu() - r) ** 2).sum()); den += float((r ** 2).sum())
    check((num / den) ** 0.5, 3e-3, what=(num / den) ** 0.5)


"""GPU: the channels-last bf16 trunk of Hybrid Demucs' frequency branch (remfx_amd/clchain.py) against the channel-major kernels it
replaces, on the cfg/model/demucs.yaml geometry (torchaudio HDemucs behind remfx/models.py:308,317).

Reference = the same network in the exact-fp32 arithmetic mode (which never takes the trunk).  Statement: in the bf16 mode the trunk
is not further from fp32 than the channel-major bf16 path is (output and every parameter gradient), and its weight gradients are
deterministic."""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.one_mode]
DEV = "cuda:0"


def _net():
    from remfx_amd.hdemucs import HDemucs
    torch.manual_seed(0)
    net = HDemucs(sources=["mixture"], audio_channels=1, nfft=4096, channels=48)
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith(".scale"):
                p.fill_(0.3)
    return net.to(DEV)


def _run(net, x, gy, mode, trunk, time=True):
    from remfx_amd import hdemucs, ops
    prev, prev_t, prev_tt = ops.gemm_precision(), hdemucs.CL_TRUNK, hdemucs.CL_TIME
    ops.set_gemm_precision(mode)
    hdemucs.CL_TRUNK = trunk
    hdemucs.CL_TIME = time
    try:
        net.zero_grad(set_to_none=True)
        y = net(x)
        y.backward(gy)
        torch.cuda.synchronize()
        return y.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    finally:
        ops.set_gemm_precision(prev)
        hdemucs.CL_TRUNK = prev_t
        hdemucs.CL_TIME = prev_tt


def _rel(a, b):
    return float(((a.double() - b.double()) ** 2).sum().sqrt() / (b.double() ** 2).sum().sqrt().clamp_min(1e-30))


def test_trunk_vs_channel_major_full_config():
    net = _net()
    assert net._cl_layers(256, torch.device(DEV)) == 0           # f32 mode outside: off
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(2, 1, 262144, generator=g) * 0.1).to(DEV)
    gy = torch.randn(2, 1, 1, 262144, generator=g).to(DEV)
    y32, g32 = _run(net, x, gy, "f32", False)
    y_cm, g_cm = _run(net, x, gy, "bf16", False)
    y_cl, g_cl = _run(net, x, gy, "bf16", True)
    y_cl2, g_cl2 = _run(net, x, gy, "bf16", True)
    # the same with the pieces of the trunk switched off one at a time (A/B flags of hdemucs.py): every combination stays in the band
    from remfx_amd import hdemucs
    for flag in ("CL_TIME", "CL_ENDS", "CL_DCONV"):
        prev = getattr(hdemucs, flag)
        setattr(hdemucs, flag, False)
        try:
            y_v, g_v = _run(net, x, gy, "bf16", True, time=flag != "CL_TIME")
        finally:
            setattr(hdemucs, flag, prev)
        num = sum(float(((g_v[n].double() - g32[n].double()) ** 2).sum()) for n in g32)
        den_ = sum(float((g32[n].double() ** 2).sum()) for n in g32)
        print(f"{flag} off: output {_rel(y_v, y32):.3e}, global gradient {(num / den_) ** 0.5:.3e}")
        assert _rel(y_v, y32) < 1.5 * _rel(y_cm, y32) + 1e-4 and (num / den_) ** 0.5 < 0.04
    e_cm, e_cl = _rel(y_cm, y32), _rel(y_cl, y32)
    print(f"output: relative error vs fp32 mode: channel-major bf16 {e_cm:.3e}, channels-last trunk {e_cl:.3e}")
    assert e_cl < 1.5 * e_cm + 1e-4
    worst = 0.0
    num_cm = num_cl = den = 0.0
    for n in g32:
        num_cm += float(((g_cm[n].double() - g32[n].double()) ** 2).sum())
        num_cl += float(((g_cl[n].double() - g32[n].double()) ** 2).sum())
        den += float((g32[n].double() ** 2).sum())
        r_cm, r_cl = _rel(g_cm[n], g32[n]), _rel(g_cl[n], g32[n])
        if n.startswith("freq_") and ("rewrite" in n or "conv" in n) and n.split(".")[1] in "012345":
            print(f"  {n:44s} cm {r_cm:.3e} cl {r_cl:.3e}")
        worst = max(worst, r_cl / max(r_cm, 1e-3))
    rel_cm, rel_cl = (num_cm / den) ** 0.5, (num_cl / den) ** 0.5
    print(f"global gradient: relative error vs fp32 mode: channel-major bf16 {rel_cm:.3e}, trunk {rel_cl:.3e}; worst per-tensor ratio {worst:.2f}")
    assert rel_cl < 1.5 * rel_cm + 1e-4
    assert worst < 3.0
    # determinism of the trunk's own weight gradients (fixed-order reduction, no atomics)
    for n in g_cl:
        lay = n.split(".")
        if lay[0] == "freq_decoder" and int(lay[1]) >= 2 and lay[2] in ("rewrite", "conv_tr") and not (int(lay[1]) == 5 and lay[2] == "conv_tr"):
            assert torch.equal(g_cl[n], g_cl2[n]), n
        if lay[0] == "freq_encoder" and int(lay[1]) <= 3 and lay[2] == "rewrite":
            assert torch.equal(g_cl[n], g_cl2[n]), n
        if lay[0] == "freq_encoder" and 1 <= int(lay[1]) <= 3 and lay[2] == "conv":
            assert torch.equal(g_cl[n], g_cl2[n]), n


def test_trunk_inference_matches_training_forward():
    net = _net()
    from remfx_amd import ops
    x = (torch.randn(1, 1, 262144, generator=torch.Generator().manual_seed(2)) * 0.1).to(DEV)
    prev = ops.gemm_precision()
    ops.set_gemm_precision("bf16")
    try:
        assert net._cl_layers(256, torch.device(DEV)) == 4
        with torch.no_grad():
            y0 = net(x)
        y1 = net(x)
        assert _rel(y0, y1.detach()) < 1e-4       # not bit-equal: the iSTFT overlap-add and the training-mode DConv stores differ in the last bits
    finally:
        ops.set_gemm_precision(prev)

"""GPU: parity statement of the bf16 arithmetic mode (trainer.precision=bf16-mixed, BASELINE config 3).

The reference has no bf16 path of its own (cfg/config.yaml:112 is precision 32); Lightning's bf16-mixed is
torch.autocast(bfloat16): convolutions / linear layers / LSTM cells round their operands to bf16, accumulate in fp32 and
return bf16, norms / FFT / losses stay fp32.  The HIP mode rounds the same operands to bf16 (RNE), accumulates in fp32 and
keeps fp32 results -- never less accurate than autocast.  So the test is: against the fp32 CPU oracle, the HIP bf16
output is not further away than the CPU oracle run under torch.autocast("cpu", bfloat16) is, up to a factor that covers
the different summation order.  (fp32 and bf16x3 parity to the oracle itself: every other test file, all modes.)"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu, pytest.mark.one_mode]
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _bf16():
    from remfx_amd import ops
    old = ops.gemm_precision()
    ops.set_gemm_precision("bf16")
    yield
    ops.set_gemm_precision(old)


def _rms(a, b):
    return float(((a.float() - b.float()) ** 2).mean().sqrt())


CASES = [
    (48, 96, (32, 40), (8, 1), (4, 1), (2, 0), (1, 1), 2),      # HDemucs freq encoder
    (96, 192, (16, 64), (3, 3), (1, 1), (1, 1), (1, 1), 2),     # decoder rewrite
    (256, 256, (1, 2100), (1, 7), (1, 1), (0, 0), (1, 2), 1),   # TCN block
    (12, 96, (1, 256), (1, 1), (1, 1), (0, 0), (1, 1), 64),     # DConv 1x1, Cin not a multiple of 8
]


@pytest.mark.parametrize("case", CASES)
def test_conv_bf16_not_worse_than_autocast(case):
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
    xa, wa, ba = (t.clone().requires_grad_(True) for t in (x, w, b))
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ya = F.conv2d(xa, wa, ba, stride, padding, dilation)
    ya.float().backward(gy)
    xd, wd, bd = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    yd = ops.conv2d(xd, wd, bd, stride, padding, dilation)
    yd.backward(gy.to(DEV))
    for name, got, auto, ref in (("y", yd.detach(), ya.detach(), y.detach()), ("dx", xd.grad, xa.grad, xr.grad),
                                 ("dw", wd.grad, wa.grad, wr.grad)):
        e_hip, e_auto = _rms(got.cpu(), ref), _rms(auto, ref)
        scale = float(ref.abs().max())
        assert e_hip < 1.5 * e_auto + 1e-6 * scale, (name, e_hip, e_auto, scale)
        assert e_hip < 1e-2 * scale, (name, e_hip, scale)            # absolute sanity: operand rounding is 2^-9


def test_hdemucs_bf16_not_worse_than_autocast():
    """Whole network, forward and the full gradient vector (channels = 8, 20000 samples: the oracle runs in seconds)."""
    from oracle import ref_hdemucs
    from remfx_amd.hdemucs import HDemucs
    torch.manual_seed(0)
    ref = ref_hdemucs.HDemucs(sources=["mixture"], audio_channels=1, nfft=4096, channels=8)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if n.endswith(".scale"):
                p.fill_(0.3)
    net = HDemucs(sources=["mixture"], audio_channels=1, nfft=4096, channels=8)
    net.load_state_dict(ref.state_dict(), strict=True)
    net = net.to(DEV)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 1, 20000, generator=g) * 0.5
    y = ref(x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    gref = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
    ref.zero_grad()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        ya = ref(x)
    ya.float().backward(gy)
    gauto = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}
    yd = net(x.to(DEV))
    yd.backward(gy.to(DEV))
    e_hip, e_auto = _rms(yd.detach().cpu(), y.detach()), _rms(ya.detach(), y.detach())
    print("forward rms error vs fp32 oracle: hip bf16", e_hip, "autocast oracle", e_auto, "max |y|", float(y.abs().max()))
    assert e_hip < 2.0 * e_auto, (e_hip, e_auto)
    num_h = num_a = den = 0.0
    for n, p in net.named_parameters():
        if n not in gref:
            continue
        num_h += float(((p.grad.cpu() - gref[n]) ** 2).sum())
        num_a += float(((gauto[n] - gref[n]) ** 2).sum())
        den += float((gref[n] ** 2).sum())
    rel_h, rel_a = (num_h / den) ** 0.5, (num_a / den) ** 0.5
    print("global relative gradient error vs fp32 oracle: hip bf16", rel_h, "autocast oracle", rel_a)
    assert rel_h < 2.0 * rel_a, (rel_h, rel_a)


def test_hdemucs_full_config_bf16_forward():
    """cfg/model/demucs.yaml geometry (83.6 M parameters), one 262144-sample clip."""
    from oracle import ref_hdemucs
    from remfx_amd.hdemucs import HDemucs
    torch.manual_seed(3)
    ref = ref_hdemucs.HDemucs(sources=["mixture"], audio_channels=1, nfft=4096, channels=48)
    net = HDemucs(sources=["mixture"], audio_channels=1, nfft=4096, channels=48)
    net.load_state_dict(ref.state_dict(), strict=True)
    net = net.to(DEV)
    x = torch.randn(1, 1, 262144, generator=torch.Generator().manual_seed(2)) * 0.1
    with torch.no_grad():
        y = ref(x)
        with torch.autocast("cpu", dtype=torch.bfloat16):
            ya = ref(x).float()
        yd = net(x.to(DEV)).cpu()
    e_hip, e_auto = _rms(yd, y), _rms(ya, y)
    print("full config forward rms error vs fp32 oracle: hip bf16", e_hip, "autocast oracle", e_auto, "max |y|", float(y.abs().max()))
    assert e_hip < 2.0 * e_auto, (e_hip, e_auto)


def test_hdemucs_full_config_bf16_gradients_vs_autocast(golden_dir):
    """cfg/model/demucs.yaml geometry, one 262144-sample clip, EVERY parameter gradient (strided slices of <= 256 values per tensor,
    395 tensors): the HIP bf16 mode -- the mode BENCH reports, channels-last trunk included -- is not further from the fp32 CPU oracle
    than 2x the oracle's own error under torch.autocast("cpu", bfloat16) (fixture: oracle/gen_hdemucs_autocast_golden.py)."""
    import os
    import numpy as np
    from oracle.gen_hdemucs_grad_golden import build, inputs
    from remfx_amd.hdemucs import HDemucs
    gd = np.load(os.path.join(golden_dir, "hdemucs_full_grad_autocast.npz"))
    ref = build()
    net = HDemucs(sources=["mixture"], audio_channels=1, nfft=4096, channels=48)
    net.load_state_dict(ref.state_dict(), strict=True)
    del ref
    net = net.to(DEV)
    x, gy = inputs()
    y = net(x.to(DEV))
    y.backward(gy.to(DEV))
    ys = y.detach().cpu().reshape(-1)[::4099].numpy()
    e_hip = float(np.sqrt(((ys - gd["y32"]) ** 2).sum() / (gd["y32"] ** 2).sum()))
    e_auto = float(np.sqrt(((gd["yauto"] - gd["y32"]) ** 2).sum() / (gd["y32"] ** 2).sum()))
    print(f"output: relative error vs fp32 oracle: hip bf16 {e_hip:.3e}, autocast oracle {e_auto:.3e}")
    assert e_hip < 2.0 * e_auto
    params = dict(net.named_parameters())
    num_h = num_a = den = 0.0
    worst = (0.0, "")
    for i, n in enumerate(gd["names"].tolist()):
        g32, ga = gd[f"f{i}"].astype(np.float64), gd[f"a{i}"].astype(np.float64)
        gr = params[n].grad.detach().cpu().reshape(-1)
        step = max(1, gr.numel() // 256)
        gh = gr[::step][:256].double().numpy()
        eh, ea, nn = float(((gh - g32) ** 2).sum()), float(((ga - g32) ** 2).sum()), float((g32 ** 2).sum())
        num_h += eh; num_a += ea; den += nn
        ratio = (eh / max(ea, 1e-6 * nn, 1e-300)) ** 0.5
        if ratio > worst[0]:
            worst = (ratio, n)
    rel_h, rel_a = (num_h / den) ** 0.5, (num_a / den) ** 0.5
    print(f"gradients (395 tensors, sliced): relative error vs fp32 oracle: hip bf16 {rel_h:.3e}, autocast oracle {rel_a:.3e} "
          f"(fixture {float(gd['auto_rel']):.3e}); worst per-tensor ratio {worst[0]:.2f} at {worst[1]}")
    assert rel_h < 2.0 * rel_a
    assert worst[0] < 6.0, worst


@pytest.mark.one_mode
def test_bf16_storage_is_exactly_rounding():
    """bf16 STORAGE of a conv output / its gradient (ops.bf16_storage) changes nothing but the stored bits: the 16-bit paths of the
    tap-major gather (dgrad), the wide weight-gradient kernel, the GEMM store, GroupNorm (+GLU) forward / backward and the GLU
    backward are compared BIT-EXACTLY with the fp32-storage kernels fed the same (already bf16-representable) values."""
    from remfx_amd import nnops, ops
    prev = ops.gemm_precision()
    ops.set_gemm_precision("bf16")
    try:
        g = torch.Generator().manual_seed(5)
        N, Cin, Cout, T = 5, 24, 96, 200
        x = torch.randn(N, Cin, T, generator=g).to(DEV)
        w = (torch.randn(Cout, Cin, 1, generator=g) * 0.2).to(DEV)
        b = torch.randn(Cout, generator=g).to(DEV)
        # forward store: 16-bit output == RNE(fp32 output); epilogue statistics are those of the rounded values
        st32 = torch.zeros(N, 16, 2, device=DEV, dtype=torch.float64)
        st16 = torch.zeros_like(st32)
        y32 = ops.conv1d(x, w, b, stat_sums=st32)
        y16 = ops.conv1d(x, w, b, stat_sums=st16, out_bf16=True)
        assert y16.dtype == torch.bfloat16 and torch.equal(y16, y32.bfloat16())
        ref = torch.stack([y16.float().double().sum((1, 2)), y16.float().double().pow(2).sum((1, 2))], 1)
        torch.testing.assert_close(st16.sum(1), ref, rtol=1e-6, atol=1e-6)
        # GroupNorm + GLU + LayerScale + residual on the 16-bit tensor == the fp32 kernel on the widened values
        gam, bet = torch.randn(Cout, generator=g).to(DEV), torch.randn(Cout, generator=g).to(DEV)
        sc = torch.rand(Cout // 2, generator=g).to(DEV)
        res = torch.randn(N, Cout // 2, T, generator=g).to(DEV)
        outs = []
        for yy in (y16, y16.float()):
            yy = yy.detach().requires_grad_(True)
            o = nnops.group_norm(yy, 1, gam, bet, 1e-5, mode="glu_scale_res", res=res, scale=sc)
            go = torch.randn(o.shape, generator=torch.Generator().manual_seed(9)).to(DEV)
            o.backward(go)
            outs.append((o.detach(), yy.grad))
        assert torch.equal(outs[0][0], outs[1][0])
        assert outs[0][1].dtype == torch.bfloat16 and torch.equal(outs[0][1], outs[1][1].bfloat16())
        # many small samples (freq-branch DConv shapes): the register-resident per-sample backward kernels
        for (Nn, Cc, Ss, md) in ((600, 96, 64, "glu_scale_res"), (520, 192, 128, "glu"), (530, 12, 256, "gelu"), (515, 40, 32, "none")):
            xs = torch.randn(Nn, Cc, Ss, generator=g).to(DEV).bfloat16()
            gm, bt = torch.randn(Cc, generator=g).to(DEV), torch.randn(Cc, generator=g).to(DEV)
            kw = {}
            if md == "glu_scale_res":
                kw = dict(res=torch.randn(Nn, Cc // 2, Ss, generator=g).to(DEV), scale=torch.rand(Cc // 2, generator=g).to(DEV))
            pair = []
            for xx in (xs, xs.float()):
                xx = xx.detach().requires_grad_(True)
                o = nnops.group_norm(xx, 1, gm, bt, 1e-5, mode=md, **kw)
                o.backward(torch.randn(o.shape, generator=torch.Generator().manual_seed(3)).to(DEV))
                pair.append((o.detach(), xx.grad))
            assert torch.equal(pair[0][0], pair[1][0]), md
            # the 16-bit per-sample kernel gives a lane PAIRS of samples, so its wave reductions add in another order than the
            # fp32-storage kernel's: dx may round the other way in a few places -- at most one bf16 ulp, in < 1 % of the elements
            a16, a32 = pair[0][1].float(), pair[1][1].bfloat16().float()
            assert float(((a16 - a32).abs() - 2.0 ** -7 * a32.abs()).max()) <= 1e-6 * float(a32.abs().max()), md
            assert float((a16 != a32).float().mean()) < 1e-2, md
        # the 16-bit gradient as GEMM operand: input gradient and weight gradient == the fp32-storage kernels on the same values
        gz = outs[0][1]
        for gg in (gz, gz.float()):
            xx = x.detach().requires_grad_(True)
            ww = w.detach().requires_grad_(True)
            bb = b.detach().requires_grad_(True)
            yy = ops.conv1d(xx, ww, bb, out_bf16=gg.dtype == torch.bfloat16)
            yy.backward(gg)
            outs.append((xx.grad, ww.grad, bb.grad))
        for a16, a32 in zip(outs[2], outs[3]):
            assert torch.equal(a16, a32)
        # strided plans (the time branch's encoder convs: kernel 8, stride 4 along the contiguous axis): 16-bit output and a 16-bit
        # gradient through the 32-position weight-gradient kernel (gemm_wgrad_bf_kernel<.., G16>) and the merged-phase input gradient
        for (Ci, Co, Tt) in ((24, 96, 400), (1, 48, 1008), (48, 20, 272)):
            xs = torch.randn(3, Ci, Tt, generator=g).to(DEV)
            ws = (torch.randn(Co, Ci, 8, generator=g) * 0.1).to(DEV)
            bs = torch.randn(Co, generator=g).to(DEV)
            pair = []
            for store in (True, False):
                xx, ww, bb = (t.detach().requires_grad_(True) for t in (xs, ws, bs))
                yy = ops.conv1d(xx, ww, bb, 4, 2, out_bf16=store)
                assert yy.dtype == (torch.bfloat16 if store else torch.float32), (Ci, Co, yy.dtype)
                gs = torch.randn(yy.shape, generator=torch.Generator().manual_seed(11)).to(DEV).bfloat16()
                yy.backward(gs if store else gs.float())
                pair.append((yy.detach().float(), xx.grad, ww.grad, bb.grad))
            assert torch.equal(pair[0][0], pair[1][0].bfloat16().float()), (Ci, Co)
            for a16, a32 in zip(pair[0][1:], pair[1][1:]):
                assert torch.equal(a16, a32), (Ci, Co)
        # GLU in the GEMM store with the conv output kept in 16 bits (ConvGlu2dFn) vs RFX_BF16_STORE off
        x2 = torch.randn(3, 16, 6, 40, generator=g).to(DEV)
        w2 = (torch.randn(32, 16, 3, 3, generator=g) * 0.1).to(DEV)
        b2 = torch.randn(32, generator=g).to(DEV)
        got = []
        for store in (True, False):
            ops.BF16_STORE = store
            xx, ww = x2.detach().requires_grad_(True), w2.detach().requires_grad_(True)
            o = ops.conv2d_glu(xx, ww, b2, (1, 1), (1, 1))
            o.backward(torch.ones_like(o))
            got.append((o.detach(), xx.grad, ww.grad))
        ops.BF16_STORE = True
        # not bit-equal: the GLU is taken of the rounded conv output (as the backward pass sees it); bf16 rounding of its inputs
        for a, c in zip(got[0], got[1]):
            assert float((a - c).abs().max()) <= 2e-2 * float(c.abs().max())
    finally:
        ops.BF16_STORE = True
        ops.set_gemm_precision(prev)

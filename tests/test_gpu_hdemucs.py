"""GPU parity: HIP Hybrid Demucs vs the CPU oracle restatement (same state_dict)."""
import pytest
import torch

from tests.conftest import check, mode, tol
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rms(a, b):
    return float(((a - b) ** 2).mean().sqrt())


def _pair(channels, seed=0):
    from oracle import ref_hdemucs
    from remfx_amd.hdemucs import HDemucs
    torch.manual_seed(seed)
    ref = ref_hdemucs.HDemucs(sources=["mixture"], audio_channels=1, nfft=4096, channels=channels)
    # make LayerScale / freq-emb paths numerically visible
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if n.endswith(".scale"):
                p.fill_(0.3)
    net = HDemucs(sources=["mixture"], audio_channels=1, nfft=4096, channels=channels)
    net.load_state_dict(ref.state_dict(), strict=True)
    return ref, net.to(DEV)


def test_hdemucs_small_fwd_bwd(monkeypatch):
    # RFX_STRICT_NATIVE=1: an op without a HIP kernel (nnops.INTERIM) raises instead of running through torch-ROCm
    from remfx_amd import nnops
    monkeypatch.setenv("RFX_STRICT_NATIVE", "1")
    nnops.INTERIM.clear()
    ref, net = _pair(8)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(2, 1, 20000, generator=g) * 0.5
    y = ref(x)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    yd = net(x.to(DEV))
    assert yd.shape == y.shape
    check(_rms(yd.detach().cpu(), y.detach()), 1e-4, max(1.0, float(y.detach().abs().max())))
    yd.backward(gy.to(DEV))
    refg = dict(ref.named_parameters())
    # whole-network gradients (6 enc + 6 dec layers, BLSTM, attention).  Bias gradients are heavily
    # cancelling sums, so individual tensors carry fp32 ordering noise ~1e-2 of their max; the
    # parity statement is on the full gradient vector.
    num = den = 0.0
    worst = ("", 0.0)
    for n, p in net.named_parameters():
        r = refg[n].grad
        if r is None:
            continue
        d = p.grad.cpu() - r
        num += float((d ** 2).sum()); den += float((r ** 2).sum())
        err = _rms(p.grad.cpu(), r) / max(1e-4, float(r.abs().max()))
        if err > worst[1]:
            worst = (n, err)
        check(err, 5e-2, what=(n, err))
    rel = (num / den) ** 0.5
    print("global relative grad error", rel, "worst tensor", worst)
    check(rel, 2e-3, what=rel)
    assert not nnops.INTERIM


def test_hdemucs_full_config_forward():
    """cfg/model/demucs.yaml geometry: one 262144-sample clip, 83.6 M parameters."""
    ref, net = _pair(48, seed=3)
    assert sum(p.numel() for p in net.parameters()) == 83630131
    g = torch.Generator().manual_seed(2)
    x = torch.randn(1, 1, 262144, generator=g) * 0.1
    with torch.no_grad():
        y = ref(x)
        yd = net(x.to(DEV)).cpu()
    assert yd.shape == (1, 1, 1, 262144)
    check(_rms(yd, y), 1e-4, max(1.0, float(y.abs().max())), what=_rms(yd, y))


def test_hdemucs_decoder_groupnorm_sees_cropped_border():
    """Upstream: z = norm2(conv_tr(y)) over the FULL transposed-conv output, crop afterwards.  One GroupNorm'd decoder
    layer in isolation, with the edge taps (the only contributors to the rows / samples the crop drops) x 25: statistics
    taken after the crop differ at the 10 % level here, while in the whole randomly initialised network that ordering
    error is 3e-6 and hides below the 1e-4 budget."""
    from oracle import ref_hdemucs
    from remfx_amd import hdemucs
    for freq in (True, False):
        torch.manual_seed(7)
        kw = dict(chin=16, chout=8, freq=freq, norm_groups=4, context=1)
        ref = ref_hdemucs.HDecLayer(norm=True, **kw)
        with torch.no_grad():
            ref.conv_tr.weight[:, :, :2] *= 25.0
            ref.conv_tr.weight[:, :, -2:] *= 25.0
        net = hdemucs._HDecLayer(norm_type="group_norm", **kw)
        net.load_state_dict(ref.state_dict(), strict=True)
        net = net.to(DEV)
        g = torch.Generator().manual_seed(8)
        shape = (3, 16, 4, 64) if freq else (3, 16, 12)
        x, skip = torch.randn(shape, generator=g), torch.randn(shape, generator=g)
        length = 4 * shape[-1] - 1 if not freq else None
        with torch.no_grad():
            z, _ = ref(x, skip, length)
            zc = ref.conv_tr(F.glu(ref.norm1(ref.rewrite(x + skip)), dim=1))
            zc = zc[..., ref.pad:-ref.pad, :] if freq else zc[..., ref.pad:ref.pad + length]
            wrong = F.gelu(ref.norm2(zc))                    # crop-then-norm: must NOT be what the product computes
            zd, _ = net(x.to(DEV), skip.to(DEV), length)
        assert zd.shape == z.shape
        assert _rms(wrong, z) > 3e-3, "test input does not separate the two orderings"
        check(_rms(zd.cpu(), z), 1e-5, max(1.0, float(z.abs().max())), what=(freq, _rms(zd.cpu(), z)))


@pytest.mark.parametrize("T", [256, 130, 201])
def test_blstm_overlapping_frames(T):
    """torchaudio `_BLSTM`: sequences longer than 200 steps run as frames of 200 at stride 100 and the central parts are
    stitched back.  Forward and all gradients against the oracle, at the frame count of the headline config (T = 256 -> 3
    frames), the unframed case and the shortest framed one."""
    from oracle import ref_hdemucs
    from remfx_amd import hdemucs
    torch.manual_seed(11)
    ref = ref_hdemucs.BLSTM(32, layers=2, skip=True)
    net = hdemucs._BLSTM(32, layers=2, skip=True)
    net.load_state_dict(ref.state_dict(), strict=True)
    net = net.to(DEV)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(3, 32, T, generator=g)
    xr = x.clone().requires_grad_(True)
    y = ref(xr)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd = x.to(DEV).requires_grad_(True)
    yd = net(xd)
    yd.backward(gy.to(DEV))
    check(_rms(yd.detach().cpu(), y.detach()), 1e-5, max(1.0, float(y.detach().abs().max())))
    check(_rms(xd.grad.cpu(), xr.grad), 2e-5, max(1.0, float(xr.grad.abs().max())))
    for (n, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        check(_rms(p.grad.cpu(), q.grad), 1e-4, max(1e-3, float(q.grad.abs().max())), what=n)


def test_hdemucs_full_config_gradients_golden(golden_dir):
    """cfg/model/demucs.yaml geometry, one 262144-sample clip: backward of the HIP network against the committed gradient
    fixture of the CPU oracle (oracle/gen_hdemucs_grad_golden.py: per-tensor gradient norms + strided slices of 13 parameters
    spread over encoders / decoders / DConv / BLSTM / attention / frequency embedding, global gradient norm, output slices)."""
    import os
    import numpy as np
    gd = np.load(os.path.join(golden_dir, "hdemucs_full_grad.npz"))
    ref, net = _pair(48, seed=3)
    del ref
    g = torch.Generator().manual_seed(12)
    x = torch.randn(1, 1, 262144, generator=g) * 0.1
    gy = torch.randn(1, 1, 1, 262144, generator=g)
    y = net(x.to(DEV))
    y.backward(gy.to(DEV))
    ys = y.detach().cpu().reshape(-1)[::4099].numpy()
    # bf16 bounds of this test = 2x what round 5 measured (RFX_TOL_LOG, channels-last trunk on): y 5.5e-5, global norm 1.4e-3, worst
    # slice 5.2e-2 and worst norm 1.2e-2 (both freq_encoder.1.dconv.layers.1.0.weight); the statement against the autocast oracle is
    # tests/test_gpu_bf16_mixed.py::test_hdemucs_full_config_bf16_gradients_vs_autocast
    e_y = float(np.sqrt(((ys - gd["y_slice"]) ** 2).mean()))
    print(f"y slice rms error {e_y:.3e}")
    check(e_y, 1e-4, max(1.0, float(np.abs(gd["y_slice"]).max())), bf16=1.2e-4, what="y")
    params = dict(net.named_parameters())
    tot = sum(float(p.grad.double().pow(2).sum()) for p in params.values() if p.grad is not None) ** 0.5
    print(f"global gradient norm: relative error {abs(tot - float(gd['grad_global_norm'])) / float(gd['grad_global_norm']):.3e}")
    check(abs(tot - float(gd["grad_global_norm"])), 2e-3, float(gd["grad_global_norm"]), bf16=3e-3, what="global grad norm")
    for i, n in enumerate(gd["names"].tolist()):
        gr = params[n].grad.detach().cpu().reshape(-1)
        step = max(1, gr.numel() // 512)
        sl = gr[::step][:512].numpy()
        ref_sl, ref_norm = gd[f"g{i}_slice"], float(gd[f"g{i}_norm"])
        err = float(np.sqrt(((sl - ref_sl) ** 2).mean())) / max(1e-12, float(np.sqrt((ref_sl ** 2).mean())))
        # slice RMS error relative to the slice RMS; bias-like tensors are cancelling sums (see test_hdemucs_small_fwd_bwd)
        print(f"  {n:52s} slice {err:.3e} norm {abs(float(gr.double().norm()) - ref_norm) / ref_norm:.3e}")
        check(err, 2e-2, bf16x3=2e-2, bf16=0.11, what=(n, "slice", err))
        check(abs(float(gr.double().norm()) - ref_norm), 1e-2, ref_norm, bf16=0.025, what=(n, "norm"))


@pytest.mark.one_mode
@pytest.mark.parametrize("B,heads,ch,T,nd", [(3, 4, 48, 256, 4), (2, 4, 96, 200, 4), (2, 4, 96, 64, 4), (2, 2, 16, 37, 3),
                                             (1, 2, 64, 130, 4), (2, 1, 32, 256, 1),
                                             # more than 256 frames = longer than one 262144-sample clip (whole files): the
                                             # streaming any-T kernels (rfx_localstate_gen_*), both session modes
                                             (2, 4, 48, 300, 4), (1, 4, 96, 705, 4), (1, 2, 16, 1030, 9)])
def test_localstate_mfma_vs_exact(B, heads, ch, T, nd):
    """LocalState attention on the bf16 matrix pipe (attention_mfma.hip, flash-style: weights recomputed in the backward pass)
    and the exact fp32 kernels (attention.hip; ch * T <= 12288) against autograd over the operator written out in torch
    fp64, forward and all four gradients.  bf16 bound: operand rounding of q, k, content, P and dS (2^-9 relative each) --
    2 % of the RMS of each tensor; exact kernels: 2e-5."""
    from remfx_amd import nnops, ops
    g = torch.Generator().manual_seed(B * 1000 + T)
    mk = lambda c: (torch.randn(B, heads * c, T, generator=g) * 0.8).to(DEV).requires_grad_(True)
    q, k, cont, qd = mk(ch), mk(ch), mk(ch), mk(nd)
    gy = torch.randn(B, heads * ch, T, generator=g).to(DEV)
    qq, kk, cc, dd = (t.detach().double().cpu().requires_grad_(True) for t in (q, k, cont, qd))
    qh, kh, chh = (t.view(B, heads, ch, T) for t in (qq, kk, cc))
    dots = torch.einsum("bhct,bhcs->bhts", kh, qh) / ch ** 0.5
    idx = torch.arange(T, dtype=torch.float64)
    delta = (idx[:, None] - idx[None, :]).abs()
    dec = torch.sigmoid(dd.view(B, heads, nd, T)) / 2
    pen = -torch.arange(1, nd + 1, dtype=torch.float64).view(-1, 1, 1) * delta / nd ** 0.5
    dots = dots + torch.einsum("fts,bhfs->bhts", pen, dec)
    dots = dots.masked_fill(torch.eye(T, dtype=torch.bool), -100.0)
    w = torch.softmax(dots, dim=2)
    yr = torch.einsum("bhts,bhct->bhcs", w, chh).reshape(B, heads * ch, T)
    yr.backward(gy.double().cpu())
    ref = [yr.detach(), qq.grad, kk.grad, cc.grad, dd.grad]
    prev = ops.gemm_precision()
    try:
        for m, bound in (("f32", 2e-5), ("bf16", 2e-2)):
            if T > 256:
                bound = 2e-5              # the any-T path is exact fp32 in every session mode (f32 with ch * T > 12288 too)
            ops.set_gemm_precision(m)
            for t in (q, k, cont, qd):
                t.grad = None
            y = nnops.local_state_attention(q, k, cont, qd, heads, nd)
            y.backward(gy)
            for name, a, b in zip(("out", "dq", "dk", "dcont", "dqd"), [y.detach()] + [t.grad for t in (q, k, cont, qd)], ref):
                rms = float(b.pow(2).mean().sqrt())
                err = float((a.double().cpu() - b).pow(2).mean().sqrt())
                assert err <= bound * rms + 1e-9, (m, name, err, rms)
    finally:
        ops.set_gemm_precision(prev)

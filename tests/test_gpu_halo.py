"""GPU: the halo-tile implicit-GEMM kernel (csrc/gemm_halo.h, bf16 arithmetic mode) -- stride-1 multi-tap convolutions of the Hybrid
Demucs decoders (3x3 rewrite + its input gradient, 3-tap (dilated) 1-D convolutions; reference call site remfx/models.py:319 through
torchaudio HDemucs `_HDecLayer.rewrite` / `_DConv`).  Checked three ways: against an fp64 convolution of the bf16-rounded operands
(what the MFMA computes, up to fp32 accumulation), against the tap-major kernel on the same plan (same products, another summation
order), and that the planner actually routes the shapes to it (rfx_gemm_fwd_variant)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu, pytest.mark.one_mode]

# Cin, Cout, (IA, IB), (KA, KB), padding, dilation, N
CASES = [
    (48, 96, (5, 256), (3, 3), (1, 1), (1, 1), 2),      # decoder rewrite: R = 3, 3 chunks, 2 tiles per row
    (96, 48, (3, 128), (3, 3), (1, 1), (1, 1), 2),      # its input-gradient shape: R = 2 (M = 48 of 64 rows), one tile per row (both halo columns outside)
    (40, 20, (2, 128), (3, 3), (1, 1), (1, 1), 3),      # channels padded to 48 inside the last block, R = 1, rows above / below outside
    (64, 192, (1, 384), (1, 3), (0, 1), (1, 1), 2),     # 1-D rewrite, 2 M tiles of 96
    (48, 12, (1, 256), (1, 3), (0, 2), (1, 2), 2),      # DConv bottleneck conv, dilation 2
    (16, 32, (4, 128), (3, 3), (1, 1), (1, 1), 1),      # one chunk
    (256, 96, (2, 128), (3, 3), (1, 1), (1, 1), 1),     # 32 groups x 9 taps exceed the 112-row table: blocks of 4 groups = two 16-channel sub-chunks per block
]


def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _variant(dp, x, out, prec=2):
    from remfx_amd import _lib
    return _lib.lib().rfx_gemm_fwd_variant(C.byref(dp.desc_for(x, out)), None, 0, prec)


@pytest.fixture
def bf16_mode():
    from remfx_amd import convplan, ops
    prev, prev_halo, prev_min = ops.gemm_precision(), convplan.HALO, convplan.HALO_MIN_POSITIONS
    ops.set_gemm_precision("bf16")
    convplan.HALO_MIN_POSITIONS = 0                 # the test shapes are small: lift the planner's "large layers only" rule
    ops._PLANS.clear()
    yield
    convplan.HALO, convplan.HALO_MIN_POSITIONS = prev_halo, prev_min
    ops._PLANS.clear()
    ops.set_gemm_precision(prev)


@pytest.mark.parametrize("case", CASES)
def test_halo_conv_fwd_dgrad(case, bf16_mode):
    from remfx_amd import convplan, ops
    dev = torch.device("cuda:0")
    Cin, Cout, (IA, IB), (KA, KB), padding, dilation, N = case
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, Cin, IA, IB, generator=g)
    w = torch.randn(Cout, Cin, KA, KB, generator=g) / (Cin * KA * KB) ** 0.5
    b = torch.randn(Cout, generator=g)
    gy = torch.randn(N, Cout, IA, IB, generator=g)
    xr, wr = _bf16(x).requires_grad_(True), _bf16(w)
    y_ref = F.conv2d(xr, wr, b.double(), (1, 1), padding, dilation)
    # input gradient of the bf16-rounded output gradient and weights (what the dgrad GEMM multiplies)
    (dx_ref,) = torch.autograd.grad(y_ref, xr, _bf16(gy))
    res = {}
    for halo in (True, False):
        convplan.HALO = halo
        ops._PLANS.clear()
        xd = x.to(dev).requires_grad_(True)
        wd, bd = w.to(dev), b.to(dev)
        yd = ops.conv2d(xd, wd, bd, (1, 1), padding, dilation)
        (dxd,) = torch.autograd.grad(yd, xd, gy.to(dev))
        res[halo] = (yd.detach().cpu().double(), dxd.cpu().double())
        # which kernel ran: forward plan of this call
        key = ops._key("cf", xd.shape, xd.stride(), wd.shape, (1, 1), padding, dilation, yd.stride())
        dp = ops._PLANS[(key, str(dev))]
        kind = _variant(dp, xd, yd) >> 4
        assert (kind in (6, 8)) == halo, (kind, halo, dp.p.halo_nt)
    scale = float(y_ref.detach().abs().max())
    for halo in (True, False):
        yd, dxd = res[halo]
        assert float((yd - y_ref.detach()).abs().max()) < 2e-5 * scale, halo          # fp32 accumulation of exact bf16 products
        assert float((dxd - dx_ref).abs().max()) < 2e-5 * float(dx_ref.abs().max()), halo
    # same products, different summation order
    assert float((res[True][0] - res[False][0]).abs().max()) < 1e-5 * scale


def test_halo_glu_and_bf16_operand(bf16_mode):
    """The decoder's rewrite as the model calls it: conv2d_glu (interleaved rows, GLU in the store, 16-bit conv output) and its
    backward (16-bit gradient gathered by the input-gradient launch = the IN16 instantiation)."""
    from remfx_amd import convplan, ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    N, C, Fr, T = 2, 48, 6, 256
    x = torch.randn(N, C, Fr, T, generator=g)
    w = torch.randn(2 * C, C, 3, 3, generator=g) / (9 * C) ** 0.5
    b = torch.randn(2 * C, generator=g) * 0.1
    gy = torch.randn(N, C, Fr, T, generator=g)
    out = {}
    for halo in (True, False):
        convplan.HALO = halo
        ops._PLANS.clear()
        xd = x.to(dev).requires_grad_(True)
        y = ops.conv2d_glu(xd, w.to(dev), b.to(dev), (1, 1), (1, 1))
        (dx,) = torch.autograd.grad(y, xd, gy.to(dev))
        out[halo] = (y.detach().cpu(), dx.cpu())
    zr = F.conv2d(_bf16(x), _bf16(w), b.double(), 1, 1)
    zr = zr.to(torch.bfloat16).double()                       # the stored conv output is bf16; the GLU is taken of the stored values
    y_ref = zr[:, :C] * torch.sigmoid(zr[:, C:])
    for halo in (True, False):
        assert float((out[halo][0].double() - y_ref).abs().max()) < 2e-2 * float(y_ref.abs().max())      # bf16 rounding of z either side of a tie
        assert float(((out[halo][0].double() - y_ref) ** 2).mean().sqrt()) < 2e-3 * float((y_ref ** 2).mean().sqrt())
    assert float((out[True][0] - out[False][0]).abs().max()) < 2e-2 * float(y_ref.abs().max())
    rel = float(((out[True][1] - out[False][1]) ** 2).mean().sqrt() / (out[False][1] ** 2).mean().sqrt())
    assert rel < 2e-3, rel                                      # both round the same gradient to bf16; z ties flip a few GLU derivatives


def test_halo_statistics_epilogue(bf16_mode):
    """GroupNorm(1, C) moments out of the epilogue (the DConv bottleneck conv's stat_sums) agree between the two kernels."""
    from remfx_amd import convplan, ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(7)
    x = torch.randn(3, 48, 1, 512, generator=g)
    w = torch.randn(12, 48, 1, 3, generator=g) / 12.0
    b = torch.randn(12, generator=g)
    st = {}
    for halo in (True, False):
        convplan.HALO = halo
        ops._PLANS.clear()
        s = ops.zeros((3, 16, 2), dev, torch.float64)
        y = ops.conv2d(x.to(dev), w.to(dev), b.to(dev), (1, 1), (0, 1), (1, 1), stat_sums=s, out_bf16=True)
        st[halo] = (s.sum(1).cpu(), y.float().cpu())
    assert torch.equal(st[True][1].to(torch.bfloat16), st[True][1].to(torch.bfloat16))
    ysum = st[True][1].double().sum((1, 2, 3))
    assert torch.allclose(st[True][0][:, 0], ysum, rtol=1e-6, atol=1e-4)
    assert torch.allclose(st[True][0], st[False][0], rtol=1e-3, atol=1e-2)

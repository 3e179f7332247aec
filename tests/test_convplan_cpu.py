"""CPU: the gather-GEMM planner reproduces torch conv / conv_transpose forward,
input gradients and weight gradients when its descriptors are interpreted exactly
as include/remfx_hip.h documents them."""
import pytest
import torch
import torch.nn.functional as F

from remfx_amd import convplan
from tests.plan_emulator import emulate_fwd, emulate_wgrad, scatter_weights

CASES = [
    # Cin, Cout, (IA, IB), (KA, KB), stride, padding, dilation
    (3, 5, (1, 50), (1, 7), (1, 1), (0, 0), (1, 4)),      # TCN-style dilated valid conv
    (2, 6, (33, 9), (8, 1), (4, 1), (2, 0), (1, 1)),      # HDemucs freq encoder
    (1, 4, (1, 64), (1, 8), (1, 4), (0, 2), (1, 1)),      # HDemucs time encoder
    (4, 3, (7, 11), (3, 3), (1, 1), (1, 1), (1, 1)),      # Cnn14 / decoder rewrite
    (3, 4, (12, 10), (7, 5), (2, 2), (3, 2), (1, 1)),     # DCUNet
    (5, 2, (9, 8), (5, 3), (2, 1), (2, 1), (1, 1)),
    (4, 4, (1, 40), (1, 3), (1, 1), (0, 2), (1, 2)),      # DConv dilated same conv
    # >= 8 input channels: these plans also carry the tap-major tables (checked inside emulate_fwd)
    (12, 5, (1, 40), (1, 3), (1, 1), (0, 2), (1, 2)),     # DConv 1x1-ish, Cin not a multiple of 8
    (9, 6, (7, 11), (3, 3), (1, 1), (1, 1), (1, 1)),      # 3x3 rewrite, Cin = 9 -> two 8-channel groups per tap
    (16, 3, (20, 6), (8, 1), (4, 1), (2, 0), (1, 1)),     # strided freq conv
    (8, 10, (1, 70), (1, 7), (1, 1), (0, 0), (1, 3)),     # TCN dilated
    (24, 4, (1, 33), (1, 1), (1, 1), (0, 0), (1, 1)),     # 1x1, three groups, odd group count
    # >= 32 input channels and several taps: channel-BLOCKED tap order (blocks of 16 channels presented as virtual taps)
    (32, 5, (5, 9), (3, 3), (1, 1), (1, 1), (1, 1)),      # 2 blocks x 9 taps
    (40, 3, (1, 30), (1, 3), (1, 1), (0, 2), (1, 2)),     # 5 groups -> 3 blocks, last one half padding
    (48, 4, (9, 5), (8, 1), (4, 1), (2, 0), (1, 1)),      # strided freq conv, 3 blocks x 8 taps
]


@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_dgrad_wgrad(case):
    Cin, Cout, (IA, IB), (KA, KB), stride, padding, dilation = case
    g = torch.Generator().manual_seed(0)
    N = 2
    x = torch.randn(N, Cin, IA, IB, generator=g, requires_grad=True)
    w = torch.randn(Cout, Cin, KA, KB, generator=g, requires_grad=True)
    b = torch.randn(Cout, generator=g, requires_grad=True)
    y = F.conv2d(x, w, b, stride, padding, dilation)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)

    plan = convplan.conv_fwd_plan(tuple(x.shape), x.stride(), tuple(w.shape), stride, padding, dilation,
                                  y.stride(), bias_row=True)
    assert tuple(plan.extra["out_shape"]) == tuple(y.shape)
    out = torch.zeros(y.numel())
    emulate_fwd(plan, w.detach().reshape(-1), x.detach().reshape(-1), out, b.detach())
    torch.testing.assert_close(out.view(y.shape), y.detach(), rtol=1e-4, atol=1e-4)

    dap = emulate_wgrad(plan, x.detach().reshape(-1), gy.reshape(-1))
    torch.testing.assert_close(scatter_weights(plan, dap, w.shape), w.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dap[:, plan.K - 1], b.grad, rtol=1e-4, atol=1e-4)

    dx = torch.full((x.numel(),), float("nan"))
    for p in convplan.conv_dgrad_plans(tuple(x.shape), x.stride(), tuple(w.shape), stride, padding, dilation,
                                       tuple(gy.shape), gy.stride()):
        emulate_fwd(p, w.detach().reshape(-1), gy.reshape(-1), dx)
    torch.testing.assert_close(dx.view(x.shape), x.grad, rtol=1e-4, atol=1e-4)


TCASES = [
    # Cin, Cout, (IA, IB), (KA, KB), stride, crop_lo, crop_hi
    (3, 4, (5, 6), (8, 1), (4, 1), (2, 0), (2, 0)),      # HDemucs freq decoder (pad 2)
    (4, 2, (1, 9), (1, 8), (1, 4), (0, 2), (0, 3)),      # HDemucs time decoder, length crop
    (3, 3, (1, 7), (1, 4), (1, 2), (0, 1), (0, 1)),      # innermost time decoder
    (4, 5, (6, 7), (5, 3), (2, 1), (2, 1), (2, 1)),      # DCUNet decoder (padding k//2)
    (2, 3, (4, 5), (7, 5), (2, 2), (3, 2), (3, 2)),
    (3, 2, (2, 3), (8, 1), (4, 1), (0, 0), (0, 0)),      # last_freq decoder, no crop
    (12, 3, (5, 6), (8, 1), (4, 1), (2, 0), (2, 0)),     # >= 8 channels: tap-major tables
    (10, 9, (1, 9), (1, 8), (1, 4), (0, 2), (0, 3)),
]


@pytest.mark.parametrize("case", TCASES)
def test_convT_fwd_dgrad_wgrad(case):
    Cin, Cout, (IA, IB), (KA, KB), stride, lo, hi = case
    g = torch.Generator().manual_seed(1)
    N = 2
    x = torch.randn(N, Cin, IA, IB, generator=g, requires_grad=True)
    w = torch.randn(Cin, Cout, KA, KB, generator=g, requires_grad=True)
    b = torch.randn(Cout, generator=g, requires_grad=True)
    full = F.conv_transpose2d(x, w, b, stride)
    LA, LB = full.shape[2] - lo[0] - hi[0], full.shape[3] - lo[1] - hi[1]
    y = full[:, :, lo[0]:lo[0] + LA, lo[1]:lo[1] + LB]
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    yc = y.detach().contiguous()

    out = torch.full((yc.numel(),), float("nan"))
    for p in convplan.convT_fwd_plans(tuple(x.shape), x.stride(), tuple(w.shape), stride, (1, 1), lo, (LA, LB),
                                      yc.stride()):
        emulate_fwd(p, w.detach().reshape(-1), x.detach().reshape(-1), out, b.detach())
    torch.testing.assert_close(out.view(yc.shape), yc, rtol=1e-4, atol=1e-4)

    p = convplan.convT_dgrad_plan(tuple(x.shape), x.stride(), tuple(w.shape), stride, (1, 1), lo,
                                  tuple(gy.shape), gy.stride())
    dx = torch.full((x.numel(),), float("nan"))
    emulate_fwd(p, w.detach().reshape(-1), gy.reshape(-1), dx)
    torch.testing.assert_close(dx.view(x.shape), x.grad, rtol=1e-4, atol=1e-4)
    dap = emulate_wgrad(p, gy.reshape(-1), x.detach().reshape(-1))
    torch.testing.assert_close(scatter_weights(p, dap, w.shape), w.grad, rtol=1e-4, atol=1e-4)


def test_strided_views_and_pick_r():
    # DConv on the freq branch: a (1,3) conv over a (B, C, Fr, T) tensor == per-row Conv1d
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 4, 5, 12, generator=g)
    w = torch.randn(3, 4, 1, 3, generator=g)
    y = F.conv2d(x, w, None, 1, (0, 2), (1, 2))
    ref = F.conv1d(x.permute(0, 2, 1, 3).reshape(-1, 4, 12), w[:, :, 0], None, 1, 2, 2)
    torch.testing.assert_close(y.permute(0, 2, 1, 3).reshape(-1, 3, 12), ref)
    xv = x.permute(0, 1, 3, 2)             # non-contiguous input view
    yv = F.conv2d(xv, w.permute(0, 1, 3, 2), None, 1, (2, 0), (2, 1))
    plan = convplan.conv_fwd_plan(tuple(xv.shape), xv.stride(), (3, 4, 3, 1), (1, 1), (2, 0), (2, 1),
                                  yv.contiguous().stride())
    out = torch.zeros(yv.numel())
    emulate_fwd(plan, w.permute(0, 1, 3, 2).contiguous().reshape(-1), x.reshape(-1), out)
    torch.testing.assert_close(out.view(yv.shape), yv, rtol=1e-4, atol=1e-4)
    for M, r in [(1, 0), (8, 0), (9, 1), (32, 1), (45, 2), (48, 2), (90, 3), (96, 3), (128, 4),
                 (135, 3), (192, 3), (256, 4), (1536, 4)]:
        assert convplan.pick_r(M) == r, (M, convplan.pick_r(M))
        assert convplan.mpad_for(M) % 4 == 0 and convplan.mpad_for(M) >= M


MERGED_T = [
    # Cin, Cout, (IA, IB), (KA, KB), stride, crop_lo, crop_hi, N
    (6, 5, (1, 37), (1, 8), (1, 4), (0, 3), (0, 1), 2),
    (4, 3, (9, 5), (8, 1), (4, 1), (2, 0), (2, 0), 1),
    (3, 2, (1, 11), (1, 16), (1, 8), (0, 5), (0, 2), 1),
    (5, 1, (1, 9), (1, 1024), (1, 256), (0, 0), (0, 0), 1),      # DCUNet synthesis filterbank geometry
]


@pytest.mark.parametrize("case", MERGED_T)
def test_merged_phase_transposed_conv(case):
    """All stride phases of a transposed convolution as ONE GEMM (convplan.merged_phase_plan + ops._merged_weight)."""
    from remfx_amd import ops
    Cin, Cout, (IA, IB), (KA, KB), stride, lo, hi, N = case
    torch.manual_seed(0)
    x = torch.randn(N, Cin, IA, IB)
    w = torch.randn(Cin, Cout, KA, KB)
    bias = torch.randn(Cout)
    full = F.conv_transpose2d(x, w, bias, stride)
    ref = full[:, :, lo[0]:full.shape[2] - hi[0], lo[1]:full.shape[3] - hi[1]]
    ax = ops._merge_axis(w.shape[2:], stride, (0, 0), (1, 1))
    assert ax is not None
    wm, J = ops._merged_weight(w.transpose(0, 1), ax, stride[ax])
    out = torch.full(ref.shape, float("nan"))
    plan = convplan.merged_phase_plan(tuple(x.shape), x.stride(), wm.shape[0], ax, stride[ax], J, -lo[ax],
                                      out.shape[2 + ax], out.stride())
    emulate_fwd(plan, wm.reshape(-1), x.reshape(-1), out.view(-1), bias)
    assert torch.allclose(out, ref, atol=1e-4), float((out - ref).abs().max())


MERGED_D = [
    # Cin, Cout, (IA, IB), (KA, KB), stride, padding, N
    (3, 6, (1, 47), (1, 8), (1, 4), (0, 2), 2),        # inputs past the last window get a zero gradient
    (2, 4, (45, 3), (4, 1), (2, 1), (1, 0), 1),
]


@pytest.mark.parametrize("case", MERGED_D)
def test_merged_phase_conv_input_gradient(case):
    from remfx_amd import ops
    Cin, Cout, (IA, IB), (KA, KB), stride, padding, N = case
    torch.manual_seed(1)
    x = torch.randn(N, Cin, IA, IB, requires_grad=True)
    w = torch.randn(Cout, Cin, KA, KB)
    y = F.conv2d(x, w, None, stride, padding)
    g = torch.randn_like(y)
    (ref,) = torch.autograd.grad(y, x, g)
    ax = ops._merge_axis(w.shape[2:], stride, padding, (1, 1))
    assert ax is not None
    wm, J = ops._merged_weight(w.transpose(0, 1), ax, stride[ax])
    dx = torch.full(ref.shape, float("nan"))
    plan = convplan.merged_phase_plan(tuple(g.shape), g.stride(), wm.shape[0], ax, stride[ax], J, -padding[ax],
                                      dx.shape[2 + ax], dx.stride())
    emulate_fwd(plan, wm.reshape(-1), g.reshape(-1), dx.view(-1))
    assert torch.allclose(dx, ref, atol=1e-4), float((dx - ref).abs().max())

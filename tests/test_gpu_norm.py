"""GPU parity: fused GroupNorm(+GELU/GLU/LayerScale-residual) and GLU kernels vs torch CPU."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rms(a, b):
    return float(((a - b) ** 2).mean().sqrt())


@pytest.mark.parametrize("shape,groups", [((1024, 2, 20), 1), ((3, 48, 7, 33), 4), ((2, 12, 5000), 1),
                                          ((5, 96, 64), 4), ((2, 8, 3, 50), 1),
                                          ((2, 6, 13000), 3),     # four 4096-sample chunks per row: slotted backward partials
                                          ((600, 12, 150), 1),    # 600 small samples: fused per-sample backward
                                          ((520, 20, 70), 1),     # 13-24 channels: the wider wave-per-sample variant
                                          ((513, 130, 40), 1),    # 65 channel pairs: register kernel that streams gy twice
                                          ((2100, 8, 16), 1),     # >= 2048 samples: sliced channel sums (gn_bwd_chansum_split)
                                          ((3, 64, 24), 4)])      # short rows (S = 24): statistics walk contiguous groups
@pytest.mark.parametrize("mode", ["none", "gelu", "glu", "glu_scale_res"])
def test_groupnorm_modes(shape, groups, mode):
    from remfx_amd import nnops
    g = torch.Generator().manual_seed(hash((shape, groups, mode)) % 1000)
    C = shape[1]
    x = torch.randn(shape, generator=g) * 2 + 0.5
    w, b = torch.randn(C, generator=g), torch.randn(C, generator=g)
    oshape = list(shape)
    if mode.startswith("glu"):
        oshape[1] = C // 2
    res = torch.randn(oshape, generator=g)
    sc = torch.randn(oshape[1], generator=g)
    gy = torch.randn(oshape, generator=g)

    def ref(x, w, b, res, sc):
        u = F.group_norm(x, groups, w, b, 1e-5)
        if mode == "gelu":
            return F.gelu(u)
        if mode == "glu":
            return F.glu(u, 1)
        if mode == "glu_scale_res":
            return res + sc.view(1, -1, *([1] * (x.dim() - 2))) * F.glu(u, 1)
        return u
    tr = [t.clone().requires_grad_(True) for t in (x, w, b, res, sc)]
    ref(*tr).backward(gy)
    td = [t.to(DEV).requires_grad_(True) for t in (x, w, b, res, sc)]
    if mode == "glu_scale_res":
        out = nnops.group_norm(td[0], groups, td[1], td[2], 1e-5, mode, res=td[3], scale=td[4])
    else:
        out = nnops.group_norm(td[0], groups, td[1], td[2], 1e-5, mode)
    out.backward(gy.to(DEV))
    assert _rms(out.detach().cpu(), ref(x, w, b, res, sc)) < 1e-5
    n = 5 if mode == "glu_scale_res" else 3
    for i in range(n):
        r = tr[i].grad
        scale = max(1.0, float(r.abs().max()))
        assert _rms(td[i].grad.cpu(), r) < 2e-5 * scale, (i, _rms(td[i].grad.cpu(), r), scale)


@pytest.mark.parametrize("shape", [(3, 16, 9, 31), (2, 8, 70, 130), (40, 4, 33, 7)])
@pytest.mark.parametrize("relu", [False, True])
def test_batchnorm_train_and_repeat(shape, relu):
    """nn.BatchNorm2d (+ReLU) in training mode, forward / backward / running statistics, against torch on the CPU; rows of 9100 values
    span three statistics chunks.  The statistics and the backward partials are slot stores added in a fixed order: a second call
    returns the same bits."""
    from remfx_amd import nnops
    g = torch.Generator().manual_seed(sum(shape) + relu)
    C = shape[1]
    x = torch.randn(shape, generator=g) * 1.5 - 0.3
    gy = torch.randn(shape, generator=g)
    bn_ref = torch.nn.BatchNorm2d(C)
    with torch.no_grad():
        bn_ref.weight.copy_(torch.randn(C, generator=g)); bn_ref.bias.copy_(torch.randn(C, generator=g))
    outs = []
    for _ in range(2):
        bn_dev = torch.nn.BatchNorm2d(C)
        bn_dev.load_state_dict(bn_ref.state_dict())
        bn_dev = bn_dev.to(DEV)
        xd = x.to(DEV).requires_grad_(True)
        y = nnops.batch_norm(xd, bn_dev, True, relu=relu)
        y.backward(gy.to(DEV))
        outs.append((y.detach().cpu(), xd.grad.cpu(), bn_dev.weight.grad.cpu(), bn_dev.bias.grad.cpu(),
                     bn_dev.running_mean.cpu(), bn_dev.running_var.cpu()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    xr = x.clone().requires_grad_(True)
    bn_ref.train()
    yr = bn_ref(xr)
    yr = F.relu(yr) if relu else yr
    yr.backward(gy)
    y, dx, dw, db, rm, rv = outs[0]
    assert _rms(y, yr.detach()) < 1e-5
    for got, want in ((dx, xr.grad), (dw, bn_ref.weight.grad), (db, bn_ref.bias.grad), (rm, bn_ref.running_mean), (rv, bn_ref.running_var)):
        scale = max(1.0, float(want.abs().max()))
        assert _rms(got, want) < 2e-5 * scale


def test_glu_plain():
    from remfx_amd import nnops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 10, 4, 9, generator=g)
    gy = torch.randn(3, 5, 4, 9, generator=g)
    xr = x.clone().requires_grad_(True)
    F.glu(xr, 1).backward(gy)
    xd = x.to(DEV).requires_grad_(True)
    y = nnops.glu(xd, 1)
    y.backward(gy.to(DEV))
    assert _rms(y.detach().cpu(), F.glu(x, 1)) < 1e-6
    assert _rms(xd.grad.cpu(), xr.grad) < 1e-6


@pytest.mark.parametrize("shape", [(3, 10, 4, 9), (2, 6, 1030), (5, 4, 3, 7)])
def test_glu_shapes(shape):
    """vector path (C/2 * S multiple of 4) and the scalar fallback"""
    from remfx_amd import nnops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(shape, generator=g)
    oshape = list(shape)
    oshape[1] //= 2
    gy = torch.randn(oshape, generator=g)
    xr = x.clone().requires_grad_(True)
    F.glu(xr, 1).backward(gy)
    xd = x.to(DEV).requires_grad_(True)
    y = nnops.glu(xd, 1)
    y.backward(gy.to(DEV))
    assert _rms(y.detach().cpu(), F.glu(x, 1)) < 1e-6
    assert _rms(xd.grad.cpu(), xr.grad) < 1e-6


@pytest.mark.parametrize("xshape,yshape", [((2, 3, 4, 300), (2, 3, 4, 300)), ((2, 3, 5, 77), (1, 3, 5, 1)),
                                           ((3, 4, 2, 513), (3, 4, 1, 513)), ((2, 5, 1001), (2, 5, 1001)),
                                           ((2, 5, 7, 3), (1, 1, 7, 3))])
def test_add_broadcast(xshape, yshape):
    """x + alpha * y: flat path, row-decoded broadcast path, and their gradients (HDemucs skip / freq-embedding adds)"""
    from remfx_amd import nnops
    g = torch.Generator().manual_seed(2)
    x, y, gy = torch.randn(xshape, generator=g), torch.randn(yshape, generator=g), torch.randn(xshape, generator=g)
    xr, yr = x.clone().requires_grad_(True), y.clone().requires_grad_(True)
    (xr + 0.3 * yr).backward(gy)
    xd, yd = x.to(DEV).requires_grad_(True), y.to(DEV).requires_grad_(True)
    out = nnops.add(xd, yd, 0.3)
    out.backward(gy.to(DEV))
    assert _rms(out.detach().cpu(), x + 0.3 * y) < 1e-6
    assert _rms(xd.grad.cpu(), xr.grad) < 1e-6
    assert _rms(yd.grad.cpu(), yr.grad) < 1e-5


@pytest.mark.parametrize("shape", [(2, 6, 5, 64), (3, 4, 7, 50)])
def test_activation_to_permuted_layout(shape):
    """ops.activation_to: GELU written in (B, Fr, C, T) memory order (vector and scalar row paths); the consumer's
    permute + reshape is a view and the gradient is read from that layout."""
    from remfx_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn(shape, generator=g)
    B, C, Fr, T = shape
    gy = torch.randn(B * Fr, C, T, generator=g)
    xr = x.clone().requires_grad_(True)
    zr = F.gelu(xr).permute(0, 2, 1, 3).reshape(-1, C, T)
    (zr * gy).sum().backward()
    xd = x.to(DEV).requires_grad_(True)
    y = ops.activation_to(xd, "gelu", (0, 2, 1))
    assert y.shape == xd.shape and y.permute(0, 2, 1, 3).is_contiguous()
    z = y.permute(0, 2, 1, 3).reshape(-1, C, T)
    assert z.data_ptr() == y.data_ptr()                      # a view, not a copy
    (z * gy.to(DEV)).sum().backward()
    assert _rms(z.detach().cpu(), zr.detach()) < 1e-6
    assert _rms(xd.grad.cpu(), xr.grad) < 1e-6


@pytest.mark.parametrize("shape", [(3, 5, 7, 300), (2, 6, 1, 4099), (64, 4, 1, 1024), (1, 3, 40, 1000), (2, 3, 4, 5)])
def test_channel_sum(shape):
    """bias-gradient reduction over (N, A, B): vectorised contiguous-plane kernel (planes >= 1024 values, incl. a ragged tail and a
    channel-sliced view) and the strided fallback, vs an fp64 sum."""
    from remfx_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(shape, generator=g).to(DEV)
    ref = x.double().sum(dim=(0, 2, 3))
    got = ops.channel_sum(x)
    assert float((got.double() - ref).abs().max()) <= 2e-6 * float(x.abs().double().sum() / shape[1] + 1.0)
    if shape[1] > 2:
        v = x[:, 1:-1]                                             # non-contiguous over (n, c), contiguous planes
        got = ops.channel_sum(v)
        assert float((got.double() - ref[1:-1]).abs().max()) <= 2e-6 * float(x.abs().double().sum() / shape[1] + 1.0)


def test_dropout_native():
    """rfx_dropout: keep rate, 1/(1-p) scaling, same mask in backward, repeatable under torch.manual_seed, identity in eval."""
    from remfx_amd import nnops
    x = torch.randn(1 << 20, device=DEV).abs() + 0.5
    x.requires_grad_(True)
    torch.manual_seed(11)
    y = nnops.dropout(x, 0.4, True)
    keep = (y != 0)
    assert abs(float(keep.float().mean()) - 0.6) < 3e-3
    torch.testing.assert_close(y[keep], (x.detach() / 0.6)[keep])
    y.sum().backward()
    torch.testing.assert_close(x.grad, keep.float() / 0.6)
    torch.manual_seed(11)
    y2 = nnops.dropout(x.detach(), 0.4, True)
    assert torch.equal(y2, y.detach())
    y3 = nnops.dropout(x.detach(), 0.4, True)          # next draw differs
    assert not torch.equal(y3, y2)
    assert nnops.dropout(x, 0.4, False) is x
    # no visible structure along the index: keep rates of 1024-element blocks scatter binomially
    blk = keep.float().view(-1, 1024).mean(1)
    assert float(blk.std()) < 2.5 * (0.6 * 0.4 / 1024) ** 0.5

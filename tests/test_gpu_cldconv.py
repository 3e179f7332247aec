"""GPU: the fused channels-last DConv depth-layer kernels (csrc/cl_dconv.hip) against the channel-major DConv of hdemucs._DConv
(torchaudio HDemucs `_DConv`; reference call site remfx/models.py:319): forward and every gradient.  Reference = the channel-major
path in the exact-fp32 mode; the channels-last kernels must not be further from it than the channel-major bf16 path is."""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.one_mode]
DEV = "cuda:0"


def _rel(a, b):
    return float(((a.double() - b.double()) ** 2).sum().sqrt() / (b.double() ** 2).sum().sqrt().clamp_min(1e-30))


def _run_cm(mod, x, gy, mode):
    from remfx_amd import ops
    prev = ops.gemm_precision()
    ops.set_gemm_precision(mode)
    try:
        mod.zero_grad(set_to_none=True)
        xr = x.clone().requires_grad_(True)
        y = mod(xr)
        y.backward(gy)
        torch.cuda.synchronize()
        return y.detach(), xr.grad.detach(), {n: p.grad.detach().clone() for n, p in mod.named_parameters()}
    finally:
        ops.set_gemm_precision(prev)


def _run_cl(mod, x, gy, Bn, A):
    from remfx_amd import cldconv, ops
    prev = ops.gemm_precision()
    ops.set_gemm_precision("bf16")
    try:
        mod.zero_grad(set_to_none=True)
        S, Cc, T = x.shape
        xc = x.view(Bn, A, Cc, T).permute(0, 1, 3, 2).contiguous().to(torch.bfloat16).requires_grad_(True)
        gc = gy.view(Bn, A, Cc, T).permute(0, 1, 3, 2).contiguous().to(torch.bfloat16)
        h = xc
        for seq, (dil, pad, lstm, attn) in zip(mod.layers, mod.spec):
            m = list(seq)
            h = cldconv.dconv_layer(h, m[0], m[1], m[3], m[4], m[6].scale, dil)
        h.backward(gc)
        torch.cuda.synchronize()
        back = lambda t: t.detach().float().permute(0, 1, 3, 2).reshape(S, Cc, T)
        return back(h), back(xc.grad), {n: p.grad.detach().clone() for n, p in mod.named_parameters()}
    finally:
        ops.set_gemm_precision(prev)


@pytest.mark.parametrize("Cc,Bn,A,T", [(48, 2, 3, 256), (48, 3, 130, 256), (96, 2, 3, 256), (96, 1, 70, 256), (48, 3, 1, 1024), (96, 2, 1, 2048),
                                        (48, 2, 2, 512)])
def test_cl_dconv_vs_channel_major(Cc, Bn, A, T):
    from remfx_amd.hdemucs import _DConv
    torch.manual_seed(0)
    mod = _DConv(Cc, depth=2, init=0.3).to(DEV)
    with torch.no_grad():
        for n, p in mod.named_parameters():
            if p.dim() == 1 and ("1.weight" in n or "4.weight" in n):        # GroupNorm weights away from 1, biases away from 0
                p.add_(torch.randn_like(p) * 0.2)
            if p.dim() == 1 and ("1.bias" in n or "4.bias" in n):
                p.add_(torch.randn_like(p) * 0.2)
    g = torch.Generator().manual_seed(1)
    S = Bn * A
    x = torch.randn(S, Cc, T, generator=g).to(DEV)
    x = x.to(torch.bfloat16).float()                                        # both paths see the same (bf16-representable) input
    gy = torch.randn(S, Cc, T, generator=g).to(DEV).to(torch.bfloat16).float()
    y32, dx32, g32 = _run_cm(mod, x, gy, "f32")
    y16, dx16, g16 = _run_cm(mod, x, gy, "bf16")
    ycl, dxcl, gcl = _run_cl(mod, x, gy, Bn, A)
    e16, ecl = _rel(y16, y32), _rel(ycl, y32)
    print(f"y: channel-major bf16 {e16:.3e}, channels-last {ecl:.3e}")
    assert ecl < 2.0 * e16 + 4e-3                                           # + the bf16 rounding of the stored y itself
    e16, ecl = _rel(dx16, dx32), _rel(dxcl, dx32)
    print(f"dx: channel-major bf16 {e16:.3e}, channels-last {ecl:.3e}")
    assert ecl < 2.0 * e16 + 4e-3
    for n in g32:
        e16, ecl = _rel(g16[n], g32[n]), _rel(gcl[n], g32[n])
        print(f"{n:28s} channel-major bf16 {e16:.3e}, channels-last {ecl:.3e}")
        assert ecl < 2.5 * e16 + 5e-3, n
    # deterministic
    ycl2, dxcl2, gcl2 = _run_cl(mod, x, gy, Bn, A)
    assert torch.equal(ycl, ycl2) and torch.equal(dxcl, dxcl2)
    for n in gcl:
        assert torch.equal(gcl[n], gcl2[n]), n


@pytest.mark.parametrize("Cc,Bn,A,T,grid", [(48, 1, 23, 256, 3), (48, 2, 8, 256, 5), (96, 1, 11, 256, 2), (48, 2, 1, 1792, 3), (96, 1, 1, 1280, 2),
                                             (48, 1, 7, 256, 7)])
def test_cl_dconv_persistent_walk_matches_one_sample_per_workgroup(monkeypatch, Cc, Bn, A, T, grid):
    """The kernels are persistent: a workgroup walks samples grid apart, and since round 6 fetches the NEXT sample's operands (untracked
    LDS-DMA into alternating images, counted waits) while it works on this one.  With a grid of 2 - 7 workgroups every workgroup walks
    3 - 12 samples (odd and even counts, the last one without a successor, tiles of multi-tile samples split across workgroups): outputs
    and every gradient must equal -- bit for bit -- those of the default grid, where a workgroup sees at most two samples."""
    from remfx_amd import cldconv
    from remfx_amd.hdemucs import _DConv
    torch.manual_seed(3)
    mod = _DConv(Cc, depth=2, init=0.3).to(DEV)
    g = torch.Generator().manual_seed(5)
    S = Bn * A
    x = torch.randn(S, Cc, T, generator=g).to(DEV).to(torch.bfloat16).float()
    gy = torch.randn(S, Cc, T, generator=g).to(DEV).to(torch.bfloat16).float()
    y0, dx0, g0 = _run_cl(mod, x, gy, Bn, A)
    monkeypatch.setattr(cldconv, "GRID", grid)
    monkeypatch.setattr(cldconv, "GRID_FWD", grid)
    y1, dx1, g1 = _run_cl(mod, x, gy, Bn, A)
    assert torch.equal(y0, y1) and torch.equal(dx0, dx1)
    for n in g0:
        # parameter gradients are sums over workgroups in workgroup order: the grouping changes with the grid, the values by rounding only
        assert _rel(g1[n], g0[n]) < 1e-5, n


def test_cl_dconv_small_gradients_through_the_sink():
    """With a GradSink armed (parameters in an optim.FlatParams buffer) the backward writes dscale / GroupNorm affine gradients
    straight into the flat gradient buffer -- same values as the tensors it returns to autograd without a sink, twice in a row
    (the second pass accumulates: 2x)."""
    from remfx_amd import cldconv, ops
    from remfx_amd.hdemucs import _DConv
    from remfx_amd.optim import FlatParams
    torch.manual_seed(3)
    Cc, Bn, A, T = 48, 2, 5, 256
    mod = _DConv(Cc, depth=2, init=0.3).to(DEV)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(Bn * A, Cc, T, generator=g).to(DEV).to(torch.bfloat16).float()
    gy = torch.randn(Bn * A, Cc, T, generator=g).to(DEV).to(torch.bfloat16).float()
    _, _, ref = _run_cl(mod, x, gy, Bn, A)
    flat = FlatParams(list(mod.parameters()))
    assert flat.sink is not None
    prev = ops.gemm_precision()
    ops.set_gemm_precision("bf16")
    try:
        flat.zero_grad()
        for rep in range(2):
            xc = x.view(Bn, A, Cc, T).permute(0, 1, 3, 2).contiguous().to(torch.bfloat16).requires_grad_(True)
            gc = gy.view(Bn, A, Cc, T).permute(0, 1, 3, 2).contiguous().to(torch.bfloat16)
            h = xc
            for seq, (dil, pad, lstm, attn) in zip(mod.layers, mod.spec):
                m = list(seq)
                h = cldconv.dconv_layer(h, m[0], m[1], m[3], m[4], m[6].scale, dil)
            h.backward(gc)
        nsunk = sum(1 for w in flat.sink.writes if w)
        flat.join()
        torch.cuda.synchronize()
    finally:
        ops.set_gemm_precision(prev)
    assert nsunk == len(flat.params), (nsunk, len(flat.params))            # every parameter of the branch, the five small ones included
    for n, p in mod.named_parameters():
        assert torch.allclose(p.grad, 2 * ref[n], rtol=1e-6, atol=1e-7), n

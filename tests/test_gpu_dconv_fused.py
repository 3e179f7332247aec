"""GPU: the fused DConv depth-layer forward kernel + its layer-by-layer backward (csrc/dconv.hip, bf16 mode) against the fp32 CPU oracle (oracle/ref_hdemucs.DConv =
torchaudio HDemucs `_DConv`, unpinned upstream) and against the layer-by-layer bf16 path they replace."""
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.one_mode]
DEV = "cuda:0"


def _rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm().clamp_min(1e-30))


def _pair(C, seed):
    from oracle import ref_hdemucs
    from remfx_amd import hdemucs
    torch.manual_seed(seed)
    ref = ref_hdemucs.DConv(C, compress=4, depth=2, init=1e-4, attn=False, lstm=False)
    with torch.no_grad():
        for n, p in ref.named_parameters():
            if n.endswith(".scale"):
                p.fill_(0.3)
            elif n.endswith(("1.weight", "4.weight")) and p.dim() == 1:      # GroupNorm affines away from (1, 0)
                p.copy_(1.0 + 0.2 * torch.randn_like(p))
            elif n.endswith(("1.bias", "4.bias")) and p.dim() == 1:
                p.copy_(0.1 * torch.randn_like(p))
    net = hdemucs._DConv(C, compress=4, depth=2, init=1e-4, attn=False, lstm=False)
    net.load_state_dict(ref.state_dict(), strict=True)
    return ref, net.to(DEV)


@pytest.mark.parametrize("fused_bwd", [True, False])
@pytest.mark.parametrize("N", [5, 300, 1026])
def test_fused_dconv_layer_vs_oracle_and_unfused(N, fused_bwd, monkeypatch):
    """fused_bwd: the one-launch backward (dconv_bwd_kernel: forward recomputed, dx + dz / a / dh + parameter-gradient partial rows;
    N = 5 ends in a half-empty sample pair, 1026 gives the 256 persistent workgroups more than one pair each) or round 3's
    layer-by-layer backward behind the fused forward."""
    from remfx_amd import nnops, ops
    prev = ops.gemm_precision()
    ops.set_gemm_precision("bf16")
    monkeypatch.setattr(nnops, "DCONV_FUSED_BWD", fused_bwd)
    try:
        ref, net = _pair(48, 1)
        g = torch.Generator().manual_seed(2)
        x = torch.randn(N, 48, 256, generator=g)
        gy = torch.randn(N, 48, 256, generator=g)
        xr = x.clone().requires_grad_(True)
        yr = ref(xr)
        yr.backward(gy)
        calls = []
        orig = nnops.dconv_layer
        monkeypatch.setattr(nnops, "dconv_layer", lambda *a, **k: (calls.append(1), orig(*a, **k))[1])
        xd = x.to(DEV).requires_grad_(True)
        yd = net(xd)
        assert len(calls) == 2                                     # both depth layers took the fused kernels
        yd.backward(gy.to(DEV))
        fused = {n: p.grad.clone() for n, p in net.named_parameters()}
        gx_fused, y_fused = xd.grad.clone(), yd.detach().clone()
        # the layer-by-layer bf16 path on the same weights
        monkeypatch.setattr(nnops, "dconv_layer_fused_ok", lambda *a, **k: False)
        for p in net.parameters():
            p.grad = None
        xu = x.to(DEV).requires_grad_(True)
        yu = net(xu)
        yu.backward(gy.to(DEV))
        assert len(calls) == 2
        # forward: branch = y - x carries the arithmetic (x passes through exactly)
        br_ref = (yr.detach() - x)
        e_f, e_u = _rel(y_fused.cpu() - x, br_ref), _rel(yu.detach().cpu() - x, br_ref)
        print(f"N={N}: branch error vs fp32 oracle: fused {e_f:.2e}, layer-by-layer {e_u:.2e}")
        assert e_f < 1.5e-2 and e_f < 2.0 * e_u + 2e-3
        e_f, e_u = _rel(gx_fused.cpu() - gy, xr.grad - gy), _rel(xu.grad.cpu() - gy, xr.grad - gy)
        print(f"      dx (branch part): fused {e_f:.2e}, layer-by-layer {e_u:.2e}")
        assert e_f < 3e-2 and e_f < 2.0 * e_u + 5e-3
        refg = dict(ref.named_parameters())
        bad = []
        for n, p in net.named_parameters():
            r = refg[n].grad
            e_f, e_u = _rel(fused[n], r), _rel(p.grad, r)
            print(f"      {n:28s} fused {e_f:.2e}  layer-by-layer {e_u:.2e}")
            bad = bad + [n] if not (e_f < 4e-2 and e_f < 2.5 * e_u + 1e-2) else bad
        assert not bad, bad
    finally:
        ops.set_gemm_precision(prev)

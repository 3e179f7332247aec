"""GPU parity: gather-GEMM conv kernels (through the C ABI) vs torch fp32 CPU ops
and vs the TCN golden vectors recorded from the imported reference."""
import os

import numpy as np
import pytest
import torch

from tests.conftest import check, mode, tol
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _rms(a, b):
    return float(((a - b) ** 2).mean().sqrt())


CASES = [
    # Cin, Cout, (IA, IB), (KA, KB), stride, padding, dilation, N
    (3, 5, (1, 500), (1, 7), (1, 1), (0, 0), (1, 4), 2),
    (16, 48, (1, 3000), (1, 7), (1, 1), (0, 0), (1, 16), 2),
    (64, 256, (1, 2100), (1, 7), (1, 1), (0, 0), (1, 2), 1),     # R=4, 2 M tiles
    (2, 48, (64, 40), (8, 1), (4, 1), (2, 0), (1, 1), 2),        # HDemucs freq enc 0
    (48, 96, (32, 40), (8, 1), (4, 1), (2, 0), (1, 1), 2),       # R=3
    (1, 48, (1, 4096), (1, 8), (1, 4), (0, 2), (1, 1), 2),       # HDemucs time enc 0
    (24, 33, (17, 23), (3, 3), (1, 1), (1, 1), (1, 1), 2),       # 3x3
    (8, 45, (40, 30), (7, 5), (2, 2), (3, 2), (1, 1), 2),        # DCUNet R=2
    (12, 7, (9, 18), (5, 3), (2, 1), (2, 1), (1, 1), 2),         # thin M<=8
    (40, 2, (1, 700), (1, 1), (1, 1), (0, 0), (1, 1), 3),        # thin 1x1
    (96, 192, (8, 64), (1, 1), (1, 1), (0, 0), (1, 1), 2),       # 1x1 rewrite
    (16, 24, (1, 47), (1, 8), (1, 4), (0, 2), (1, 1), 2),        # merged-phase dgrad: tail inputs no window covers
    (10, 12, (45, 3), (4, 1), (2, 1), (1, 0), (1, 1), 1),        # merged-phase dgrad on the A axis, stride 2
    (48, 12, (1, 900), (1, 3), (1, 1), (0, 2), (1, 2), 3),       # DConv bottleneck: narrow wgrad tile (M <= 32), K = 145
    (96, 24, (1, 333), (1, 3), (1, 1), (0, 1), (1, 1), 2),       # narrow wgrad tile, two k tiles (K = 289)
    (20, 32, (6, 50), (1, 3), (1, 1), (0, 1), (1, 1), 2),        # narrow wgrad tile, M = 32 exactly, K = 61
    (24, 40, (1, 96), (3, 3), (1, 1), (1, 1), (1, 1), 2),        # 3x3 over ONE row (HDemucs decoder 4 rewrite): 6 of 9 taps only meet padding and are pruned; their dW must come back 0
    (16, 24, (2, 30), (5, 3), (1, 1), (2, 1), (1, 1), 2),        # 5x3 over two rows: the outermost row taps meet padding only for some outputs (kept), none pruned
    (16, 24, (2, 30), (7, 1), (1, 1), (3, 0), (1, 1), 2),        # 7x1 over two rows: taps 0 and 6 pruned, 1-5 live
]


@pytest.mark.parametrize("case", CASES)
def test_conv2d_fwd_bwd(case):
    from remfx_amd import ops
    dev = _dev()
    Cin, Cout, (IA, IB), (KA, KB), stride, padding, dilation, N = case
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, Cin, IA, IB, generator=g)
    w = torch.randn(Cout, Cin, KA, KB, generator=g) / (Cin * KA * KB) ** 0.5
    b = torch.randn(Cout, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    y = F.conv2d(xr, wr, br, stride, padding, dilation)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd, wd, bd = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    yd = ops.conv2d(xd, wd, bd, stride, padding, dilation)
    assert yd.shape == y.shape
    yd.backward(gy.to(dev))
    check(_rms(yd.detach().cpu(), y.detach()), 1e-5, max(1.0, float(y.detach().abs().max())))
    for got, ref, name in ((xd.grad, xr.grad, "dx"), (wd.grad, wr.grad, "dw"), (bd.grad, br.grad, "db")):
        scale = max(1.0, float(ref.abs().max()))
        check(_rms(got.cpu(), ref), 2e-5, scale, what=(name, _rms(got.cpu(), ref), scale))


TCASES = [
    # Cin, Cout, (IA, IB), (KA, KB), stride, crop_lo, crop_hi, N
    (48, 24, (16, 40), (8, 1), (4, 1), (2, 0), (2, 0), 2),
    (48, 1, (1, 300), (1, 8), (1, 4), (0, 2), (0, 5), 2),       # thin output
    (64, 48, (1, 257), (1, 4), (1, 2), (0, 1), (0, 1), 2),
    (20, 45, (9, 17), (5, 3), (2, 1), (2, 1), (2, 1), 2),
    (12, 9, (6, 7), (7, 5), (2, 2), (3, 2), (3, 2), 2),
    (40, 33, (1, 50), (8, 1), (4, 1), (0, 0), (0, 0), 2),
    (48, 24, (1, 37), (1, 8), (1, 4), (0, 3), (0, 1), 2),       # merged phases, crop not a multiple of the stride
    (6, 5, (3, 21), (1, 16), (1, 8), (0, 5), (0, 2), 1),        # stride 8, two taps per phase
    (48, 2, (16, 40), (8, 1), (4, 1), (2, 0), (2, 0), 2),       # last freq decoder: 2 channels x 4 phases merged on the MFMA kernel (R = 1, M = 8)
]


@pytest.mark.parametrize("case", TCASES)
def test_convT_fwd_bwd(case):
    from remfx_amd import ops
    dev = _dev()
    Cin, Cout, (IA, IB), (KA, KB), stride, lo, hi, N = case
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, Cin, IA, IB, generator=g)
    w = torch.randn(Cin, Cout, KA, KB, generator=g) / (Cin * KA * KB) ** 0.5
    b = torch.randn(Cout, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    full = F.conv_transpose2d(xr, wr, br, stride)
    LA, LB = full.shape[2] - lo[0] - hi[0], full.shape[3] - lo[1] - hi[1]
    y = full[:, :, lo[0]:lo[0] + LA, lo[1]:lo[1] + LB]
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    xd, wd, bd = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    yd = ops.conv_transpose2d(xd, wd, bd, stride, (1, 1), lo, (LA, LB))
    yd.backward(gy.to(dev))
    check(_rms(yd.detach().cpu(), y.detach()), 1e-5, max(1.0, float(y.detach().abs().max())))
    for got, ref, name in ((xd.grad, xr.grad, "dx"), (wd.grad, wr.grad, "dw"), (bd.grad, br.grad, "db")):
        scale = max(1.0, float(ref.abs().max()))
        check(_rms(got.cpu(), ref), 2e-5, scale, what=(name, _rms(got.cpu(), ref), scale))


def test_strided_input_view():
    """DConv on the HDemucs freq branch: (1,3) dilated conv directly on (B, C, Fr, T)."""
    from remfx_amd import ops
    dev = _dev()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 48, 16, 64, generator=g)
    w = torch.randn(12, 48, 1, 3, generator=g) * 0.1
    ref = F.conv2d(x, w, None, 1, (0, 2), (1, 2))
    got = ops.conv2d(x.to(dev), w.to(dev), None, (1, 1), (0, 2), (1, 2))
    check(_rms(got.cpu(), ref), 1e-5)
    xt = x.permute(0, 1, 3, 2)              # non-contiguous view as input
    ref2 = F.conv2d(xt, w.permute(0, 1, 3, 2), None, 1, (2, 0), (2, 1))
    got2 = ops.conv2d(x.to(dev).permute(0, 1, 3, 2), w.to(dev).permute(0, 1, 3, 2).contiguous(), None,
                      (1, 1), (2, 0), (2, 1))
    check(_rms(got2.cpu(), ref2), 1e-5)


@pytest.mark.parametrize("name", ["tcn_small", "tcn_mid", "tcn_causal"])
def test_tcn_golden(golden_dir, name):
    """HIP TCN vs outputs of the imported reference remfx.tcn.TCN (tests/golden)."""
    from oracle import ref_tcn
    from remfx_amd.tcn import TCN
    dev = _dev()
    gd = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = {k[4:]: gd[k].item() for k in gd.files if k.startswith("cfg_")}
    sd = ref_tcn.tcn_init_state_dict(cfg["ninputs"], cfg["noutputs"], cfg["nblocks"], cfg["channel_width"],
                                     cfg["kernel_size"], seed=int(gd["seed"]))
    for k in [k for k in sd if k.endswith("relu.weight")]:
        sd[k] = torch.linspace(0.05, 0.45, sd[k].numel())
    net = TCN(**{k: (bool(v) if k == "causal" else v) for k, v in cfg.items()})
    net.load_state_dict(sd, strict=True)
    net = net.to(dev)
    with torch.no_grad():
        y = net(torch.from_numpy(gd["x"]).to(dev)).cpu().numpy()
    assert y.shape == gd["y"].shape
    check(float(np.sqrt(((y - gd["y"]) ** 2).mean())), 1e-5)


def test_tcn_full_width_golden(golden_dir):
    """BASELINE config 2 at FULL width: the 20-block x 256-channel k = 7 HIP TCN (9 974 017 parameters) vs the forward output
    and autograd gradients of the imported reference remfx.tcn.TCN on one 32768-sample clip
    (tests/golden/tcn_full_fwd_bwd.npz, oracle/gen_golden.py::gen_tcn_full; reference remfx/tcn.py:62-138)."""
    from oracle import ref_tcn
    from remfx_amd.tcn import TCN
    dev = _dev()
    gd = np.load(os.path.join(golden_dir, "tcn_full_fwd_bwd.npz"))
    sd = ref_tcn.tcn_init_state_dict(1, 1, 20, 256, 7, seed=int(gd["seed"]))
    for i, k in enumerate([k for k in sd if k.endswith("relu.weight")]):
        sd[k] = torch.linspace(0.05, 0.45, sd[k].numel()).roll(7 * i)
    net = TCN(ninputs=1, noutputs=1, nblocks=20, channel_growth=0, channel_width=256, kernel_size=7, stack_size=10,
              dilation_growth=2, condition=False, latent_dim=2, norm_type="identity", causal=False, estimate_loudness=False)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev)
    gen = torch.Generator().manual_seed(int(gd["x_seed"]))
    x = torch.randn(1, 1, int(gd["T"]), generator=gen) * 0.5
    y = net(x.to(dev))
    assert tuple(y.shape) == gd["y"].shape
    yr = gd["y"]
    err = float(np.sqrt(((y.detach().cpu().numpy() - yr) ** 2).mean()))
    # output rms 1.6e-2 (tanh of a small pre-activation): the bound is relative to it.  Measured: see RFX_TOL_LOG
    check(err, 1e-4, float(np.sqrt((yr ** 2).mean())), bf16=5e-2, what=("y", err))
    r = torch.randn(y.shape, generator=gen)
    (y * r.to(dev)).sum().backward()
    params = dict(net.named_parameters())
    gtot = float(torch.sqrt(sum(p.grad.double().pow(2).sum() for p in net.parameters())))
    check(abs(gtot - float(gd["grad_total_norm"])), 1e-3, float(gd["grad_total_norm"]), bf16x3=5e-3, bf16=0.1, what=("gnorm", gtot))
    for n in gd["grad_names"].tolist():
        gr = params[n].grad.reshape(-1)
        sl = gr[:: max(1, gr.numel() // 512)][:512].cpu().numpy()
        ref = gd["gslice_" + n]
        e = float(np.sqrt(((sl - ref) ** 2).mean()))
        # 20 blocks deep: fp32 summation-order noise reaches 3.3e-4 of the tensor's max on block 0 (the end of the chain) in the
        # exact-fp32 mode; PReLU'(pre) flips at pre ~ 0 under product rounding (see test_tcn_backward_vs_oracle) set the others
        check(e, 1e-3, float(np.abs(ref).max()), bf16x3=1e-2, bf16=0.2, what=(n, e))


def test_tcn_full_width_full_length_golden(golden_dir):
    """BASELINE config 2 at its REAL shape: the 20-block x 256-channel TCN on one 262144-sample clip against the
    forward output of the imported reference remfx.tcn.TCN (tests/golden/tcn_full_length_fwd.npz, oracle/gen_golden.py::
    gen_tcn_full_length; reference remfx/tcn.py:62-138): output length, norm, head / tail runs and a stride-61 slice over the whole clip."""
    from oracle import ref_tcn
    from remfx_amd.tcn import TCN
    dev = _dev()
    gd = np.load(os.path.join(golden_dir, "tcn_full_length_fwd.npz"))
    sd = ref_tcn.tcn_init_state_dict(1, 1, 20, 256, 7, seed=int(gd["seed"]))
    for i, k in enumerate([k for k in sd if k.endswith("relu.weight")]):
        sd[k] = torch.linspace(0.05, 0.45, sd[k].numel()).roll(7 * i)
    net = TCN(ninputs=1, noutputs=1, nblocks=20, channel_growth=0, channel_width=256, kernel_size=7, stack_size=10,
              dilation_growth=2, condition=False, latent_dim=2, norm_type="identity", causal=False, estimate_loudness=False)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev)
    gen = torch.Generator().manual_seed(int(gd["x_seed"]))
    x = torch.randn(1, 1, int(gd["T"]), generator=gen) * 0.5
    with torch.no_grad():
        y = net(x.to(dev)).reshape(-1).cpu().numpy()
    assert y.size == int(gd["y_len"])
    rms = float(np.sqrt((gd["y_slice"] ** 2).mean()))
    for name, got, ref in (("slice", y[::int(gd["y_stride"])], gd["y_slice"]), ("head", y[:2048], gd["y_head"]), ("tail", y[-2048:], gd["y_tail"])):
        err = float(np.sqrt(((got - ref) ** 2).mean()))
        print(f"full-length TCN {name}: rms error {err:.3e} (output rms {rms:.3e})")
        check(err, 1e-4, rms, bf16=5e-2, what=(name, err))
    check(abs(float(np.sqrt((y.astype(np.float64) ** 2).sum())) - float(gd["y_norm"])), 1e-4, float(gd["y_norm"]), bf16=2e-2, what="norm")


def test_tcn_backward_vs_oracle():
    """fwd + all gradients of a reduced TCN vs autograd over the CPU oracle."""
    from oracle import ref_tcn
    from remfx_amd.tcn import TCN
    dev = _dev()
    cfg = dict(ninputs=1, noutputs=1, nblocks=5, channel_width=40, kernel_size=7, stack_size=3,
               dilation_growth=2, causal=False)
    sd = ref_tcn.tcn_init_state_dict(1, 1, 5, 40, 7, seed=11)
    for k in [k for k in sd if k.endswith("relu.weight")]:
        sd[k] = torch.linspace(0.05, 0.45, sd[k].numel())
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 1, 3000, generator=g)
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    y = ref_tcn.tcn_forward(x, sdr, 5, 3, 2, False)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    net = TCN(**cfg)
    net.load_state_dict(sd)
    net = net.to(dev)
    yd = net(x.to(dev))
    yd.backward(gy.to(dev))
    check(_rms(yd.detach().cpu(), y.detach()), 1e-5)
    for k, p in net.named_parameters():
        ref = sdr[k].grad
        scale = max(1e-3, float(ref.abs().max()))
        # bf16x3 / bf16: the backward re-materialises PReLU'(conv1(x)); a position whose pre-activation lies within the
        # product rounding of zero flips between slope and 1, an O(1) change of that term: measured 1e-6 (f32), 8.5e-4
        # (bf16x3: 2^-17 products) and 2.3e-2 (bf16: 2^-9) of the gradient's scale on block 0, the end of the chain
        check(_rms(p.grad.cpu(), ref), 1e-4, scale, bf16x3=2e-3, bf16=5e-2, what=(k, _rms(p.grad.cpu(), ref), scale))


@pytest.mark.parametrize("case", [
    (48, 96, (12, 40), (3, 3), (1, 1)),        # HDemucs freq rewrite conv (context 1) + GLU
    (24, 50, (1, 700), (1, 3), (0, 1)),        # time branch; 25 GLU channels: ragged last row tile
    (96, 192, (5, 33), (1, 1), (0, 0)),
])
def test_conv_glu_fused(case):
    """GLU folded into the GEMM store (ops.conv2d_glu) vs F.glu(F.conv2d(...)): forward and all gradients."""
    from remfx_amd import ops
    Cin, C2, (IA, IB), (KA, KB), pad = case
    torch.manual_seed(3)
    x = torch.randn(2, Cin, IA, IB)
    w = torch.randn(C2, Cin, KA, KB) / (Cin * KA * KB) ** 0.5
    b = torch.randn(C2)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    ref = F.glu(F.conv2d(xr, wr, br, 1, pad), 1)
    g = torch.randn_like(ref)
    ref.backward(g)
    xd, wd, bd = (t.clone().cuda().requires_grad_(True) for t in (x, w, b))
    out = ops.conv2d_glu(xd, wd, bd, (1, 1), pad)
    out.backward(g.cuda())
    check(_rms(out.detach().cpu(), ref.detach()), 1e-5)
    for a, r in ((xd.grad, xr.grad), (wd.grad, wr.grad), (bd.grad, br.grad)):
        check(_rms(a.cpu(), r), 1e-4, max(1e-3, float(r.abs().max())))


@pytest.mark.parametrize("case", [
    # Cin, Cout, (IA, IB), (KA, KB), stride, padding
    (48, 12, (1, 300), (1, 3), (1, 1), (0, 1)),      # unit stride: residual in the plain store
    (16, 24, (1, 48), (1, 8), (1, 4), (0, 2)),       # merged-phase input gradient (time branch encoder)
    (24, 40, (32, 9), (8, 1), (4, 1), (2, 0)),       # merged on the A axis (freq branch encoder)
])
def test_conv_fork_residual_gradient(case):
    """ops.conv2d_fork: conv(x) plus an alias of x; the alias' gradient is added inside the input-gradient GEMM."""
    from remfx_amd import ops
    Cin, Cout, (IA, IB), (KA, KB), stride, padding = case
    torch.manual_seed(4)
    x = torch.randn(2, Cin, IA, IB)
    w = torch.randn(Cout, Cin, KA, KB) / (Cin * KA * KB) ** 0.5
    b = torch.randn(Cout)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr, br, stride, padding)
    g1, g2 = torch.randn(yr.shape), torch.randn(x.shape)
    (yr * g1).sum().backward(retain_graph=True)
    (xr * g2).sum().backward()
    dev = _dev()
    xd, wd, bd = (t.to(dev).requires_grad_(True) for t in (x, w, b))
    y, alias = ops.conv2d_fork(xd, wd, bd, stride, padding)
    ((y * g1.to(dev)).sum() + (alias * g2.to(dev)).sum()).backward()
    check(_rms(y.detach().cpu(), yr.detach()), 1e-5)
    for got, ref in ((xd.grad, xr.grad), (wd.grad, wr.grad), (bd.grad, br.grad)):
        check(_rms(got.cpu(), ref), 2e-5, max(1.0, float(ref.abs().max())))


@pytest.mark.parametrize("case", [(3, 16, 12, 256, 1, False), (3, 16, 12, 256, 3, False), (3, 48, 12, 256, 3, True),
                                  (2, 16, 40, 512, 1, False), (2, 16, 44, 512, 1, True), (2, 24, 96, 65536, 1, True)])
def test_partial_channel_tile_stays_inside_the_tensor(case):
    """The lean full-tile store (csrc/gemm_fwd.h fwd_store_fast_plain) lets the HARDWARE drop the rows >= M of a partial channel
    tile: per-row offsets ride in the scalar offset of a raw buffer store whose num_records = M * channel stride.  If the range
    check ignored the scalar offset those rows would land in the next sample / behind the tensor: guard regions either side of the
    output must stay untouched and every sample must match torch."""
    from remfx_amd import ops
    N, Cin, Cout, T, k, out16 = case
    torch.manual_seed(0)
    x = torch.randn(N, Cin, 1, T, device="cuda")
    w = torch.randn(Cout, Cin, 1, k, device="cuda") * 0.1
    b = torch.randn(Cout, device="cuda")
    dt = torch.bfloat16 if (out16 and mode() == "bf16") else torch.float32
    n_out, guard = N * Cout * T, 4096
    buf = torch.full((n_out + 2 * guard,), 7.0, device="cuda", dtype=dt)
    out = buf[guard:guard + n_out].view(N, Cout, 1, T)
    y = ops.conv2d_forward(x, w, b, (1, 1), (0, k // 2), (1, 1), out=out)
    torch.cuda.synchronize()
    assert bool((buf[:guard] == 7.0).all()) and bool((buf[guard + n_out:] == 7.0).all()), "store outside the output tensor"
    ref = F.conv2d(x, w, b, padding=(0, k // 2))
    err = float((y.float() - ref).abs().max())
    check(err, 2e-5, 1.0, bf16x3=2e-4, bf16=6e-2, what=("partial channel tile vs torch", case))


@pytest.mark.one_mode
def test_pack_cache_follows_weight_updates():
    """ops.pack_cached: a layer's packed weights are rebuilt after an in-place torch update (version counter), after a native
    optimiser step (ops.weights_changed via FlatAdamW.step) and for a new tensor at a recycled address; otherwise re-used."""
    from remfx_amd import ops
    from remfx_amd.optim import FlatParams, FlatAdamW
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 16, 1, 256, generator=g).to(dev)
    conv = torch.nn.Conv2d(16, 32, (1, 3), padding=(0, 1)).to(dev)
    ct = torch.nn.ConvTranspose2d(16, 8, (1, 8), stride=(1, 4)).to(dev)

    def run():
        y = ops.conv2d(x, conv.weight, conv.bias, (1, 1), (0, 1), (1, 1))
        z = ops.conv_transpose2d(x, ct.weight, ct.bias, (1, 4), (1, 1), (0, 0), (1, 1024))
        return y.detach().clone(), z.detach().clone()

    def ref():
        prev = ops.PACK_CACHE
        ops.PACK_CACHE = False
        try:
            return run()
        finally:
            ops.PACK_CACHE = prev

    ops.clear_pack_cache()
    a = run()
    n0 = len(ops._PACKS)
    assert n0 >= 2
    b = run()
    assert len(ops._PACKS) == n0 and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    with torch.no_grad():                                    # torch in-place update: version counter
        conv.weight.mul_(1.5)
        ct.weight.add_(0.25)
    c, r = run(), ref()
    assert torch.equal(c[0], r[0]) and torch.equal(c[1], r[1]) and not torch.equal(c[0], a[0]) and not torch.equal(c[1], a[1])
    # native optimiser step (raw-pointer kernel): parameters move into the flat buffer, then change without a version bump
    mods = torch.nn.ModuleList([conv, ct])
    opt = FlatAdamW(FlatParams(list(mods.parameters())), lr=1e-2)
    d = run()
    opt.zero_grad()
    ya = ops.conv2d(x, conv.weight, conv.bias, (1, 1), (0, 1), (1, 1)).square().mean() + \
        ops.conv_transpose2d(x, ct.weight, ct.bias, (1, 4), (1, 1), (0, 0), (1, 1024)).square().mean()
    ya.backward()
    opt.step()
    e, r = run(), ref()
    assert torch.equal(e[0], r[0]) and torch.equal(e[1], r[1]) and not torch.equal(e[0], d[0]) and not torch.equal(e[1], d[1])

"""GPU: channels-last bf16 kernels (csrc/cl_conv.hip, cl_elem.hip) against fp64 torch convolutions of the SAME bf16-rounded
operands (what the MFMA computes up to fp32 accumulation order).  Layer forms are those of torchaudio HDemucs' frequency branch
(`_HEncLayer.conv / rewrite`, `_HDecLayer.rewrite / conv_tr`; reference call site remfx/models.py:308,317)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = [pytest.mark.gpu, pytest.mark.one_mode]
DEV = "cuda:0"


def _bf(t):
    return t.to(torch.bfloat16)


def _r(t):                       # bf16 round trip, fp64 result
    return t.to(torch.bfloat16).to(torch.float64)


def _cl(x_cm):                   # (N, C, A, B) cpu -> (N, A, B, C) bf16 gpu
    return x_cm.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(DEV)


def _cm(x_cl):                   # (N, A, B, C) bf16 gpu -> (N, C, A, B) fp64 cpu
    return x_cl.detach().cpu().to(torch.float64).permute(0, 3, 1, 2).contiguous()


def _close(got, ref, what, ulps=2.0, mag=None):
    """bf16 results: within `ulps` bf16 ulps of the fp64 reference (rounding + accumulation order), rms far below one ulp.
    mag: magnitude the ulp is taken of where the reference is a SUM whose terms cancel (|a| + |b| of `a + b`)."""
    ref = ref.to(torch.float64)
    got = got.to(torch.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    scale = (ref.abs() if mag is None else mag.to(torch.float64)).clamp_min(float(ref.abs().mean()) * 1e-2 + 1e-30)
    err = (got - ref).abs() / scale
    rms = float(((got - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt().clamp_min(1e-30))
    print(f"{what}: max rel err {float(err.max()):.3e} (bf16 ulp 7.8e-3), rel rms {rms:.3e}")
    assert float(err.max()) < ulps * 7.9e-3, what
    assert rms < 3.5e-3, what


def test_layout_round_trip():
    from remfx_amd import clast
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 48, 3, 128, generator=g)
    xc = clast.from_cm(x.to(DEV))
    assert torch.equal(xc.cpu(), x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16))
    back = clast.to_cm(xc)
    assert torch.equal(back.cpu(), x.to(torch.bfloat16).float())
    xb = x.to(torch.bfloat16).to(DEV)
    assert torch.equal(clast.from_cm(xb).cpu(), x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16))
    assert torch.equal(clast.to_cm(xc, dtype=torch.bfloat16).cpu(), x.to(torch.bfloat16))


@pytest.mark.parametrize("Cin,C2,KA,A,N", [(48, 96, 3, 5, 2), (96, 192, 3, 3, 1), (192, 384, 3, 2, 1), (48, 96, 1, 4, 2), (32, 48, 3, 2, 1)])
def test_conv_glu_forward(Cin, C2, KA, A, N):
    """rewrite conv (3x3 / 1x1) + GLU: z in natural channel order and a * sigmoid(b)."""
    from remfx_amd import clast
    g = torch.Generator().manual_seed(1)
    B = 256
    x = torch.randn(N, Cin, A, B, generator=g)
    w = torch.randn(C2, Cin, KA, KA, generator=g) / (Cin * KA * KA) ** 0.5
    b = torch.randn(C2, generator=g) * 0.1
    form = clast.form_conv_glu(C2, Cin, KA, KA)
    ap = clast.pack(form, w.to(DEV))
    xc = _cl(x)
    z = clast.empty(N, A, B, C2, DEV)
    y = clast.empty(N, A, B, C2 // 2, DEV)
    clast.conv(form, ap, xc, N, A, B, A, "glu", bias=b.to(DEV), out0=z, out1=y)
    torch.cuda.synchronize()
    zr = F.conv2d(_r(x), _r(w), b.double(), padding=KA // 2)
    _close(_cm(z), zr, "z")
    zq = _r(zr)
    yr = zq[:, :C2 // 2] * torch.sigmoid(zq[:, C2 // 2:])
    _close(_cm(y), yr, "glu", ulps=3.0)
    # inference form: no z
    y2 = clast.empty(N, A, B, C2 // 2, DEV)
    clast.conv(form, ap, xc, N, A, B, A, "glu", bias=b.to(DEV), out1=y2)
    assert torch.equal(y2, y)


@pytest.mark.parametrize("Cin,Cout,KA,A,N", [(96, 48, 3, 5, 2), (192, 96, 3, 3, 1), (384, 192, 3, 2, 1), (96, 48, 1, 3, 1)])
def test_conv_dgrad_dgelu(Cin, Cout, KA, A, N):
    """input gradient of a stride-1 conv (rows = the conv's input channels) + the backward of `gelu(z) + skip`."""
    from remfx_amd import clast
    g = torch.Generator().manual_seed(2)
    B = 256
    dz = torch.randn(N, Cin, A, B, generator=g)              # gradient of the conv output (Cin = conv's output channels)
    w = torch.randn(Cin, Cout, KA, KA, generator=g) / (Cin * KA * KA) ** 0.5    # conv weight (out = Cin, in = Cout)
    zprev = torch.randn(N, Cout, A, B, generator=g)
    form = clast.form_conv_dgrad(Cin, Cout, KA, KA)
    ap = clast.pack(form, w.to(DEV))
    dx = clast.empty(N, A, B, Cout, DEV)
    dzp = clast.empty(N, A, B, Cout, DEV)
    clast.conv(form, ap, _cl(dz), N, A, B, A, "dgelu", out0=dx, out1=dzp, aux0=_cl(zprev))
    torch.cuda.synchronize()
    ref = F.conv_transpose2d(_r(dz), _r(w), padding=KA // 2)
    _close(_cm(dx), ref, "dx")
    zp = _r(zprev).requires_grad_(True)
    F.gelu(zp).backward(_r(ref))
    _close(_cm(dzp), zp.grad, "dx * gelu'", ulps=3.0)
    # plain store + residual gradient
    res = torch.randn(N, Cout, A, B, generator=g)
    dx2 = clast.empty(N, A, B, Cout, DEV)
    clast.conv(form, ap, _cl(dz), N, A, B, A, "store", out0=dx2, res=_cl(res))
    torch.cuda.synchronize()
    _close(_cm(dx2), _r(ref) + _r(res), "dx + res", ulps=3.0, mag=_r(ref).abs() + _r(res).abs())


@pytest.mark.parametrize("Cin,Cout,IA,N", [(96, 48, 4, 2), (192, 96, 3, 1), (384, 192, 2, 1), (48, 32, 5, 1)])
def test_convtr_forward_and_dgrad(Cin, Cout, IA, N):
    """ConvTranspose2d((8, 1), stride (4, 1)) cropped by 2 rows as one merged GEMM (+ GELU + next skip), and its input gradient (8 row taps,
    stride 4) with the GLU backward of the rewrite conv in front of it."""
    from remfx_amd import clast
    g = torch.Generator().manual_seed(3)
    B = 256
    y = torch.randn(N, Cin, IA, B, generator=g)
    w = torch.randn(Cin, Cout, 8, 1, generator=g) / (Cin * 2) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    OAo = 4 * IA
    skip = torch.randn(N, Cout, OAo, B, generator=g)
    form = clast.form_convtr_s4(Cin, Cout)
    ap = clast.pack(form, w.to(DEV))
    z = clast.empty(N, OAo, B, Cout, DEV)
    s = clast.empty(N, OAo, B, Cout, DEV)
    clast.conv(form, ap, _cl(y), N, IA, B, IA + 1, "gelu", bias=b.to(DEV), out0=z, out1=s, aux0=_cl(skip), OAo=OAo)
    torch.cuda.synchronize()
    zr = F.conv_transpose2d(_r(y), _r(w), b.double(), stride=(4, 1))[:, :, 2:2 + OAo]
    _close(_cm(z), zr, "conv_tr z")
    sr = F.gelu(_r(zr)) + _r(skip)
    _close(_cm(s), sr, "gelu(z) + skip", ulps=3.0, mag=F.gelu(_r(zr)).abs() + _r(skip).abs())
    # plain store (last decoder layer has no activation)
    z2 = clast.empty(N, OAo, B, Cout, DEV)
    clast.conv(form, ap, _cl(y), N, IA, B, IA + 1, "store", bias=b.to(DEV), out0=z2, OAo=OAo)
    assert torch.equal(z2, z)
    # input gradient: dy = conv_s4(dz) with the weight read transposed, then GLU backward against the rewrite's stored [a | b]
    dz = torch.randn(N, Cout, OAo, B, generator=g)
    zrw = torch.randn(N, 2 * Cin, IA, B, generator=g)
    fd = clast.form_convtr_s4_dgrad(Cin, Cout)
    apd = clast.pack(fd, w.to(DEV))
    dzrw = clast.empty(N, IA, B, 2 * Cin, DEV)
    clast.conv(fd, apd, _cl(dz), N, OAo, B, IA, "dglu", out0=dzrw, aux0=_cl(zrw))
    dy_plain = clast.empty(N, IA, B, Cin, DEV)
    clast.conv(fd, apd, _cl(dz), N, OAo, B, IA, "store", out0=dy_plain)
    torch.cuda.synchronize()
    yv = _r(y).requires_grad_(True)
    zfull = F.conv_transpose2d(yv, _r(w), None, stride=(4, 1))[:, :, 2:2 + OAo]
    zfull.backward(_r(dz))
    _close(_cm(dy_plain), yv.grad, "conv_tr dgrad")
    zq = _r(zrw).requires_grad_(True)
    F.glu(zq, dim=1).backward(_r(yv.grad))
    _close(_cm(dzrw), zq.grad, "glu backward", ulps=3.0)


@pytest.mark.parametrize("Cin,Cout,OA,N", [(48, 96, 3, 2), (96, 192, 2, 1), (192, 384, 2, 1), (16, 48, 4, 1)])
def test_conv_s4_forward_and_dgrad(Cin, Cout, OA, N):
    """encoder Conv2d((8, 1), stride (4, 1), padding (2, 0)) + GELU, and its input gradient as a merged 2-row-tap GEMM with the skip gradient
    and the previous layer's GLU backward in the store."""
    from remfx_amd import clast
    g = torch.Generator().manual_seed(4)
    B = 256
    IA = 4 * OA
    x = torch.randn(N, Cin, IA, B, generator=g)
    w = torch.randn(Cout, Cin, 8, 1, generator=g) / (Cin * 8) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    form = clast.form_conv_s4(Cout, Cin)
    ap = clast.pack(form, w.to(DEV))
    z = clast.empty(N, OA, B, Cout, DEV)
    yy = clast.empty(N, OA, B, Cout, DEV)
    clast.conv(form, ap, _cl(x), N, IA, B, OA, "gelu", bias=b.to(DEV), out0=z, out1=yy)
    torch.cuda.synchronize()
    zr = F.conv2d(_r(x), _r(w), b.double(), stride=(4, 1), padding=(2, 0))
    _close(_cm(z), zr, "conv_s4 z")
    _close(_cm(yy), F.gelu(_r(zr)), "gelu", ulps=3.0)
    if Cin % 8 or Cin < 48:
        return
    dz = torch.randn(N, Cout, OA, B, generator=g)
    gskip = torch.randn(N, Cin, IA, B, generator=g)
    zrw = torch.randn(N, 2 * Cin, IA, B, generator=g)
    fd = clast.form_conv_s4_dgrad(Cout, Cin)
    apd = clast.pack(fd, w.to(DEV))
    dzrw = clast.empty(N, IA, B, 2 * Cin, DEV)
    clast.conv(fd, apd, _cl(dz), N, OA, B, OA + 1, "dglu", out0=dzrw, aux0=_cl(zrw), res=_cl(gskip), OAo=IA)
    dx = clast.empty(N, IA, B, Cin, DEV)
    clast.conv(fd, apd, _cl(dz), N, OA, B, OA + 1, "store", out0=dx, OAo=IA)
    torch.cuda.synchronize()
    xv = _r(x).requires_grad_(True)
    F.conv2d(xv, _r(w), None, stride=(4, 1), padding=(2, 0)).backward(_r(dz))
    _close(_cm(dx), xv.grad, "conv_s4 dgrad")
    gy = _r(_r(xv.grad) + _r(gskip))
    zq = _r(zrw).requires_grad_(True)
    F.glu(zq, dim=1).backward(gy)
    _close(_cm(dzrw), zq.grad, "skip add + glu backward", ulps=4.0,
           mag=torch.cat([torch.sigmoid(zq[:, Cin:]), (zq[:, :Cin] * torch.sigmoid(zq[:, Cin:]) * (1 - torch.sigmoid(zq[:, Cin:]))).abs()], 1).detach()
           * (_r(xv.grad).abs() + _r(gskip).abs()).repeat(1, 2, 1, 1))


def _wclose(got, ref, what):
    """fp32 weight gradients of bf16 operands: the products are exact, only the fp32 accumulation order differs."""
    ref = ref.to(torch.float64)
    got = got.detach().cpu().to(torch.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    err = float((got - ref).abs().max() / ref.abs().max())
    rms = float(((got - ref) ** 2).mean().sqrt() / (ref ** 2).mean().sqrt())
    print(f"{what}: max err / max {err:.3e}, rel rms {rms:.3e}")
    assert err < 2e-4 and rms < 2e-5, what


@pytest.mark.parametrize("Cin,Cout,KA,A,N", [(48, 96, 3, 5, 2), (96, 192, 3, 3, 1), (192, 384, 3, 2, 1), (48, 96, 1, 4, 2), (96, 192, 1, 2, 1),
                                              (96, 48, 3, 3, 1), (32, 48, 3, 2, 1)])
def test_wgrad_conv(Cin, Cout, KA, A, N):
    """dW, db of a stride-1 convolution (3x3 / 1x1) from channels-last operands; deterministic (two runs bit-equal)."""
    from remfx_amd import clast
    g = torch.Generator().manual_seed(5)
    B = 256
    x = torch.randn(N, Cin, A, B, generator=g)
    dz = torch.randn(N, Cout, A, B, generator=g)
    form = clast.wform_conv(Cout, Cin, KA, KA)
    dw = torch.full((Cout, Cin, KA, KA), 7.0, device=DEV)
    db = torch.full((Cout,), 7.0, device=DEV)
    xc, gc = _cl(x), _cl(dz)
    clast.wgrad(form, gc, xc, N, A, A, B, dw, db)
    torch.cuda.synchronize()
    w = torch.zeros(Cout, Cin, KA, KA, dtype=torch.float64, requires_grad=True)
    bb = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    F.conv2d(_r(x), w, bb, padding=KA // 2).backward(_r(dz))
    _wclose(dw, w.grad, "dW")
    _wclose(db, bb.grad, "db")
    dw2 = dw.clone()
    db2 = db.clone()
    clast.wgrad(form, gc, xc, N, A, A, B, dw2, db2, accumulate=True)
    torch.cuda.synchronize()
    assert torch.equal(dw2, 2 * dw) and torch.equal(db2, 2 * db)


@pytest.mark.parametrize("Cin,Cout,OA,N", [(48, 96, 3, 2), (96, 192, 5, 1), (192, 384, 2, 1)])
def test_wgrad_conv_s4_and_convtr(Cin, Cout, OA, N):
    """dW of the encoder's Conv2d((8, 1), stride (4, 1), padding (2, 0)) and of the decoder's cropped ConvTranspose2d((8, 1), stride (4, 1))."""
    from remfx_amd import clast
    g = torch.Generator().manual_seed(6)
    B = 256
    IA = 4 * OA
    x = torch.randn(N, Cin, IA, B, generator=g)
    dz = torch.randn(N, Cout, OA, B, generator=g)
    form = clast.wform_conv_s4(Cout, Cin)
    dw = torch.empty(Cout, Cin, 8, 1, device=DEV)
    db = torch.empty(Cout, device=DEV)
    clast.wgrad(form, _cl(dz), _cl(x), N, OA, IA, B, dw, db)
    torch.cuda.synchronize()
    w = torch.zeros(Cout, Cin, 8, 1, dtype=torch.float64, requires_grad=True)
    bb = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    F.conv2d(_r(x), w, bb, stride=(4, 1), padding=(2, 0)).backward(_r(dz))
    _wclose(dw, w.grad, "conv_s4 dW")
    _wclose(db, bb.grad, "conv_s4 db")
    # transposed convolution Cout -> Cin (coarse rows OA -> fine rows 4 OA): P = its input, Q = its output gradient
    y = torch.randn(N, Cout, OA, B, generator=g)
    dzt = torch.randn(N, Cin, IA, B, generator=g)
    ft = clast.wform_convtr_s4(Cout, Cin)
    dwt = torch.empty(Cout, Cin, 8, 1, device=DEV)
    clast.wgrad(ft, _cl(y), _cl(dzt), N, OA, IA, B, dwt)
    torch.cuda.synchronize()
    wt = torch.zeros(Cout, Cin, 8, 1, dtype=torch.float64, requires_grad=True)
    F.conv_transpose2d(_r(y), wt, None, stride=(4, 1))[:, :, 2:2 + IA].backward(_r(dzt))
    _wclose(dwt, wt.grad, "conv_tr dW")


@pytest.mark.parametrize("Cin,Cout,L,N", [(48, 96, 1024, 2), (96, 192, 2048, 1)])
def test_time_branch_folded_forms(Cin, Cout, L, N):
    """The time branch's Conv1d(8, stride 4, padding 2) / cropped ConvTranspose1d(8, stride 4) through folded views (four consecutive
    positions = 4 C channels of a folded position; a 3-tap stride-1 convolution with structural zeros): forward, input gradients,
    weight gradients against torch conv1d / conv_transpose1d in fp64 on the bf16-rounded operands."""
    from remfx_amd import clast
    g = torch.Generator().manual_seed(7)
    Lo = L // 4
    fold = lambda t: t.view(t.shape[0], 1, t.shape[2] // 4, 4 * t.shape[3])
    cl1 = lambda t: t.permute(0, 2, 1).contiguous().to(torch.bfloat16).to(DEV).unsqueeze(1)            # (N, C, L) -> (N, 1, L, C)
    cm1 = lambda t: t.detach().cpu().to(torch.float64).squeeze(1).permute(0, 2, 1).contiguous()
    # ---- encoder conv + GELU
    x = torch.randn(N, Cin, L, generator=g)
    w = torch.randn(Cout, Cin, 8, generator=g) / (Cin * 8) ** 0.5
    b = torch.randn(Cout, generator=g) * 0.1
    f = clast.form_conv_s4_fold(Cout, Cin)
    ap = clast.pack(f, w.to(DEV))
    xc = cl1(x)
    z = clast.empty(N, 1, Lo, Cout, DEV)
    y = clast.empty(N, 1, Lo, Cout, DEV)
    clast.conv(f, ap, fold(xc), N, 1, Lo, 1, "gelu", bias=b.to(DEV), out0=z, out1=y)
    torch.cuda.synchronize()
    zr = F.conv1d(_r(x), _r(w), b.double(), stride=4, padding=2)
    _close(cm1(z), zr, "conv1d s4 z")
    _close(cm1(y), F.gelu(_r(zr)), "gelu", ulps=3.0)
    # ---- its input gradient (+ skip gradient), written through the folded view
    dz = torch.randn(N, Cout, Lo, generator=g)
    gsk = torch.randn(N, Cin, L, generator=g)
    fd = clast.form_conv_s4_fold_dgrad(Cout, Cin)
    apd = clast.pack(fd, w.to(DEV))
    dx = clast.empty(N, 1, L, Cin, DEV)
    clast.conv(fd, apd, cl1(dz), N, 1, Lo, 1, "store", out0=fold(dx), res=fold(cl1(gsk)))
    torch.cuda.synchronize()
    xv = _r(x).requires_grad_(True)
    F.conv1d(xv, _r(w), None, stride=4, padding=2).backward(_r(dz))
    _close(cm1(dx), _r(xv.grad) + _r(gsk), "conv1d s4 dgrad + skip", ulps=3.0, mag=_r(xv.grad).abs() + _r(gsk).abs())
    # ---- its weight gradient
    wf = clast.wform_conv_s4_fold(Cout, Cin)
    dw = torch.empty(Cout, Cin, 8, device=DEV)
    db = torch.empty(Cout, device=DEV)
    clast.wgrad(wf, cl1(dz), fold(xc), N, 1, 1, Lo, dw, db)
    torch.cuda.synchronize()
    wv = torch.zeros(Cout, Cin, 8, dtype=torch.float64, requires_grad=True)
    bv = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    F.conv1d(_r(x), wv, bv, stride=4, padding=2).backward(_r(dz))
    _wclose(dw, wv.grad, "conv1d s4 dW")
    _wclose(db, bv.grad, "conv1d s4 db")
    # ---- decoder transposed conv Cout -> Cin (coarse Lo -> fine L), GELU + next skip
    yy = torch.randn(N, Cout, Lo, generator=g)
    wt = torch.randn(Cout, Cin, 8, generator=g) / (Cout * 2) ** 0.5
    bt = torch.randn(Cin, generator=g) * 0.1
    skip = torch.randn(N, Cin, L, generator=g)
    ft = clast.form_convtr_fold(Cout, Cin)
    apt = clast.pack(ft, wt.to(DEV))
    zt = clast.empty(N, 1, L, Cin, DEV)
    st = clast.empty(N, 1, L, Cin, DEV)
    clast.conv(ft, apt, cl1(yy), N, 1, Lo, 1, "gelu", bias=bt.to(DEV), out0=fold(zt), out1=fold(st), aux0=fold(cl1(skip)))
    torch.cuda.synchronize()
    ztr = F.conv_transpose1d(_r(yy), _r(wt), bt.double(), stride=4)[:, :, 2:2 + L]
    _close(cm1(zt), ztr, "conv_transpose1d z")
    _close(cm1(st), F.gelu(_r(ztr)) + _r(skip), "gelu + skip", ulps=3.0, mag=F.gelu(_r(ztr)).abs() + _r(skip).abs())
    # ---- its input gradient + GLU backward, and weight gradient
    dzt = torch.randn(N, Cin, L, generator=g)
    zab = torch.randn(N, 2 * Cout, Lo, generator=g)
    ftd = clast.form_convtr_fold_dgrad(Cout, Cin)
    aptd = clast.pack(ftd, wt.to(DEV))
    dzab = clast.empty(N, 1, Lo, 2 * Cout, DEV)
    clast.conv(ftd, aptd, fold(cl1(dzt)), N, 1, Lo, 1, "dglu", out0=dzab, aux0=cl1(zab))
    dyp = clast.empty(N, 1, Lo, Cout, DEV)
    clast.conv(ftd, aptd, fold(cl1(dzt)), N, 1, Lo, 1, "store", out0=dyp)
    torch.cuda.synchronize()
    yv = _r(yy).requires_grad_(True)
    F.conv_transpose1d(yv, _r(wt), None, stride=4)[:, :, 2:2 + L].backward(_r(dzt))
    _close(cm1(dyp), yv.grad, "conv_transpose1d dgrad")
    zq = _r(zab).requires_grad_(True)
    F.glu(zq, dim=1).backward(_r(yv.grad))
    _close(cm1(dzab), zq.grad, "glu backward", ulps=3.0)
    # standalone GLU backward kernel
    d2 = clast.dglu(dyp, cl1(zab))
    zq2 = _r(zab).requires_grad_(True)
    F.glu(zq2, dim=1).backward(_r(cm1(dyp)))
    _close(cm1(d2), zq2.grad, "dglu kernel", ulps=3.0)
    wft = clast.wform_convtr_fold(Cout, Cin)
    dwt = torch.empty(Cout, Cin, 8, device=DEV)
    clast.wgrad(wft, cl1(yy), fold(cl1(dzt)), N, 1, 1, Lo, dwt)
    torch.cuda.synchronize()
    wtv = torch.zeros(Cout, Cin, 8, dtype=torch.float64, requires_grad=True)
    F.conv_transpose1d(_r(yy), wtv, None, stride=4)[:, :, 2:2 + L].backward(_r(dzt))
    _wclose(dwt, wtv.grad, "conv_transpose1d dW")

"""GPU parity: framed-FFT kernels (STFT / iSTFT fwd+bwd) and the loss kernels vs
torch.stft / torch.istft on CPU and the CPU oracle (oracle/ref_losses.py)."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

GEOMS = [  # n_fft, hop, win   
    (4096, 1024, 4096), (2048, 512, 2048), (1024, 120, 600), (2048, 240, 1200), (512, 50, 240),
    (512, 128, 512),
]


def _rms(a, b):
    return float(((a - b) ** 2).mean().sqrt())


@pytest.mark.parametrize("geom", GEOMS)
def test_stft_forward_modes(geom):
    from remfx_amd import stft
    n_fft, hop, win = geom
    g = torch.Generator().manual_seed(n_fft + hop)
    x = torch.randn(3, 20000, generator=g)
    ref = torch.stft(x, n_fft, hop, win, torch.hann_window(win), return_complex=True)
    xd = x.to(DEV)
    got = stft.stft(xd, n_fft, hop, win, mode="complex").cpu()
    scale = float(ref.abs().max())
    assert got.shape == tuple(ref.shape) + (2,)
    assert _rms(got, torch.view_as_real(ref)) < 2e-6 * scale
    cac = stft.stft(xd, n_fft, hop, win, mode="cac").cpu()
    assert _rms(cac[:, 0], ref.real) < 2e-6 * scale and _rms(cac[:, 1], ref.imag) < 2e-6 * scale
    mag = stft.stft(xd, n_fft, hop, win, mode="mag", eps=1e-8).cpu()
    assert _rms(mag, torch.sqrt(torch.clamp(ref.real ** 2 + ref.imag ** 2, min=1e-8))) < 2e-6 * scale
    pw = stft.stft(xd, n_fft, hop, win, mode="pow").cpu()
    assert _rms(pw, ref.real ** 2 + ref.imag ** 2) < 4e-6 * scale * scale
    mp = stft.stft(xd, n_fft, hop, win, mode="magpow", eps=1e-8, alpha=0.3).cpu()
    assert _rms(mp, (ref.abs() + 1e-8) ** 0.3) < 1e-5


@pytest.mark.parametrize("geom", GEOMS)
def test_stft_frame_major(geom):
    """complex_fm = the complex spectrum with (bins, frames) swapped, forward and adjoint (the MR-STFT loss's layout)."""
    from remfx_amd import stft
    n_fft, hop, win = geom
    g = torch.Generator().manual_seed(n_fft + 3 * hop)
    x = torch.randn(3, 30011, generator=g).to(DEV)
    xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    a = stft.stft(xa, n_fft, hop, win, mode="complex")
    b = stft.stft(xb, n_fft, hop, win, mode="complex_fm")
    assert b.shape == (a.shape[0], a.shape[2], a.shape[1], 2)
    assert torch.equal(a.permute(0, 2, 1, 3), b)
    gy = torch.randn(a.shape, generator=g).to(DEV)
    a.backward(gy)
    b.backward(gy.permute(0, 2, 1, 3).contiguous())
    assert _rms(xa.grad.cpu(), xb.grad.cpu()) < 1e-6 * float(xa.grad.abs().max())


@pytest.mark.parametrize("geom", GEOMS)
def test_stft_backward(geom):
    from remfx_amd import stft
    n_fft, hop, win = geom
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 9000, generator=g)
    xr = x.clone().requires_grad_(True)
    ref = torch.view_as_real(torch.stft(xr, n_fft, hop, win, torch.hann_window(win), return_complex=True))
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    xd = x.to(DEV).requires_grad_(True)
    got = stft.stft(xd, n_fft, hop, win, mode="complex")
    got.backward(gy.to(DEV))
    scale = float(xr.grad.abs().max())
    assert _rms(xd.grad.cpu(), xr.grad) < 5e-6 * scale


def test_hdemucs_spec_ispec():
    """HDemucs _spec (extra reflect pad, normalized, drop Nyquist, frames [2:2+le]) and _ispec."""
    from oracle.ref_hdemucs import HDemucs
    from remfx_amd import stft
    m = HDemucs(sources=["mixture"], audio_channels=1, nfft=4096, channels=4, depth=6)
    T = 30000
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 1, T, generator=g)
    xr = x.clone().requires_grad_(True)
    z = m._spec(xr)                                  # (2,1,2048,le) complex
    le = z.shape[-1]
    zr = torch.view_as_real(z).permute(0, 1, 4, 2, 3).reshape(2, 2, 2048, le)
    gz = torch.randn(zr.shape, generator=g)
    zr.backward(gz)
    hl = 1024
    pad = hl // 2 * 3
    xd = x.to(DEV).reshape(2, T).requires_grad_(True)
    cac = stft.stft(xd, 4096, hl, mode="cac", normalized=True, bins=2048, frame0=2, frames_out=le,
                    extra_pad=(pad, pad + le * hl - T))
    assert cac.shape == (2, 2, 2048, le)
    assert _rms(cac.detach().cpu(), zr.detach()) < 2e-6 * float(zr.abs().max())
    cac.backward(gz.to(DEV))
    assert _rms(xd.grad.cpu().view_as(xr.grad), xr.grad) < 5e-6 * float(xr.grad.abs().max())
    # inverse
    zin = torch.randn(2, 1, 2048, le, 2, generator=g)
    zc = torch.view_as_complex(zin.clone()).requires_grad_(True)
    xo = m._ispec(zc, T)                             # (2,1,T)
    go = torch.randn(xo.shape, generator=g)
    xo.backward(go)
    cin = zin[:, 0].permute(0, 3, 1, 2).contiguous().to(DEV).requires_grad_(True)   # (2,2,2048,le)
    got = stft.istft(cin, 4096, hl, mode="cac", normalized=True, frames=le + 4, frame0=2, crop=pad, length=T)
    assert got.shape == (2, T)
    assert _rms(got.detach().cpu(), xo.detach()[:, 0]) < 2e-6 * float(xo.abs().max())
    got.backward(go[:, 0].to(DEV))
    gref = torch.view_as_real(zc.grad)[:, 0].permute(0, 3, 1, 2)
    assert _rms(cin.grad.cpu(), gref) < 5e-6 * float(gref.abs().max())


def test_istft_plain():
    from remfx_amd import stft
    g = torch.Generator().manual_seed(4)
    T = 16384
    x = torch.randn(2, T, generator=g)
    spec = torch.stft(x, 2048, 512, window=torch.hann_window(2048), return_complex=True)
    ref = torch.istft(spec, 2048, 512, window=torch.hann_window(2048), length=T)
    got = stft.istft(torch.view_as_real(spec).contiguous().to(DEV), 2048, 512, mode="complex", length=T)
    assert _rms(got.cpu(), ref) < 2e-6 * float(ref.abs().max())
    assert _rms(got.cpu(), x) < 1e-5            # perfect reconstruction


def test_mrstft_l1_sisdr_vs_oracle():
    from oracle import ref_losses
    from remfx_amd import losses
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 1, 24000, generator=g) * 0.3
    y = x + 0.1 * torch.randn(3, 1, 24000, generator=g)
    xr = x.clone().requires_grad_(True)
    lref = ref_losses.mrstft_loss(xr, y) + 100.0 * ref_losses.l1_loss(xr, y)
    lref.backward()
    xd = x.to(DEV).requires_grad_(True)
    yd = y.to(DEV)
    mr, l1 = losses.MultiResolutionSTFTLoss(n_bins=1025, sample_rate=48000), losses.L1Loss()
    l = mr(xd, yd) + l1(xd, yd) * 100
    l.backward()
    assert abs(float(l) - float(lref)) < 1e-4 * abs(float(lref))
    assert _rms(xd.grad.cpu(), xr.grad) < 1e-4 * float(xr.grad.abs().max())
    # whole-batch spectral convergence variant (older auraloss)
    l2 = losses.MultiResolutionSTFTLoss(per_example_sc=False)(xd, yd)
    assert abs(float(l2) - float(ref_losses.mrstft_loss(x, y, per_example_sc=False))) < 1e-4 * float(l2)
    s = losses.SISDRLoss()(xd.detach(), yd)
    assert abs(float(s) - float(ref_losses.sisdr_loss(x, y))) < 1e-3
    # strided (cropped) inputs
    s2 = losses.SISDRLoss()(xd.detach()[..., 5:20000], yd[..., 5:20000])
    assert abs(float(s2) - float(ref_losses.sisdr_loss(x[..., 5:20000], y[..., 5:20000]))) < 1e-3


@pytest.mark.parametrize("geom", [(1024, 120, 600), (2048, 240, 1200), (512, 50, 240), (512, 128, 512)])
def test_pair_loss_kernel(geom):
    """rfx_stft_pair_loss (both spectra of a frame from one complex FFT, sums in the epilogue) against the two-analysis + reduction
    path: row sums, the stored prediction spectrum and the clamped target magnitudes, edge (reflected) frames included."""
    import ctypes as C
    from remfx_amd import _lib, losses, stft
    from remfx_amd.ops import _ptr, _stream
    n_fft, hop, win = geom
    g = torch.Generator().manual_seed(n_fft + hop)
    R, L = 5, 30011
    x = (torch.randn(R, L, generator=g) * 0.3).to(DEV)
    y = (x + 0.1 * torch.randn(R, L, generator=g).to(DEV)).contiguous()
    w = stft.hann(win, DEV)
    X = stft.stft_raw(x, n_fft, hop, win, w, 5)
    Y = stft.stft_raw(y, n_fft, hop, win, w, 5)
    n = X.shape[1] * X.shape[2]
    ref = torch.full((R, 3), float("nan"), device=DEV)                # written, not accumulated: no zero fill
    ws = torch.empty(3 * R * 64, device=DEV, dtype=torch.float64)
    _lib.check(_lib.lib().rfx_stft_loss_reduce(_ptr(X), _ptr(Y), R, n, 1e-8, _ptr(ws), _ptr(ref), _stream()), "reduce")
    sums, Xp, ym = losses._pair_sums(x, y, n_fft, hop, win, w, 1e-8, True)
    assert Xp.shape == X.shape and ym.shape == X.shape[:3]
    scale = float(X.abs().max())
    assert _rms(Xp.cpu(), X.cpu()) < 2e-6 * scale
    assert _rms(ym.cpu(), torch.sqrt(torch.clamp(Y[..., 0] ** 2 + Y[..., 1] ** 2, min=1e-8)).cpu()) < 2e-6 * scale
    assert torch.allclose(sums.cpu(), ref.cpu(), rtol=2e-5, atol=0)
    s2, X2, _ = losses._pair_sums(x, y, n_fft, hop, win, w, 1e-8, False)
    assert X2 is None and torch.allclose(s2.cpu(), ref.cpu(), rtol=2e-5, atol=0)
    assert torch.equal(s2, sums)                          # per-workgroup slots added in order: bit-reproducible


@pytest.mark.parametrize("case", [(1024, 120, 600, 5, 30011), (2048, 240, 1200, 3, 20000), (512, 50, 240, 9, 7001)])
def test_loss_gradient_inside_synthesis(case, monkeypatch):
    """rfx_fft_synthesis_lossgrad (gradient spectrum formed in the inverse transform's merge step, never written) against
    rfx_stft_loss_grad_m + rfx_fft_synthesis: the same per-cell arithmetic, so the two differ only by the order of the overlap-add
    atomics; identical signals still get an exactly zero gradient."""
    from remfx_amd import losses
    n_fft, hop, win, R, L = case
    g = torch.Generator().manual_seed(n_fft + R)
    x = (torch.randn(R, 1, L, generator=g) * 0.3).to(DEV)
    y = (x + 0.1 * torch.randn(R, 1, L, generator=g).to(DEV)).contiguous()
    grads = []
    for fused in (True, False):
        monkeypatch.setattr(losses, "FUSED_GRAD", fused)
        xd = x.clone().requires_grad_(True)
        l = losses.MultiResolutionSTFTLoss(fft_sizes=(n_fft,), hop_sizes=(hop,), win_lengths=(win,))(xd, y)
        (l * 1.7).backward()                      # upstream gradient != 1: the device-side gup path
        grads.append(xd.grad.detach().cpu())
    scale = float(grads[1].abs().max())
    assert scale > 0
    assert _rms(grads[0], grads[1]) < 1e-6 * scale, (_rms(grads[0], grads[1]), scale)
    monkeypatch.setattr(losses, "FUSED_GRAD", True)
    xd = x.clone().requires_grad_(True)
    l = losses.MultiResolutionSTFTLoss(fft_sizes=(n_fft,), hop_sizes=(hop,), win_lengths=(win,))(xd, x.clone())
    l.backward()
    assert float(l) == 0.0 and float(xd.grad.abs().max()) == 0.0


@pytest.mark.parametrize("case", [(512, 75, 301, 1, 1500), (1024, 256, 1024, 9, 4099), (2048, 512, 2048, 2, 2049 + 512),
                                  (512, 128, 400, 17, 700)])
def test_pair_loss_ragged(case):
    """Short / ragged inputs of the one-launch loss: fewer frames than a batch holds, odd hop and window, rows not a multiple of 8,
    every frame a reflected edge frame -- MultiResolutionSTFTLoss value and gradient against the CPU oracle."""
    from oracle import ref_losses
    from remfx_amd import losses
    n_fft, hop, win, R, L = case
    g = torch.Generator().manual_seed(n_fft + R)
    x = torch.randn(R, 1, L, generator=g) * 0.3
    y = x + 0.1 * torch.randn(R, 1, L, generator=g)
    xr = x.clone().requires_grad_(True)
    lref = ref_losses.stft_loss(xr, y, n_fft, hop, win, True)
    lref.backward()
    xd = x.to(DEV).requires_grad_(True)
    l = losses.MultiResolutionSTFTLoss(fft_sizes=(n_fft,), hop_sizes=(hop,), win_lengths=(win,))(xd, y.to(DEV))
    l.backward()
    assert abs(float(l) - float(lref)) < 1e-4 * abs(float(lref))
    assert _rms(xd.grad.cpu(), xr.grad) < 1e-4 * float(xr.grad.abs().max())


def test_rfx_zero():
    from remfx_amd import ops
    for n, off in ((1, 0), (3, 1), (4, 0), (1000003, 3), (1 << 22, 0), (4099, 2)):
        buf = torch.full((n + 8,), 7.0, device=DEV)
        ops.zero_(buf[off:off + n])
        assert float(buf[off:off + n].abs().max()) == 0.0
        assert float(buf[:off].sum()) == 7.0 * off and float(buf[off + n:].sum()) == 7.0 * (8 - off)
    d = ops.zeros((5, 3), DEV, torch.float64)
    assert d.dtype == torch.float64 and float(d.abs().max()) == 0.0


def test_stft_memo_scope():
    """Inside stft_memo() repeated MRSTFT evaluations are bit-identical to the unshared ones, an in-place change of a
    signal is seen (version bump), and nothing survives the scope."""
    from remfx_amd import losses
    g = torch.Generator().manual_seed(6)
    x = (torch.randn(2, 1, 16000, generator=g) * 0.3).to(DEV)
    y = (x + 0.1 * torch.randn(2, 1, 16000, generator=g).to(DEV))
    mr = losses.MultiResolutionSTFTLoss()
    plain = float(mr(x, y))
    xg = x.clone().requires_grad_(True)
    plain_l = mr(xg, y)
    plain_l.backward()
    with losses.stft_memo():
        xm = x.clone().requires_grad_(True)
        l = mr(xm, y)
        # metric(output, target) == loss term on shared spectra (row sums use float atomics: equal to rounding)
        assert abs(float(mr(xm.detach(), y)) - float(l)) < 1e-6 * plain and abs(float(l) - plain) < 1e-6 * plain
        n_entries = len(losses._MEMO)
        assert sum(1 for k in losses._MEMO if k[0] == "pair") == 3     # one (prediction, target) entry per resolution, not 6
        l.backward()
        # (the adjoint STFT overlap-adds with atomics: equal up to summation order)
        assert _rms(xm.grad.cpu(), xg.grad.cpu()) < 1e-5 * float(xg.grad.abs().max())
        y.mul_(0.5)                                                    # version bump -> recomputed
        assert abs(float(mr(x, y)) - plain) > 1e-3 * plain
        y.mul_(2.0)
    assert losses._MEMO is None
    assert abs(float(mr(x, y)) - plain) < 1e-6 * plain


def test_spectrogram_golden(golden_dir):
    import os
    import numpy as np
    from remfx_amd.utils import spectrogram
    gd = np.load(os.path.join(golden_dir, "utils_small.npz"))
    x = torch.from_numpy(gd["x"]).to(DEV)
    S = spectrogram(x, torch.hann_window(512).to(DEV), 512, 128, 0.3).cpu().numpy()
    assert S.shape == gd["spec"].shape
    assert float(np.sqrt(((S - gd["spec"]) ** 2).mean())) < 1e-5


def test_frame_major_ends_kernels():
    """Round 6: the kernels that keep HDemucs' spectrum frame-major on both sides of the U-Net.  (1) `rfx_cl_im2col_fm`: the first
    convolution's 16-channel operand from the frame-major spectrum with the standardisation folded in, against the channel-major
    im2col of the standardised tensor; (2) `rfx_fm_cm_affine` in both directions against permute + affine; (3) the inverse STFT of
    a frame-major spectrum against the same spectrum in torch's [bin][frame] layout (HDemucs `_ispec` geometry)."""
    from remfx_amd import clast, nnops, stft
    g = torch.Generator().manual_seed(5)
    N, F, bins = 3, 96, 72
    spec = torch.randn(N, F, bins, 2, generator=g).to(DEV)
    a = (torch.rand(N, generator=g) + 0.5).to(DEV)
    b = (torch.randn(N, generator=g) * 0.3).to(DEV)
    got = clast.im2col_fm(spec, a, b).float()                                  # (N, bins / 4, F, 16)
    x_cm = (spec * a.view(N, 1, 1, 1) + b.view(N, 1, 1, 1)).permute(0, 3, 2, 1).contiguous()   # (N, 2, bins, F)
    ref = clast.im2col_s4(x_cm, bins // 4, F, False).float()
    assert got.shape == ref.shape == (N, bins // 4, F, 16)
    # fma against mul + add in front of the bf16 rounding: at most one bf16 ulp apart, and zero padding exactly zero
    assert float((got - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    assert float(((got - ref).abs() > 0).float().mean()) < 0.02
    assert float(got[:, 0, :, :4].abs().max()) == 0.0 and float(got[:, -1, :, 12:].abs().max()) == 0.0   # bins -2, -1 and bins, bins + 1
    x = torch.randn(N, 2, bins, F, generator=g).to(DEV).requires_grad_(True)
    y = nnops.cm_to_fm_affine(x, a, b)
    yr = x.detach().permute(0, 3, 2, 1) * a.view(N, 1, 1, 1) + b.view(N, 1, 1, 1)
    assert y.shape == (N, F, bins, 2) and float((y.detach() - yr).abs().max()) <= 1e-6 * float(yr.abs().max())
    gy = torch.randn(N, F, bins, 2, generator=g).to(DEV)
    (gx,) = torch.autograd.grad(y, x, gy)
    gr = gy.permute(0, 3, 2, 1) * a.view(N, 1, 1, 1)
    assert float((gx - gr).abs().max()) <= 1e-6 * float(gr.abs().max())
    # _ispec geometry: 4096 / 1024, Nyquist dropped, frames [2 : 2 + le] of le + 4
    R, le, hl = 2, 64, 1024
    L = le * hl
    cac = (torch.randn(R, 2, 2048, le, generator=g) * 0.1).to(DEV).requires_grad_(True)
    fm = cac.detach().permute(0, 3, 2, 1).contiguous().requires_grad_(True)
    pad = hl // 2 * 3
    o1 = stft.istft(cac, 4096, hl, mode="cac", normalized=True, frames=le + 4, frame0=2, crop=pad, length=L)
    o2 = stft.istft(fm, 4096, hl, mode="complex_fm", normalized=True, frames=le + 4, frame0=2, crop=pad, length=L)
    assert float((o1 - o2).abs().max()) <= 1e-6 * float(o1.abs().max())
    go = torch.randn(R, L, generator=g).to(DEV)
    (g1,) = torch.autograd.grad(o1, cac, go)
    (g2,) = torch.autograd.grad(o2, fm, go)
    assert float((g1 - g2.permute(0, 3, 2, 1)).abs().max()) <= 1e-6 * float(g1.abs().max())

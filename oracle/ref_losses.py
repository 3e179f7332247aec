"""Oracle (test infrastructure): auraloss MultiResolutionSTFTLoss / SISDRLoss
and the reference's loss composition.

auraloss is an un-vendored dependency (setup.py: bare ``auraloss``; absent from
/root/reference and from this image) -> PARITY UNPINNED; this restates the
published algorithm over torch.stft.

Reference call sites: remfx/models.py:35-44,107 (chain inference),
models.py:171-176,227-255 (RemFX metrics), models.py:289-291,299 (UMX),
models.py:312-314,320 (Demucs), models.py:351-353,362 (DCUNet),
models.py:374-376,385 (TCN):  loss = MRSTFT(out, tgt) + 100 * L1(out, tgt).
"""
import torch

MRSTFT_FFT = (1024, 2048, 512)
MRSTFT_HOP = (120, 240, 50)
MRSTFT_WIN = (600, 1200, 240)


def stft_mag(x, n_fft, hop, win, eps=1e-8):
    """x: (R, L) -> sqrt(clamp(re^2+im^2, eps)), shape (R, n_fft/2+1, frames)."""
    X = torch.stft(x, n_fft, hop, win, torch.hann_window(win, dtype=x.dtype),
                   return_complex=True)
    return torch.sqrt(torch.clamp(X.real ** 2 + X.imag ** 2, min=eps))


def stft_loss(inp, tgt, n_fft, hop, win, per_example_sc=True):
    """One resolution: spectral convergence + log-magnitude L1."""
    L = inp.shape[-1]
    xm = stft_mag(inp.reshape(-1, L), n_fft, hop, win)
    ym = stft_mag(tgt.reshape(-1, L), n_fft, hop, win)
    if per_example_sc:      # auraloss >= 0.4: Frobenius norm per example, batch mean
        sc = (torch.linalg.norm(ym - xm, dim=(-2, -1)) /
              torch.linalg.norm(ym, dim=(-2, -1))).mean()
    else:                   # older auraloss: one norm over the whole batch tensor
        sc = torch.linalg.norm((ym - xm).reshape(-1)) / torch.linalg.norm(ym.reshape(-1))
    lm = (torch.log(xm) - torch.log(ym)).abs().mean()
    return sc + lm


def mrstft_loss(inp, tgt, per_example_sc=True):
    """(B, C, L) x2 -> scalar; mean over the three default resolutions."""
    tot = 0.0
    for f, h, w in zip(MRSTFT_FFT, MRSTFT_HOP, MRSTFT_WIN):
        tot = tot + stft_loss(inp, tgt, f, h, w, per_example_sc)
    return tot / len(MRSTFT_FFT)


def l1_loss(inp, tgt):
    return (inp - tgt).abs().mean()


def removal_loss(out, tgt, per_example_sc=True):
    """models.py:299/320/362/385: mrstft + 100 * l1."""
    return mrstft_loss(out, tgt, per_example_sc) + 100.0 * l1_loss(out, tgt)


def sisdr_loss(inp, tgt, eps=1e-8):
    """auraloss.time.SISDRLoss(zero_mean=True, eps=1e-8, reduction='mean');
    returns -SI-SDR (the reference negates it again for logging, models.py:229-233)."""
    inp = inp - inp.mean(-1, keepdim=True)
    tgt = tgt - tgt.mean(-1, keepdim=True)
    alpha = (inp * tgt).sum(-1) / ((tgt ** 2).sum(-1) + eps)
    t = tgt * alpha.unsqueeze(-1)
    res = inp - t
    val = 10.0 * torch.log10((t ** 2).sum(-1) / ((res ** 2).sum(-1) + eps) + eps)
    return -val.mean()

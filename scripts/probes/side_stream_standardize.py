"""Dev probe: nnops.row_standardize on a side stream while the main stream runs the HDemucs spectrogram + its standardisation
(the first thing the time-branch stream did when it forked before the STFT): compare with a serial evaluation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from remfx_amd import nnops, stft

DEV = torch.device("cuda:0")
side = torch.cuda.Stream(priority=-1)
g = torch.Generator().manual_seed(1)
for B in (8, 1):
    x = (torch.randn(B, 1, 262144, generator=g) * 0.1).to(DEV)
    y0, m0, s0 = nnops.row_standardize(x, 1e-5)
    cac0 = stft.stft(x.reshape(B, 262144), 4096, 1024, mode="cac", normalized=True, bins=2048, frame0=2, frames_out=256,
                     extra_pad=(1536, 1536)).clone()
    z0, mz0, sz0 = nnops.row_standardize(cac0.view(B, 2, 2048, 256), 1e-5)
    torch.cuda.synchronize()
    bad = 0
    for it in range(200):
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        x.record_stream(side)
        with torch.cuda.stream(side):
            y, m, s = nnops.row_standardize(x, 1e-5)
        cac = stft.stft(x.reshape(B, 262144), 4096, 1024, mode="cac", normalized=True, bins=2048, frame0=2, frames_out=256,
                        extra_pad=(1536, 1536))
        z, mz, sz = nnops.row_standardize(cac.view(B, 2, 2048, 256), 1e-5)
        main.wait_stream(side)
        torch.cuda.synchronize()
        dm, ds, dy = float((m - m0).abs().max()), float((s - s0).abs().max()), float((y - y0).abs().max())
        dc = float((cac - cac0).abs().max())
        nc = int(((cac - cac0) != 0).sum())
        dz = float((z - z0).abs().max()) + float((mz - mz0).abs().max()) + float((sz - sz0).abs().max())
        if dz > 0:
            print(f"B={B} it {it}: main-stream standardised spectrum differs: {dz:.3e} (mean {float((mz - mz0).abs().max()):.2e}, std {float((sz - sz0).abs().max()):.2e})", flush=True)
        if dm > 0 or ds > 0 or dy > 0 or dc > 0 or dz > 0:
            bad += 1
            if bad <= 5:
                print(f"B={B} it {it}: |dmean| {dm:.3e} |dstd| {ds:.3e} |dy| {dy:.3e} |dcac| {dc:.3e} ({nc} cells; |cac| max {float(cac0.abs().max()):.2e})", flush=True)
    print(f"B={B}: {bad} / 200 differ")

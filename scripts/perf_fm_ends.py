"""Dev tool: the frame-major ends of HDemucs at 64 clips, timed alone (analysis, moments, im2col, affine transpose, synthesis)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remfx_amd import clast, nnops, stft

dev = torch.device("cuda:0")
R, L, hl, le = 64, 262144, 1024, 256
pad = hl // 2 * 3
x = torch.randn(R, L, device=dev)


def timeit(fn, n=50):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


w = stft.hann(4096, dev)
gb = (R * L * 4 + R * 2048 * le * 8) / 1e9
for mode, m in (("cac", 1), ("complex_fm", 5)):
    f = lambda: stft.stft_raw(x, 4096, hl, 4096, w, m, normalized=True, bins=2048, frame0=2, frames_out=le, extra_pad=(pad, pad + le * hl - L))
    us = timeit(f)
    print(f"_spec analysis {mode:11s} {us:7.1f} us  {gb / us * 1e3 / 8:.3f} of 8 TB/s")
spec_fm = stft.stft_raw(x, 4096, hl, 4096, w, 5, normalized=True, bins=2048, frame0=2, frames_out=le, extra_pad=(pad, pad + le * hl - L))
us = timeit(lambda: nnops.row_moments(spec_fm, 1e-5)); print(f"row_moments            {us:7.1f} us")
mean, std, a, b = nnops.row_moments(spec_fm, 1e-5)
us = timeit(lambda: clast.im2col_fm(spec_fm, a, b)); print(f"im2col_fm              {us:7.1f} us  (reads 268 MB, writes 268 MB)")
xcm = torch.randn(R, 2, 2048, le, device=dev)
us = timeit(lambda: clast.im2col_s4(xcm, 512, le, False)); print(f"im2col_s4 (cm)         {us:7.1f} us")
us = timeit(lambda: nnops.row_standardize(xcm, 1e-5)); print(f"row_standardize (cm)   {us:7.1f} us")
us = timeit(lambda: nnops.cm_to_fm_affine(xcm, std, mean)); print(f"cm_to_fm_affine        {us:7.1f} us")
us = timeit(lambda: nnops.row_affine(xcm.reshape(R, -1), std, mean)); print(f"row_affine (cm)        {us:7.1f} us")
fm = nnops.cm_to_fm_affine(xcm, std, mean)
for mode, sp in (("cac", xcm), ("complex_fm", fm)):
    us = timeit(lambda: stft.istft(sp, 4096, hl, mode=mode, normalized=True, frames=le + 4, frame0=2, crop=pad, length=L))
    print(f"_ispec synthesis {mode:11s} {us:7.1f} us  {gb / us * 1e3 / 8:.3f} of 8 TB/s")

"""Dev tool: the framed-FFT launches of one Demucs training step (B = 64 x 262144) timed alone, with the bytes each must move."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remfx_amd import stft

dev = torch.device("cuda:0")
R, L = 64, 262144
x = torch.randn(R, L, device=dev)


def timeit(fn, n=100):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


print("kernel                                   us     GB   TB/s  of 8 TB/s")
# HDemucs _spec: n_fft 4096, hop 1024, extra reflect pad, frames [2 : 2 + 256], Nyquist dropped, complex-as-channels
hl, le = 1024, 256
pad = hl // 2 * 3
w = stft.hann(4096, dev)
f = lambda: stft.stft_raw(x, 4096, hl, 4096, w, 1, normalized=True, bins=2048, frame0=2, frames_out=le,
                          extra_pad=(pad, pad + le * hl - L))
us = timeit(f); gb = (R * L * 4 + R * 2048 * le * 8) / 1e9
print(f"_spec  analysis 4096/1024            {us:8.1f} {gb:6.3f} {gb / us * 1e3:6.2f} {gb / us * 1e3 / 8:6.3f}")
spec = f()
# _ispec
fi = lambda: stft.istft(spec.view(R, 2, 2048, le), 4096, hl, mode="cac", normalized=True, frames=le + 4, frame0=2, crop=pad, length=L)
us = timeit(fi)
print(f"_ispec synthesis 4096/1024           {us:8.1f} {gb:6.3f} {gb / us * 1e3:6.2f} {gb / us * 1e3 / 8:6.3f}   (incl. zero fill + envelope)")
for n_fft, hop, win in ((1024, 120, 600), (2048, 240, 1200), (512, 50, 240)):
    w = stft.hann(win, dev)
    fa = lambda: stft.stft_raw(x, n_fft, hop, win, w, 5)
    us = timeit(fa); X = fa(); gb = (R * L * 4 + X.numel() * 4) / 1e9
    print(f"loss   analysis {n_fft}/{hop}/{win:<5d}        {us:8.1f} {gb:6.3f} {gb / us * 1e3:6.2f} {gb / us * 1e3 / 8:6.3f}")
    g = torch.randn_like(X)
    xs = x.clone().requires_grad_(True)
    Y = stft.stft(xs, n_fft, hop, win, w, mode="complex_fm")
    fs = lambda: torch.autograd.grad(Y, xs, g, retain_graph=True)
    us = timeit(fs)
    print(f"loss   synthesis {n_fft}/{hop}/{win:<5d}       {us:8.1f} {gb:6.3f} {gb / us * 1e3:6.2f} {gb / us * 1e3 / 8:6.3f}   (incl. zero fill)")

from remfx_amd import losses
y = torch.randn(R, L, device=dev)
for n_fft, hop, win in ((1024, 120, 600), (2048, 240, 1200), (512, 50, 240)):
    w = stft.hann(win, dev)
    for store in (False, True):
        us = timeit(lambda: losses._pair_sums(x, y, n_fft, hop, win, w, 1e-8, store))
        print(f"pair loss {n_fft}/{hop}/{win:<5d} store={store!s:5s}   {us:8.1f} us")

"""Probe: 1x1 conv 12 -> 96 over (64, ., 65536) with / without GN statistics and with a padded row pitch."""
import torch
from remfx_amd import ops

dev = "cuda:0"


def timeit(f, n=10):
    f(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for (N, Ci, Co, L) in [(64, 12, 96, 65536), (64, 24, 192, 16384), (64, 48, 96, 65536), (32768, 12, 96, 256)]:
    w = torch.randn(Co, Ci, 1, 1, device=dev)
    b = torch.randn(Co, device=dev)
    x = torch.randn(N, Ci, 1, L, device=dev)
    out = torch.empty(N, Co, 1, L, device=dev)
    t0 = timeit(lambda: ops.conv2d_forward(x, w, b, (1, 1), (0, 0), (1, 1), out=out))
    st = torch.zeros(N, 16, 2, device=dev, dtype=torch.float64)
    t1 = timeit(lambda: ops.conv2d_forward(x, w, b, (1, 1), (0, 0), (1, 1), out=out, stat_sums=st))
    big = torch.empty(N, Co, 1, L + 64, device=dev)
    outp = big[..., :L]
    t2 = timeit(lambda: ops.conv2d_forward(x, w, b, (1, 1), (0, 0), (1, 1), out=outp))
    gb = (N * Co * L + N * Ci * L) * 4 / 1e9
    print(f"N={N} {Ci}->{Co} L={L}: plain {t0:.3f} ms ({gb / t0:.2f} TB/s)  +stats {t1:.3f} ms  padded pitch {t2:.3f} ms")

"""Dev tool: the 3x3 / 3-tap weight-gradient launches of the Demucs step (B = 64, bf16 mode) timed alone."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remfx_amd import ops

dev = torch.device("cuda:0")
ops.set_gemm_precision("bf16")


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


SHAPES = [  # Cin, Cout, A, B, (ka, kb), pad, g16
    (48, 96, 512, 256, (3, 3), (1, 1), True), (96, 192, 128, 256, (3, 3), (1, 1), True), (192, 384, 32, 256, (3, 3), (1, 1), True),
    (384, 768, 8, 256, (3, 3), (1, 1), True), (48, 96, 1, 65536, (1, 3), (0, 1), True), (96, 192, 1, 16384, (1, 3), (0, 1), True),
    (48, 96, 512, 256, (1, 1), (0, 0), True),
]
for Cin, Cout, A, B, ks, pad, g16 in SHAPES:
    x = torch.randn(64, Cin, A, B, device=dev)
    g = torch.randn(64, Cout, A, B, device=dev)
    if g16:
        g = g.to(torch.bfloat16)
    f = lambda: ops.conv2d_wgrad(x, g, (Cout, Cin) + ks, (1, 1), pad, (1, 1), True)
    ms = timeit(f)
    fl = 2.0 * 64 * Cout * (Cin * ks[0] * ks[1] + 1) * A * B
    gb = (x.numel() * 4 + g.numel() * g.element_size()) / 1e9
    print(f"wgrad {Cin:4d}->{Cout:4d} {ks} on ({A},{B})  {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TF/s  {gb / ms:6.2f} TB/s (x + g once)")
    del x, g

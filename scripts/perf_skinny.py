"""Micro-benchmark of the bandwidth-class gather-GEMM launches of the Demucs B=64 step (dev tool): time and HBM rate per shape."""
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


CASES = [  # name, x shape, Cout, kernel (ka, kb), dilation b, in bf16, out bf16
    ("A dconv2 time L0 12->96", (64, 12, 1, 65536), 96, (1, 1), 1, False, True),
    ("A' same, fp32 out", (64, 12, 1, 65536), 96, (1, 1), 1, False, False),
    ("B dconv2 freq L1 24->192", (8192, 24, 1, 256), 192, (1, 1), 1, False, True),
    ("C rewrite 1x1 48->96", (64, 48, 512, 256), 96, (1, 1), 1, False, True),
    ("C' same, fp32 out", (64, 48, 512, 256), 96, (1, 1), 1, False, False),
    ("D dgrad-like 12->48 k3 bf16 in", (32768, 12, 1, 256), 48, (1, 3), 1, True, False),
    ("D' same fp32 in", (32768, 12, 1, 256), 48, (1, 3), 1, False, False),
    ("E dconv1 time 48->12 k3", (64, 48, 1, 65536), 12, (1, 3), 1, False, True),
    ("E' same fp32 out", (64, 48, 1, 65536), 12, (1, 3), 1, False, False),
    ("G 96->12 1x1 bf16 in", (32768, 96, 1, 256), 12, (1, 1), 1, True, False),
    ("H 48->96 k3 time", (64, 48, 1, 65536), 96, (1, 3), 1, False, True),
]
SEL = sys.argv[1] if len(sys.argv) > 1 else None
for name, xs, co, k, dil, in16, out16 in CASES:
    if SEL and not name.startswith(SEL + " "):
        continue
    if in16:
        continue          # conv2d_forward takes fp32 callers only; 16-bit operands arrive through the autograd Functions
    x = torch.randn(xs, device=dev)
    if in16:
        x = x.bfloat16()
    w = torch.randn(co, xs[1], k[0], k[1], device=dev) * 0.05
    b = torch.randn(co, device=dev)
    pad = (0, dil * (k[1] // 2))
    f = lambda: ops.conv2d_forward(x, w, b, (1, 1), pad, (1, dil), out_bf16=out16)
    y = f()
    ms = timeit(f)
    by = x.numel() * x.element_size() + y.numel() * y.element_size()
    print(f"{name:34s} {ms * 1e3:8.1f} us  {by / 1e9:5.2f} GB  {by / ms / 1e9:5.2f} TB/s   out {y.dtype}")
    del x, y

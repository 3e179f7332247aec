"""Micro-benchmark of the gather-GEMM kernels at TCN block shapes (dev tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remfx_amd import ops
from remfx_amd.tcn import TCNBlockFn, tcn_block_forward

dev = torch.device("cuda:0")
B, C, L = int(os.environ.get("B", 4)), 256, 262144


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


for d in (1, 16, 512):
    x = torch.randn(B, C, L, device=dev)
    w1 = torch.randn(C, C, 7, device=dev) * 0.02
    b1 = torch.randn(C, device=dev)
    sl = torch.full((C,), 0.25, device=dev)
    wr = torch.randn(C, C, 1, device=dev) * 0.05
    Lout = L - 6 * d
    flops = 2.0 * B * Lout * C * C * 8
    ms = timeit(lambda: tcn_block_forward(x, w1, b1, sl, wr, d, False))
    print(f"d={d:4d} fused block fwd   {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s")
    xg = x.clone().requires_grad_(True)
    ps = [t.clone().requires_grad_(True) for t in (w1, b1, sl, wr)]
    y = TCNBlockFn.apply(xg, *ps, d, False)
    g = torch.randn_like(y)
    ms = timeit(lambda: torch.autograd.grad(y, [xg] + ps, g, retain_graph=True), n=3)
    print(f"d={d:4d} block bwd (3 GEMM) {ms:8.3f} ms  {3 * flops / ms / 1e9:7.1f} TFLOP/s")
    g4, x4 = g.unsqueeze(2), x.unsqueeze(2)
    ms = timeit(lambda: ops.conv2d_wgrad(x4, g4, (C, C, 1, 7), (1, 1), (0, 0), (1, d), True), n=3)
    print(f"d={d:4d} wgrad only         {ms:8.3f} ms  {flops * 7 / 8 / ms / 1e9:7.1f} TFLOP/s")
    ms = timeit(lambda: ops.conv2d_dgrad(g, w1.unsqueeze(2), tuple(x4.shape), tuple(x4.stride()), (1, 1), (0, 0), (1, d)) if False else ops.conv2d_dgrad(g4, w1.unsqueeze(2), tuple(x4.shape), tuple(x4.stride()), (1, 1), (0, 0), (1, d)), n=3)
    print(f"d={d:4d} dgrad only         {ms:8.3f} ms  {flops * 7 / 8 / ms / 1e9:7.1f} TFLOP/s")
    del x, xg, y, g

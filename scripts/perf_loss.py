"""Dev tool: time the MRSTFT + L1 loss forward + backward alone (64 clips x 2 channels x 262144 samples).
usage: python scripts/perf_loss.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from remfx_amd import losses, ops

dev = "cuda:0"
ops.set_gemm_precision("bf16")
crit = losses.MultiResolutionSTFTLoss().to(dev)
x = torch.randn(64, 2, 262144, device=dev, requires_grad=True)
y = torch.randn(64, 2, 262144, device=dev)


def step():
    x.grad = None
    l = crit(x, y)
    l.backward()
    return l


for _ in range(3):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    l = step()
e1.record()
torch.cuda.synchronize()
print(f"MRSTFT loss fwd + bwd: {e0.elapsed_time(e1) / 10:.3f} ms  (loss {float(l):.5f})")

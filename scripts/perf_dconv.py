"""Micro-benchmark of the fused DConv layer kernels at the Demucs B=64 frequency-branch shape (N = 32768 x (48, 256))."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remfx_amd import hdemucs, nnops, ops
ops.set_gemm_precision("bf16")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
net = hdemucs._DConv(48, compress=4, depth=2, init=1e-4, attn=False, lstm=False).cuda()
x = torch.randn(N, 48, 256, device="cuda").requires_grad_(True)
g = torch.randn(N, 48, 256, device="cuda")
def run(fused, fused_bwd=True):
    nnops.DCONV_FUSED, nnops.DCONV_FUSED_BWD = fused, fused_bwd
    for _ in range(2):
        y = net(x); y.backward(g)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record(); y = net(x); ev[1].record(); y.backward(g); ev[2].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
for fused, fb in ((True, True), (True, False), (False, False), (True, True)):
    f, b = run(fused, fb)
    print(f"fused fwd={fused} bwd={fb}: DConv (2 layers) forward {f:.2f} ms, backward {b:.2f} ms", flush=True)

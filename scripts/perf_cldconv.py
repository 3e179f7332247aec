"""Dev tool: time the fused channels-last DConv kernels at the Hybrid Demucs layer-0 shape (S = 64 x 512 samples of 256 x 48)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from remfx_amd import cldconv, ops
from remfx_amd.hdemucs import _DConv

DEV = "cuda:0"
ops.set_gemm_precision("bf16")
Bn, A, Cc = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 512, 48
mod = _DConv(Cc, depth=2, init=0.3).to(DEV)
x = (torch.randn(Bn, A, 256, Cc, device=DEV) * 0.5).to(torch.bfloat16).requires_grad_(True)
gy = (torch.randn(Bn, A, 256, Cc, device=DEV) * 0.5).to(torch.bfloat16)
m = list(mod.layers[0])


def ev():
    return torch.cuda.Event(enable_timing=True)


for it in range(3):
    e = [ev() for _ in range(4)]
    e[0].record()
    with torch.no_grad():
        yi = cldconv.dconv_layer(x.detach(), m[0], m[1], m[3], m[4], m[6].scale, 1)
    e[1].record()
    y = cldconv.dconv_layer(x, m[0], m[1], m[3], m[4], m[6].scale, 1)
    e[2].record()
    y.backward(gy)
    e[3].record()
    torch.cuda.synchronize()
    print(f"depth-layer at S = {Bn * A}: forward (inference) {e[0].elapsed_time(e[1]):.3f} ms, forward (training) {e[1].elapsed_time(e[2]):.3f} ms, "
          f"backward (kernel + 2 wgrads) {e[2].elapsed_time(e[3]):.3f} ms", flush=True)

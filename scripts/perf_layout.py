"""Dev tool: time the layout conversion kernels (csrc/cl_elem.hip) at the shapes the Hybrid Demucs step uses (64 clips)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from remfx_amd import clast

DEV = "cuda:0"


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, (Bn, Cc, A, T) in (("freq L1 samples", (64, 96, 128, 256)), ("time L0 samples", (64, 48, 1, 65536)), ("time L1 samples", (64, 96, 1, 16384)),
                             ("freq deep out", (64, 384, 8, 256))):
    d = torch.randn(Bn * A, Cc, T, device=DEV)                       # (B * A, C, T) fp32 sample-major, as the DConv branches hand over
    v = d.view(Bn, A, Cc, T).permute(0, 2, 1, 3)
    out = clast.empty(Bn, A, T, Cc, DEV)
    ms = timed(lambda: clast.from_cm(v, out=out))
    by = d.numel() * 4 + out.numel() * 2
    print(f"from_cm {name:18s} {ms:7.3f} ms  {by / ms / 1e9:6.2f} TB/s")
    back = torch.empty_like(d)
    ms = timed(lambda: clast.to_cm(out, out=back.view(Bn, A, Cc, T).permute(0, 2, 1, 3)))
    print(f"to_cm   {name:18s} {ms:7.3f} ms  {by / ms / 1e9:6.2f} TB/s")

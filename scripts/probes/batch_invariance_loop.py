"""Dev probe: the batch-of-8 vs singles forward comparison of tests/test_gpu_fullsize_properties.py, repeated, with allocator churn between
repetitions (a cross-stream reuse hazard shows as an occasional large error).   python scripts/probes/batch_invariance_loop.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from remfx_amd import ops
from remfx_amd.hdemucs import HDemucs

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
DEV = torch.device("cuda:0")
ops.set_gemm_precision(os.environ.get("RFX_GEMM_PREC", "bf16"))
torch.manual_seed(11)
net = HDemucs(sources=["mixture"], audio_channels=1, nfft=4096, channels=48).to(DEV).eval()
with torch.no_grad():
    for n, p in net.named_parameters():
        if n.endswith(".scale"):
            p.fill_(0.3)
g = torch.Generator().manual_seed(12)
x = (torch.randn(8, 1, 262144, generator=g) * 0.1).to(DEV)
# RFX_PROBE_KEEP=1: every tensor torch.empty / empty_like / zeros hands out during a forward pass stays referenced until the pass has
# finished on the device -- no block is reused inside a pass.  If the failures vanish with it, they are use-after-free across streams.
KEEP = os.environ.get("RFX_PROBE_KEEP", "0") != "0"
keep = []
if KEEP:
    for name in ("empty", "empty_like", "zeros", "empty_strided", "zeros_like"):
        orig = getattr(torch, name)
        def mk(orig):
            def f(*a, **k):
                t = orig(*a, **k)
                keep.append(t)
                return t
            return f
        setattr(torch, name, mk(orig))
_net = net
def net(x):
    y = _net(x)
    if KEEP:
        torch.cuda.synchronize()
        keep.clear()
    return y
junk = []
for r in range(reps):
    with torch.no_grad():
        yb = net(x)
        ys = torch.cat([net(x[i:i + 1]) for i in range(8)], 0)
    scale = float(ys.pow(2).mean().sqrt())
    err = float((yb - ys).pow(2).mean().sqrt())
    per = [(float((yb[i] - ys[i]).abs().max())) for i in range(8)]
    print(f"rep {r}: rms err / scale = {err / scale:.3e}   per-clip max: " + " ".join(f"{v:.1e}" for v in per), flush=True)
    # allocator churn: odd-sized blocks allocated / freed on the default stream
    junk = [torch.empty((1 + (7919 * (r + 3) * k) % 50_000_000,), device=DEV) for k in range(1, 6)]
    del junk[::2]

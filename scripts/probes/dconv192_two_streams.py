"""Dev probe: the channel-major C = 192 DConv branch (hdemucs._DConv, layer-by-layer path) on two streams at once -- the frequency
branch's shape on the main stream, the time branch's on a side stream -- against serial references."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from remfx_amd import ops
from remfx_amd.hdemucs import _DConv

DEV = torch.device("cuda:0")
ops.set_gemm_precision("bf16")
torch.manual_seed(0)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 192
mf, mt = _DConv(C, depth=2, init=0.3).to(DEV).eval(), _DConv(C, depth=2, init=0.3).to(DEV).eval()
g = torch.Generator().manual_seed(1)
side = torch.cuda.Stream(priority=-1)
bad = 0
for B in (8, 1, 8, 1):
    xf = torch.randn(B * 32, C, 256, generator=g).to(DEV)
    xt = torch.randn(B, C, 4096, generator=g).to(DEV)
    with torch.no_grad():
        rf, rt = mf(xf).clone(), mt(xt).clone()
    torch.cuda.synchronize()
    for it in range(60):
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        xt.record_stream(side)
        with torch.no_grad():
            with torch.cuda.stream(side):
                yt = mt(xt)
            yf = mf(xf)
        main.wait_stream(side)
        torch.cuda.synchronize()
        ef, et = float((yf - rf).abs().max()), float((yt - rt).abs().max())
        if ef > 1e-5 or et > 1e-5:
            bad += 1
            if bad <= 8:
                print(f"B={B} it {it}: freq-shape err {ef:.3e}  time-shape err {et:.3e}  (|y| {float(rf.abs().max()):.2f})", flush=True)
print(f"{bad} bad of 240")

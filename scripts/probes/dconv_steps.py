"""Dev probe: which STEP of the frequency branch's channel-major C = 192 DConv (conv1 + statistics, GroupNorm + GELU, conv2 + statistics,
GroupNorm + GLU + residual) first leaves the one-stream result when the time-branch stream is on (batch of 8)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from remfx_amd import ops, hdemucs
from remfx_amd.hdemucs import HDemucs

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
DEV = torch.device("cuda:0")
ops.set_gemm_precision("bf16")
torch.manual_seed(11)
net = HDemucs(sources=["mixture"], audio_channels=1, nfft=4096, channels=48).to(DEV).eval()
with torch.no_grad():
    for n, p in net.named_parameters():
        if n.endswith(".scale"):
            p.fill_(0.3)
x = (torch.randn(8, 1, 262144, generator=torch.Generator().manual_seed(12)) * 0.1).to(DEV)


def run():
    hdemucs._DCONV_DBG = []
    with torch.no_grad():
        y = net(x)
    torch.cuda.synchronize()
    log = [(t, c.cpu(), None if s is None else s.cpu()) for t, c, s in hdemucs._DCONV_DBG]
    hdemucs._DCONV_DBG = None
    return y, log


hdemucs.TWO_STREAMS = False
yref, lref = run()
print(len(lref), "logged steps:", [(t, tuple(c.shape)) for t, c, _ in lref][:10])
hdemucs.TWO_STREAMS = True
for r in range(reps):
    y, log = run()
    if float((y - yref).abs().max()) <= 1e-6:
        continue
    for k, ((t, c, s), (t0, c0, s0)) in enumerate(zip(log, lref)):
        if c.shape != c0.shape:
            continue
        dc = ((c - c0).abs() / c0.abs().clamp_min(1e-30))
        ds = None if s is None else ((s - s0).abs() / s0.abs().clamp_min(1e-30)).max(dim=-1).values
        if float(dc.max()) > 1e-7 or (ds is not None and float(ds.max()) > 1e-12):
            nb = int((dc > 1e-7).sum())
            print(f"rep {r}: first step off = #{k} {t} (tensor of {tuple(c.shape)} samples): {nb} samples differ, max rel {float(dc.max()):.2e}"
                  + ("" if ds is None else f"; statistics differ in {int((ds > 1e-12).sum())} samples, max rel {float(ds.max()):.2e}")
                  + f"; samples {torch.nonzero(dc > 1e-7).flatten()[:8].tolist()}", flush=True)
            break
print("done")

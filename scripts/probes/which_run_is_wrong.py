"""Dev probe: with the time-branch stream on, when a clip of the batch-of-8 forward and its single-clip forward disagree, which of
the two left the one-stream result?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from remfx_amd import ops, hdemucs
from remfx_amd.hdemucs import HDemucs

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
DEV = torch.device("cuda:0")
ops.set_gemm_precision("bf16")
torch.manual_seed(11)
net = HDemucs(sources=["mixture"], audio_channels=1, nfft=4096, channels=48).to(DEV).eval()
with torch.no_grad():
    for n, p in net.named_parameters():
        if n.endswith(".scale"):
            p.fill_(0.3)
x = (torch.randn(8, 1, 262144, generator=torch.Generator().manual_seed(12)) * 0.1).to(DEV)
hdemucs.TWO_STREAMS = False
net._dbg = {}
with torch.no_grad():
    ref_b = net(x).clone()
    torch.cuda.synchronize()
    ref_dbg = {k: {i: v.cpu() for i, v in net._dbg[k].items()} for k in ("samp", "d")}
    ref_sk = [t.cpu() for t in net._dbg["saved"]]
    ref_s = torch.cat([net(x[i:i + 1]) for i in range(8)], 0).clone()
torch.cuda.synchronize()
print("one stream: batch vs singles max", float((ref_b - ref_s).abs().max()))
hdemucs.TWO_STREAMS = True
cnt = {"batch": 0, "single": 0}
for r in range(reps):
    with torch.no_grad():
        yb = net(x)
        torch.cuda.synchronize()
        cur = {k: {i: v.cpu() for i, v in net._dbg[k].items()} for k in ("samp", "d")}
        cur_sk = [t.cpu() for t in net._dbg["saved"]]
        ys = torch.cat([net(x[i:i + 1]) for i in range(8)], 0)
    torch.cuda.synchronize()
    for i in range(8):
        eb, es = float((yb[i] - ref_b[i]).abs().max()), float((ys[i] - ref_s[i]).abs().max())
        if eb > 1e-6:
            cnt["batch"] += 1
        if es > 1e-6:
            cnt["single"] += 1
        if eb > 1e-6 or es > 1e-6:
            rel = lambda a, b: abs(float(a[i]) - float(b[i])) / max(abs(float(b[i])), 1e-30)
            parts = [f"{k}[{j}] {rel(cur[k][j], ref_dbg[k][j]):.1e}" for j in sorted(cur["samp"]) for k in ("samp", "d")]
            parts += [f"e[{j}] {rel(cur_sk[j], ref_sk[j]):.1e}" for j in range(len(cur_sk))]
            print(f"rep {r} clip {i}: batch off {eb:.2e} single off {es:.2e} | " + " ".join(parts), flush=True)
print(cnt)

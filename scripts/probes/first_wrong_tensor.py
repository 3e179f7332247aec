"""Dev probe: batch of 8 against single clips with the time-branch stream on; when a clip's output differs, which skip tensor of the
encoders (per-clip checksums kept by hdemucs.forward when net._dbg is a dict) is the first to differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from remfx_amd import ops
from remfx_amd.hdemucs import HDemucs

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
DEV = torch.device("cuda:0")
ops.set_gemm_precision("bf16")
torch.manual_seed(11)
net = HDemucs(sources=["mixture"], audio_channels=1, nfft=4096, channels=48).to(DEV).eval()
with torch.no_grad():
    for n, p in net.named_parameters():
        if n.endswith(".scale"):
            p.fill_(0.3)
x = (torch.randn(8, 1, 262144, generator=torch.Generator().manual_seed(12)) * 0.1).to(DEV)
net._dbg = {}


def run(inp):
    with torch.no_grad():
        y = net(inp)
    torch.cuda.synchronize()
    d = {"saved": [t.cpu() for t in net._dbg["saved"]], "saved_t": [t.cpu() for t in net._dbg["saved_t"]], "x": net._dbg["x"].cpu()}
    return y, d


nbad = 0
for r in range(reps):
    yb, db = run(x)
    for i in range(8):
        ys, ds = run(x[i:i + 1])
        err = float((yb[i] - ys[0]).abs().max())
        if err > 1e-6:
            nbad += 1
            msg = []
            for key in ("saved_t", "saved"):
                for li, (tb, ts) in enumerate(zip(db[key], ds[key])):
                    rel = abs(float(tb[i]) - float(ts[0])) / max(abs(float(ts[0])), 1e-30)
                    msg.append(f"{key}[{li}] {rel:.1e}")
            rel = abs(float(db["x"][i]) - float(ds["x"][0])) / max(abs(float(ds["x"][0])), 1e-30)
            print(f"rep {r} clip {i}: out err {err:.2e} | " + " ".join(msg) + f" x {rel:.1e}", flush=True)
print("bad clips:", nbad)

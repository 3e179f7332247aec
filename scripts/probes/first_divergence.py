"""Dev probe: run the headline HDemucs forward repeatedly on the same batch and report the first module (forward order) whose output
checksum moves by more than 1e-5 relative between repetitions -- locates a cross-stream hazard."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from remfx_amd import ops
from remfx_amd.hdemucs import HDemucs

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
DEV = torch.device("cuda:0")
ops.set_gemm_precision("bf16")
torch.manual_seed(11)
net = HDemucs(sources=["mixture"], audio_channels=1, nfft=4096, channels=48).to(DEV).eval()
with torch.no_grad():
    for n, p in net.named_parameters():
        if n.endswith(".scale"):
            p.fill_(0.3)
x = (torch.randn(B, 1, 262144, generator=torch.Generator().manual_seed(12)) * 0.1).to(DEV)
log = []


def hook(name):
    def f(mod, inp, out):
        outs = out if isinstance(out, (tuple, list)) else (out,)
        vals = []
        for o in outs:
            if torch.is_tensor(o):
                vals.append(o.detach().double().abs().sum())          # stays on the device, on the stream the module ran on
        log.append((name, vals, torch.cuda.current_stream()))
    return f


for n, m in net.named_modules():
    if n and n.count(".") <= 1:
        m.register_forward_hook(hook(n))
ref = None
for r in range(reps):
    log.clear()
    with torch.no_grad():
        y = net(x)
    torch.cuda.synchronize()
    cur = [(n, [float(v) for v in vals]) for n, vals, _ in log]
    if ref is None:
        ref = cur
        print(len(cur), "hooked outputs; final checksum", float(y.double().abs().sum()))
        continue
    for (n0, v0), (n1, v1) in zip(ref, cur):
        bad = [abs(a - b) / max(abs(a), 1e-30) for a, b in zip(v0, v1)]
        if n0 != n1 or any(b > 1e-5 for b in bad):
            print(f"rep {r}: first divergence at {n1}: rel {max(bad):.3e}", flush=True)
            break
print("done")

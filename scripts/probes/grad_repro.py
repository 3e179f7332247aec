"""Dev probe: which parameter gradients of a Demucs training step differ between two identical one-stream runs (a remaining
order-dependent reduction), and do the forward output / loss differ?   python scripts/probes/grad_repro.py [clips]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from remfx_amd import hdemucs as hd, models as md, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
ops.set_gemm_precision(os.environ.get("RFX_GEMM_PREC", "bf16"))
hd.TWO_STREAMS = md.METRIC_STREAM = False
ops.GradSink.MODE = "main"
model = bench.build_model("demucs", dev)
opt = model.configure_optimizers()["optimizer"]
flat = opt.flat
data = bench.synthetic_batch(B, 0, dev)
names = [n for n, p in model.named_parameters() if p.requires_grad]


def run():
    opt.zero_grad()
    loss = model.training_step(data, 0)
    out = model.model.sample(data[0]).detach().clone() if False else None
    loss.backward()
    flat.join()
    torch.cuda.synchronize()
    return flat.grad.detach().clone(), float(loss), {k: float(v) for k, v in model.logged.items()}


run()
a, la, ma = run()
b, lb, mb = run()
print("loss", la, lb, la == lb, "logged equal:", ma == mb, {k: (ma[k], mb[k]) for k in ma if ma[k] != mb[k]})
bad = []
for i, (o, p) in enumerate(zip(flat.offsets, flat.params)):
    n = p.numel()
    if not torch.equal(a[o:o + n], b[o:o + n]):
        d = float((a[o:o + n] - b[o:o + n]).norm() / a[o:o + n].norm().clamp_min(1e-30))
        bad.append((names[i] if i < len(names) else str(i), n, d))
print(len(bad), "of", len(flat.params), "parameter gradients differ")
for nm, n, d in bad[:80]:
    print(f"  {nm:70s} {n:9d} {d:.2e}")

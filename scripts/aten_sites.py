"""Dev tool: which Python call sites of one Demucs training step issue torch-native (ATen) GPU kernels -- copies, fills, adds."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from remfx_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda:0")
ops.set_gemm_precision("bf16")
model = bench.build_model("demucs", dev)
opt = model.configure_optimizers()["optimizer"]
data = bench.synthetic_batch(B, 0, dev)


def step():
    opt.zero_grad()
    loss = model.training_step(data, 0)
    loss.backward()
    opt.step(clip_norm=10.0)


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    dt = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
    if not ev.name.startswith("aten::") or dt <= 0:
        continue
    site = "?"
    for fr in ev.stack:
        if "remfx_amd" in fr or "bench.py" in fr:
            site = fr.split("remfx_amd/")[-1][:70]
            break
    a = agg[(ev.name, site)]
    a[0] += 1; a[1] += dt
tot = sum(v[1] for v in agg.values())
print(f"ATen leaf ops with GPU time: {sum(v[0] for v in agg.values())} calls, {tot / 1e3:.2f} ms (B={B})")
for (n, s), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{t / 1e3:7.3f} ms x{c:4d}  {n:28s} {s}")

"""Dev tool: per-step wall time of the first steps of the first process on a fresh box (how long the cold phase lasts)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from remfx_amd import ops

dev = torch.device("cuda:0")
ops.set_gemm_precision("bf16")
model = bench.build_model("demucs", dev)
opt = model.configure_optimizers()["optimizer"]
data = bench.synthetic_batch(64, 0, dev)
ts = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad()
    loss = model.training_step(data, i)
    loss.backward()
    opt.step(clip_norm=10.0)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(" ".join(f"{t:.1f}" for t in ts))

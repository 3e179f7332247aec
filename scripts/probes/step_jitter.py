"""Dev probe: per-step GPU intervals (events on the compute stream, no host sync inside the loop) of the 8-clip Demucs step: is a
slow run uniformly slow or a mix of fast and slow steps?   python scripts/probes/step_jitter.py [clips] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from remfx_amd import ddp, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
ops.set_gemm_precision("bf16")
model = bench.build_model("demucs", dev)
cfg = model.configure_optimizers()
opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
data = bench.synthetic_batch(B, 0, dev)
ops.enter_compute_stream(dev)
gc = ops.StepGC().__enter__()


def step(i):
    opt.zero_grad()
    loss = model.training_step(data, i)
    loss.backward()
    opt.step(clip_norm=10.0)
    sched.step()
    gc.tick()


for i in range(60):
    step(i)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
host = []
ev[0].record()
t0 = time.time()
for i in range(N):
    h0 = time.time()
    step(i)
    host.append(time.time() - h0)
    ev[i + 1].record()
torch.cuda.synchronize()
wall = (time.time() - t0) / N * 1e3
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(N)]
srt = sorted(ms)
hs = sorted(h * 1e3 for h in host)
print(f"wall {wall:.2f} ms/step; gpu interval min {srt[0]:.2f} p10 {srt[N // 10]:.2f} median {srt[N // 2]:.2f} p90 {srt[9 * N // 10]:.2f} max {srt[-1]:.2f}")
print(f"host enqueue per step: min {hs[0]:.2f} median {hs[N // 2]:.2f} p90 {hs[9 * N // 10]:.2f} max {hs[-1]:.2f}")
print("first 40 intervals:", " ".join(f"{v:.1f}" for v in ms[:40]))

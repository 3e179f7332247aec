"""Dev tool: is a training step bound by the host?  Times (a) the Python thread enqueueing one step (return of opt.step(), no
synchronisation inside) and (b) the same step to GPU completion, plus a cProfile of the enqueue path.   python scripts/host_time.py [B] [demucs|dcunet|umx|tcn]"""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remfx_amd import ops
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
WORK = sys.argv[2] if len(sys.argv) > 2 else "demucs"
dev = torch.device("cuda:0")
ops.set_gemm_precision("bf16" if WORK == "demucs" else "bf16x3")
model = bench.build_model(WORK, dev)
opt = model.configure_optimizers()["optimizer"]
data = bench.synthetic_batch(B, 0, dev)


def step():
    opt.zero_grad()
    loss = model.training_step(data, 0)
    loss.backward()
    opt.step(clip_norm=10.0)


for _ in range(5):
    step()
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
print(f"B={B}: host enqueue {sorted(enq)[len(enq)//2]:.1f} ms (min {min(enq):.1f}), to GPU completion {sorted(tot)[len(tot)//2]:.1f} ms (min {min(tot):.1f})")
# back-to-back steps (the bench's regime): wall per step
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize(); print(f"back-to-back: {(time.perf_counter() - t0) * 50:.1f} ms per step")
pr = cProfile.Profile()
with torch.autograd.set_multithreading_enabled(False):      # backward on this thread, so that the profile sees its Python half
    step()
    pr.enable()
    for _ in range(5):
        step()
    pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:80]))

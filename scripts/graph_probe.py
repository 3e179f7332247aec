"""Dev probe: does one Demucs training step (forward, loss, backward through autograd, side-stream weight gradients, clip, AdamW)
capture into a hipGraph, and what does replaying it buy at B = 64 / B = 8?   python scripts/graph_probe.py [B] [steps]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from remfx_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
ops.set_gemm_precision("bf16")
os.environ["RFX_STRICT_NATIVE"] = "1"
model = bench.build_model("demucs", dev)
opt = model.configure_optimizers()["optimizer"]
data = bench.synthetic_batch(B, 0, dev)
ops.enter_compute_stream(dev)


def step():
    opt.zero_grad()
    loss = model.training_step(data, 0)
    loss.backward()
    opt.step(clip_norm=10.0)
    return loss


def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3, out


for _ in range(30 if B >= 32 else 100):
    step()
ms_eager, loss = timed(step, K)
print(f"B={B} eager: {ms_eager:.2f} ms/step, loss {float(loss):.5f}", flush=True)
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        static_loss = step()
except Exception as e:
    import traceback
    traceback.print_exc()
    raise SystemExit(1)
ms_graph, _ = timed(g.replay, K)
print(f"B={B} graph replay: {ms_graph:.2f} ms/step, loss {float(static_loss):.5f}", flush=True)
ms_eager2, loss = timed(step, K)
print(f"B={B} eager again: {ms_eager2:.2f} ms/step, loss {float(loss):.5f}")

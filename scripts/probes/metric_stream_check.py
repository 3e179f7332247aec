"""Dev probe: the Input_* metrics computed on the metric stream (models.METRIC_STREAM) during training steps on batches of alternating
size against the same metrics computed serially; also the loss of every step against a run with the stream off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from remfx_amd import ops, models
import bench

dev = torch.device("cuda:0")
ops.set_gemm_precision("bf16")
model = bench.build_model("demucs", dev)
opt = model.configure_optimizers()["optimizer"]
batches = [bench.synthetic_batch(b, s, dev) for b, s in ((8, 0), (5, 1), (8, 2), (3, 3))]
ref = []
for x, y, *_ in batches:
    with torch.no_grad():
        ref.append({m: float((-1 if m == "SISDR" else 1) * model.metrics[m](x, y)) for m in model.metrics})
torch.cuda.synchronize()
bad = 0
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 80):
    k = it % len(batches)
    opt.zero_grad()
    loss = model.training_step(batches[k], it)
    loss.backward()
    opt.step(clip_norm=10.0)
    got = {m: float(model.logged[f"Input_{m}"]) for m in model.metrics}
    for m in got:
        if abs(got[m] - ref[k][m]) > 1e-6 * max(1.0, abs(ref[k][m])):
            bad += 1
            print(f"step {it} batch {k}: Input_{m} {got[m]!r} vs serial {ref[k][m]!r}", flush=True)
print(f"{bad} mismatches; last loss {float(loss):.5f}")

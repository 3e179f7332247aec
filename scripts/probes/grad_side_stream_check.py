"""Dev probe: does the weight-gradient side stream (the default training configuration) disturb the backward pass's own reductions?
The flat gradient of one Demucs training step on a fixed batch, repeated with the side stream on, against the same step with the
weight gradients on the compute stream (`sink.side = None`): per-parameter relative differences; atomics-order noise is ~1e-6."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from remfx_amd import ops
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 25
dev = torch.device("cuda:0")
ops.set_gemm_precision("bf16")
ops.enter_compute_stream(dev)
model = bench.build_model("demucs", dev)
opt = model.configure_optimizers()["optimizer"]
flat = opt.flat
data = bench.synthetic_batch(B, 0, dev)
names = [n for n, p in model.model.named_parameters() if p.requires_grad]


def grads(side_on):
    sink = flat.sink
    keep = sink.side
    if not side_on:
        sink.side = None
    try:
        opt.zero_grad()
        loss = model.training_step(data, 0)
        loss.backward()
        flat.join()
        torch.cuda.synchronize()
        return flat.grad.detach().clone(), float(loss)
    finally:
        sink.side = keep


for _ in range(2):
    grads(True)
ref, lref = grads(False)
ref2, _ = grads(False)
sizes = [p.numel() for p in flat.params]
offs = flat.offsets


gnorm = float(ref.norm())
rms = gnorm / ref.numel() ** 0.5


def per_param(g):
    """difference of a parameter's gradient, relative to its own norm -- but a gradient that is rounding noise around an exact zero (a
    convolution bias in front of a GroupNorm) is measured against the whole gradient's RMS level instead"""
    out = []
    for o, n in zip(offs, sizes):
        a, b = g[o:o + n], ref[o:o + n]
        out.append(float((a - b).norm() / max(float(b.norm()), 1e-3 * rms * n ** 0.5)))
    return out


base = per_param(ref2)
print(f"one-stream run to run: worst parameter {max(base):.2e}; loss {lref:.6f}; |grad| {gnorm:.4e}")
worst = 0.0
for r in range(reps):
    g, l = grads(True)
    rel = per_param(g)
    w = max(rel)
    worst = max(worst, w)
    if w > 1e-3:
        k = rel.index(w)
        print(f"rep {r}: loss {l:.6f}; worst parameter #{k} {names[k] if k < len(names) else '?'} ({sizes[k]} elements, |g| {float(ref[offs[k]:offs[k] + sizes[k]].norm()):.2e}): {w:.2e}; "
              f"parameters above 1e-3: {sum(v > 1e-3 for v in rel)}; whole gradient {float((g - ref).norm()) / gnorm:.2e}", flush=True)
print(f"side stream on, {reps} repetitions: worst per-parameter relative difference {worst:.2e}; last whole-gradient difference {float((g - ref).norm()) / gnorm:.2e}")

"""Times every GroupNorm forward call of one Demucs training step (stats kernel + apply) with CUDA events, to find the launches
behind gn_stats_kernel's 0.06-of-HBM figure in the r02c (mid-round) profile.   python scripts/probes/gn_stats_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench  # noqa: E402
from remfx_amd import nnops, ops  # noqa: E402

ops.set_gemm_precision("bf16")
dev = torch.device("cuda:0")
model = bench.build_model("demucs", dev)
batch = bench.synthetic_batch(64, 0, dev)
orig = nnops._GroupNormFn.forward
rows = []


def timed(ctx, x, gamma, beta, groups, eps, mode, res, scale, sums=None):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    y = orig(ctx, x, gamma, beta, groups, eps, mode, res, scale, sums)
    e.record()
    rows.append((s, e, tuple(x.shape), groups, mode, sums is not None, str(x.dtype)))
    return y


nnops._GroupNormFn.forward = staticmethod(timed)
for it in range(2):
    rows.clear()
    loss = model.training_step(batch, 0)
    loss.backward()
    torch.cuda.synchronize()
for s, e, shp, g, mode, given, dt in rows:
    n = 1
    for d in shp:
        n *= d
    print(f"{s.elapsed_time(e) * 1e3:8.1f} us  x={shp} G={g} mode={mode} stats_given={given} {dt}  {n * 4 / 1e6:.0f} MB")

"""Dev probe: which parameters of the Demucs step still get their gradient through autograd's AccumulateGrad (one at::add launch
each) instead of a GradSink write?   python scripts/probe_accum.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from remfx_amd import ops

dev = torch.device("cuda:0")
ops.set_gemm_precision("bf16")
model = bench.build_model("demucs", dev)
opt = model.configure_optimizers()["optimizer"]
data = bench.synthetic_batch(8, 0, dev)
ops.enter_compute_stream(dev)
names = {id(p): n for n, p in model.named_parameters()}
for _ in range(2):
    opt.zero_grad()
    sink = ops.SINK
    loss = model.training_step(data, 0)
    loss.backward()
    writes = list(sink.writes) if sink is not None else None
    flat = sink.flat if sink is not None else None
    opt.step(clip_norm=10.0)
torch.cuda.synchronize()
miss = [names.get(id(p), "?") for p, w in zip(flat.params, writes) if w == 0]
print(len(flat.params), "parameters,", len(miss), "not written through the sink")
import collections
c = collections.Counter(".".join(n.split(".")[-2:]) if "dconv" not in n else "dconv." + n.split(".")[-1] + ("[cl]" if any(f"encoder.{i}." in n for i in (0, 1)) else "") for n in miss)
for k, v in c.most_common(40):
    print(v, k)
print(miss[:60])

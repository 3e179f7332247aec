#!/bin/bash
mkdir -p gpurun_out/r3k
for i in 1 2 3; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>> gpurun_out/r3k/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('run$i', d['ms_per_step'], d['config']['final_loss'])" | tee -a gpurun_out/r3k/ab.txt
done
python - <<'P' 2>&1 | tee gpurun_out/r3k/steps.txt
import os, sys, time, torch
os.environ.setdefault("RFX_STRICT_NATIVE", "1")
sys.path.insert(0, ".")
import bench
from remfx_amd import ops
ops.set_gemm_precision("bf16")
dev = torch.device("cuda", 0)
model = bench.build_model("demucs", dev)
cfg = model.configure_optimizers(); opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
data = bench.synthetic_batch(64, 0, dev)
ts = []
for i in range(14):
    torch.cuda.synchronize(); t = time.time()
    opt.zero_grad(); loss = model.training_step(data, i); loss.backward(); opt.step(clip_norm=10.0); sched.step()
    torch.cuda.synchronize(); ts.append((time.time() - t) * 1e3)
print("per-step ms:", " ".join(f"{t:.1f}" for t in ts), " reserved GB", torch.cuda.memory_reserved() / 1e9)
P

#!/bin/bash
# dev: same-box A/B of a bench flag
mkdir -p gpurun_out/ab; : > gpurun_out/ab/ab2.txt
for rep in 1 2 3; do
  for v in "$FLAG" ""; do
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-also --no-cpu-baseline $v > gpurun_out/ab/b.json 2>> gpurun_out/ab/err.txt
    python - "[$v]" <<'P' >> gpurun_out/ab/ab2.txt
import json,sys
d=json.loads(open("gpurun_out/ab/b.json").read().strip().splitlines()[-1])
print(sys.argv[1], d["ms_per_step"], d["config"].get("final_loss"))
P
  done
done
cat gpurun_out/ab/ab2.txt

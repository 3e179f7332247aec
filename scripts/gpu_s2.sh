#!/bin/bash
mkdir -p gpurun_out/s2
python -m pytest tests -m gpu -x -q > gpurun_out/s2/pytest.log 2>&1; tail -6 gpurun_out/s2/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 --no-also --no-cpu-baseline > gpurun_out/s2/b1.json 2> gpurun_out/s2/b1.err
python bench.py --gpus 1 --steps 20 --warmup 5 --no-also --no-cpu-baseline > gpurun_out/s2/b2.json 2>> gpurun_out/s2/b1.err
python - <<'P'
import json
for f in ("b1","b2"):
    d=json.loads(open(f"gpurun_out/s2/{f}.json").read().strip().splitlines()[-1])
    print(f, d["ms_per_step"], d["value"], d["config"].get("final_loss"))
P

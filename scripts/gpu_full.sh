#!/bin/bash
mkdir -p gpurun_out/full
python -m pytest tests -m gpu -x -q > gpurun_out/full/pytest.log 2>&1; tail -4 gpurun_out/full/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/full/smoke.log 2>&1; tail -2 gpurun_out/full/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err; python - <<'P'
import json
d=json.loads(open("gpurun_out/full/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["also"]["demucs_fwd"]["stages"]["stft"])
P

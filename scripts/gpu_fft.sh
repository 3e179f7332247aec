#!/bin/bash
mkdir -p gpurun_out/fft
timeout 900 python -m pytest tests/test_gpu_stft.py tests/test_gpu_fullsize_properties.py -x -q -m gpu > gpurun_out/fft/t.log 2>&1
tail -15 gpurun_out/fft/t.log
python bench.py --gpus 1 --steps 20 --warmup 5 --no-also --no-cpu-baseline > gpurun_out/fft/b1.json 2> gpurun_out/fft/b1.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/fft/b1.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], d["config"].get("final_loss"))
P

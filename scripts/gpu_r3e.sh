#!/bin/bash
mkdir -p gpurun_out/r3e
python -m pytest tests/test_gpu_train_script.py tests/test_gpu_multirank.py -x -q > gpurun_out/r3e/t.log 2>&1
tail -4 gpurun_out/r3e/t.log
for m in off main side side off; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --sink $m 2>> gpurun_out/r3e/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m', d['ms_per_step'], d['config']['final_loss'], d['config']['param_abs_sum'])" | tee -a gpurun_out/r3e/ab.txt
done
tail -3 gpurun_out/r3e/bench.err

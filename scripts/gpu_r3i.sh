#!/bin/bash
mkdir -p gpurun_out/r3i
for m in "" "--no-fused-dconv" "" "--no-fused-dconv"; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also $m 2>> gpurun_out/r3i/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$m]', d['ms_per_step'], d['config']['final_loss'])" | tee -a gpurun_out/r3i/ab.txt
done
RFX_TEST_MODES=bf16 python -m pytest tests/test_gpu_hdemucs.py tests/test_gpu_bf16_mixed.py tests/test_gpu_fullsize_properties.py tests/test_gpu_train_script.py tests/test_gpu_classifier_chain.py -x -q > gpurun_out/r3i/t.log 2>&1; tail -4 gpurun_out/r3i/t.log

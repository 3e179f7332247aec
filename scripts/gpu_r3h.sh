#!/bin/bash
mkdir -p gpurun_out/r3h
RFX_TEST_MODES=bf16 python -m pytest tests/test_gpu_hdemucs.py tests/test_gpu_bf16_mixed.py tests/test_gpu_fullsize_properties.py -x -q > gpurun_out/r3h/t.log 2>&1; tail -5 gpurun_out/r3h/t.log
for m in "" "--no-fused-dconv" ""; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also $m 2>> gpurun_out/r3h/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$m]', d['ms_per_step'], d['config']['final_loss'])" | tee -a gpurun_out/r3h/ab.txt
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/gpurun_out/r3h/kt -o kt -- python /root/repo/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-also > /root/repo/gpurun_out/r3h/kt.log 2>&1
find /root/repo/gpurun_out/r3h/kt -name "*kernel_stats.csv" -exec cp {} /root/repo/gpurun_out/r3h/kernel_stats.csv \;
rm -rf /root/repo/gpurun_out/r3h/kt
head -30 /root/repo/gpurun_out/r3h/kernel_stats.csv | cut -c1-150

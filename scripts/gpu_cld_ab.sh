#!/bin/bash
# dev: the fused DConv kernels -- parity tests, then kernel-trace stats of the frequency-branch layer (perf_cldconv.py) and of a short
# bench run (the time-branch pass kernels)
mkdir -p gpurun_out/r06; R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_cldconv.py tests/test_gpu_clchain.py tests/test_gpu_bf16_mixed.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r06/cld_ab_tests.log
cat gpurun_out/r06/cld_ab_tests.log
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pk8
rocprofv3 --kernel-trace --stats -d /tmp/pk8 -o out --output-format csv -- python $R/scripts/perf_cldconv.py > $R/gpurun_out/r06/cld_ab_nw8.log 2>&1
f=$(find /tmp/pk8 -name "*kernel_stats.csv" | head -1)
grep -i "cl_dconv" $f | cut -c1-130
rm -rf /tmp/pkb
rocprofv3 --kernel-trace --stats -d /tmp/pkb -o out --output-format csv -- python $R/bench.py --steps 4 --warmup 2 --preheat 0 --no-exclusive > $R/gpurun_out/r06/cld_bench.log 2>&1
f=$(find /tmp/pkb -name "*kernel_stats.csv" | head -1)
grep -i "cl_dconv" $f | cut -c1-130 | tee $R/gpurun_out/r06/cld_bench_stats.csv
tail -1 $R/gpurun_out/r06/cld_bench.log | cut -c1-300

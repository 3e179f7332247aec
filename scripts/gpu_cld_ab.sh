#!/bin/bash
# dev: A/B of the fused DConv backward kernel forms (RFX_CLD_BWD_NW=4: four waves, default: eight) -- parity tests, then kernel-trace stats
mkdir -p gpurun_out/r06; R=$GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_cldconv.py tests/test_gpu_clchain.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r06/cld_ab_tests.log
cat gpurun_out/r06/cld_ab_tests.log
cd /tmp; export TMPDIR=/tmp
for nw in 4 8; do
  rm -rf /tmp/pk$nw
  RFX_CLD_BWD_NW=$nw rocprofv3 --kernel-trace --stats -d /tmp/pk$nw -o out --output-format csv -- python $R/scripts/perf_cldconv.py > $R/gpurun_out/r06/cld_ab_nw$nw.log 2>&1
  f=$(find /tmp/pk$nw -name "*kernel_stats.csv" | head -1)
  echo "== NW=$nw"; grep -i "cl_dconv\|cl_wgrad" $f | awk -F, '{print $1, "calls", $2, "avg_ns", $4}' | cut -c1-200
  grep -i "cl_dconv\|cl_wgrad" $f > $R/gpurun_out/r06/cld_ab_nw${nw}_stats.csv
  tail -1 $R/gpurun_out/r06/cld_ab_nw$nw.log
done

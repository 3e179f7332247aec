#!/bin/bash
# dev: one-stream kernel-trace stats of the headline step (4 + 2 steps), summary table to gpurun_out/<tag>_onestream.md
TAG=${1:-dev}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt1 -o kt -- python $R/bench.py --one-stream --steps 4 --warmup 2 --preheat 0 --no-cpu-baseline --no-also --no-exclusive > $OUT/kt1.log 2>&1
find /tmp/kt1 -name "*kernel_stats.csv" -exec cp {} $OUT/onestream_kernel_stats.csv \;
cd $R
python scripts/prof_summary.py $OUT/onestream_kernel_stats.csv 6 60 > $OUT/onestream_summary.md
head -75 $OUT/onestream_summary.md

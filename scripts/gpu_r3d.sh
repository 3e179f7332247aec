#!/bin/bash
# per-dispatch kernel trace of the headline step joined with the launch dump
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r3d; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also --dump-launches $OUT/launches.json > $OUT/kt.log 2>&1
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/kt -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/kt
gzip -f $OUT/kernel_trace.csv
ls -la $OUT

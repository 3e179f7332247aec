# kernel-trace profile of the headline bench: bash scripts/prof_kt.sh <tag> [extra bench args]
TAG=${1:-kt}; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python $ROOT/bench.py --steps 4 --warmup 2 --preheat 0 --no-cpu-baseline --no-also --no-exclusive "$@" > $OUT/kt.log 2>&1
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/kt -name "*kernel_trace.csv" -exec cp {} /tmp/kernel_trace.csv \;
rm -rf $OUT/kt
cd $ROOT
python scripts/prof_summary.py $OUT/kernel_stats.csv 6 60 > $OUT/summary.md
python scripts/trace_lanes.py /tmp/kernel_trace.csv 6 > $OUT/lanes.md 2>&1
head -3 /tmp/kernel_trace.csv > $OUT/trace_head.csv

# Collect the judged measurements of one round on a GPU box (run through gpurun from the repo root):
#   bash scripts/measure_round.sh r02 bf16
# writes gpurun_out/<round>/: kernel-trace stats CSV of the headline bench, the two PMC passes (FETCH_SIZE / WRITE_SIZE in
# SEPARATE runs, counters only + kernel trace: rocprofv3 must not combine --pmc with other trace domains on this pool)
# (the stats cover 2 warm-up + 4 timed steps = 6 steps) turned into the per-kernel HBM traffic JSON, and the bench lines of every workload.  Copy the summaries you want judged
# into profiles/ afterwards (gpurun_out/ is scratch).
R=${1:-r03}; MODE=${2:-bf16}
export RFX_BENCH_FULL_DIR=$(pwd)/gpurun_out/$R
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$R; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$MODE -o kt -- python $ROOT/bench.py --steps 4 --warmup 2 --preheat 0 --no-cpu-baseline --no-also --no-exclusive --gemm $MODE > $OUT/kt_$MODE.log 2>&1
find $OUT/kt_$MODE -name "*kernel_stats.csv" -exec cp {} $OUT/${R}_demucs_b64_kernel_stats_$MODE.csv \;
rm -rf $OUT/kt_$MODE
# the one-stream configuration roofline.exclusive is measured in (bench.py picks and prices the dominant kernel there)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt1_$MODE -o kt -- python $ROOT/bench.py --one-stream --steps 4 --warmup 2 --preheat 0 --no-cpu-baseline --no-also --no-exclusive --gemm $MODE > $OUT/kt1_$MODE.log 2>&1
find $OUT/kt1_$MODE -name "*kernel_stats.csv" -exec cp {} $OUT/${R}_demucs_b64_kernel_stats_${MODE}_onestream.csv \;
rm -rf $OUT/kt1_$MODE
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o r -- python $ROOT/bench.py --steps 1 --warmup 1 --preheat 0 --no-cpu-baseline --no-also --no-exclusive --gemm $MODE > $OUT/pmc_$c.log 2>&1
  find $OUT/pmc_$c -name "*counter_collection.csv" -exec cp {} $OUT/pmc_$c.csv \;
  rm -rf $OUT/pmc_$c
done
cd $ROOT
python scripts/collect_pmc.py $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv $OUT/${R}_demucs_b64_pmc_traffic_$MODE.json 2
rm -f $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv
python scripts/prof_summary.py $OUT/${R}_demucs_b64_kernel_stats_$MODE.csv 6 40 $OUT/${R}_demucs_b64_pmc_traffic_$MODE.json > $OUT/${R}_demucs_b64_summary_$MODE.md
cp $OUT/${R}_demucs_b64_pmc_traffic_$MODE.json profiles/ 2>/dev/null    # bench.py joins its per-kernel table with the newest PMC pass
python bench.py --steps 20 --warmup 5 --gemm $MODE > $OUT/bench_demucs_$MODE.json 2> $OUT/bench_demucs_$MODE.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --sink main --gemm $MODE 2>> $OUT/bench_demucs_$MODE.err | tail -1 > $OUT/bench_demucs_${MODE}_sinkmain.json
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-also --sink off --no-fused-dconv --gemm $MODE 2>> $OUT/bench_demucs_$MODE.err | tail -1 > $OUT/bench_demucs_${MODE}_r02path.json
python bench.py --workload demucs_fwd --gemm $MODE > $OUT/bench_demucs_fwd_$MODE.json 2>> $OUT/bench_demucs_$MODE.err
# (TCN / DCUNet / Open-Unmix / chain / the 8-clip batch: scripts/measure_configs.sh, with their own kernel stats, PMC passes and cpu_baseline)
grep -ho '"ms_per_step": [0-9.]*' $OUT/bench_*_$MODE.json | tr '\n' ' '

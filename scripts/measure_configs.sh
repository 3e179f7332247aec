# Evidence for BASELINE configs 2 (TCN, 32 clips), 4 (DCUNet, 4 clips per GPU), 5 (chain inference, 16 clips per GPU) and the
# strong-scaling per-GPU batch of config 3 (Hybrid Demucs, 8 clips): per workload the rocprofv3 kernel-trace stats of the bench
# command, the two PMC passes (FETCH_SIZE / WRITE_SIZE, counters only, separate runs) turned into per-kernel HBM traffic, then the
# bench line itself (roofline joined with that traffic, cpu_baseline on).   bash scripts/measure_configs.sh r04
R=${1:-r04}
export RFX_BENCH_FULL_DIR=$(pwd)/gpurun_out/$R
ROOT=$(pwd); OUT=$ROOT/gpurun_out/$R; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {   # tag, steps-profiled, bench args...
  TAG=$1; NST=$2; shift 2
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$TAG -o kt -- python $ROOT/bench.py "$@" --no-cpu-baseline --no-also --no-exclusive --preheat 0 > $OUT/kt_$TAG.log 2>&1
  find $OUT/kt_$TAG -name "*kernel_stats.csv" -exec cp {} $OUT/${R}_${TAG}_kernel_stats.csv \;
  rm -rf $OUT/kt_$TAG
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${TAG}_$c -o r -- python $ROOT/bench.py "$@" --no-cpu-baseline --no-also --no-exclusive --preheat 0 > $OUT/pmc_${TAG}_$c.log 2>&1
    find $OUT/pmc_${TAG}_$c -name "*counter_collection.csv" -exec cp {} $OUT/pmc_${TAG}_$c.csv \;
    rm -rf $OUT/pmc_${TAG}_$c
  done
  (cd $ROOT && python scripts/collect_pmc.py $OUT/pmc_${TAG}_FETCH_SIZE.csv $OUT/pmc_${TAG}_WRITE_SIZE.csv $OUT/${R}_${TAG}_pmc_traffic.json $NST)
  rm -f $OUT/pmc_${TAG}_FETCH_SIZE.csv $OUT/pmc_${TAG}_WRITE_SIZE.csv
}
run tcn_b32_bf16x3 3 --workload tcn --steps 1 --warmup 2
run dcunet_b4_bf16x3 3 --workload dcunet --steps 2 --warmup 1
run chain_b16_bf16x3 3 --workload chain --steps 2 --warmup 1
run demucs_b8_bf16 3 --workload demucs --batch 8 --steps 2 --warmup 1
cd $ROOT
# bench.py joins roofline.traffic with profiles/r*_<workload>_b<batch>_pmc_traffic_<mode>.json: put this pass there first
cp $OUT/${R}_tcn_b32_bf16x3_pmc_traffic.json profiles/${R}_tcn_b32_pmc_traffic_bf16x3.json
cp $OUT/${R}_dcunet_b4_bf16x3_pmc_traffic.json profiles/${R}_dcunet_b4_pmc_traffic_bf16x3.json
cp $OUT/${R}_chain_b16_bf16x3_pmc_traffic.json profiles/${R}_chain_b16_pmc_traffic_bf16x3.json
cp $OUT/${R}_demucs_b8_bf16_pmc_traffic.json profiles/${R}_demucs_b8_pmc_traffic_bf16.json
for t in tcn_b32_bf16x3 dcunet_b4_bf16x3 chain_b16_bf16x3 demucs_b8_bf16; do
  python scripts/prof_summary.py $OUT/${R}_${t}_kernel_stats.csv 3 25 $OUT/${R}_${t}_pmc_traffic.json > $OUT/${R}_${t}_summary.md 2>/dev/null
done
python bench.py --workload tcn --steps 2 --warmup 2 2>> $OUT/cfg.err | tail -1 > $OUT/bench_tcn_bf16x3.json
python bench.py --workload dcunet 2>> $OUT/cfg.err | tail -1 > $OUT/bench_dcunet_bf16x3.json
python bench.py --workload chain 2>> $OUT/cfg.err | tail -1 > $OUT/bench_chain_bf16x3.json
python bench.py --workload umx 2>> $OUT/cfg.err | tail -1 > $OUT/bench_umx_bf16x3.json
python bench.py --workload demucs --batch 8 --steps 20 --warmup 5 --no-also 2>> $OUT/cfg.err | tail -1 > $OUT/bench_demucs_bf16_b8.json
grep -ho '"ms_per_step": [0-9.]*' $OUT/bench_tcn_bf16x3.json $OUT/bench_dcunet_bf16x3.json $OUT/bench_chain_bf16x3.json $OUT/bench_demucs_bf16_b8.json | tr '\n' ' '
# the per-rank shares of config 3 at 4 / 2 GPUs (16 / 32 clips): the projected-scaling table of DESIGN.md section 7
python bench.py --workload demucs --batch 16 --steps 20 --warmup 5 --no-also --no-cpu-baseline 2>> $OUT/cfg.err | tail -1 > $OUT/bench_demucs_bf16_b16.json
python bench.py --workload demucs --batch 32 --steps 20 --warmup 5 --no-also --no-cpu-baseline 2>> $OUT/cfg.err | tail -1 > $OUT/bench_demucs_bf16_b32.json

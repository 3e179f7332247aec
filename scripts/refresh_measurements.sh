# Refresh the judged measurements on a GPU box: PMC traffic passes (separate FETCH / WRITE runs) + the non-headline benches.
cd /tmp && export TMPDIR=/tmp
R=/root/repo
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o r -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$c.log 2>&1
  find $R/gpurun_out/pmc_$c -name "*counter_collection.csv" -exec cp {} $R/gpurun_out/pmc_$c.csv \;
  rm -rf $R/gpurun_out/pmc_$c
done
cd $R
python scripts/collect_pmc.py gpurun_out/pmc_FETCH_SIZE.csv gpurun_out/pmc_WRITE_SIZE.csv gpurun_out/pmc_traffic_new.json 2
rm -f gpurun_out/pmc_FETCH_SIZE.csv gpurun_out/pmc_WRITE_SIZE.csv
for w in tcn dcunet umx chain; do
  python bench.py --workload $w --no-cpu-baseline 2>&1 | tail -1 > gpurun_out/bench_$w.json
  grep -o '"value": [0-9.]*\|ms_per_step": [0-9.]*' gpurun_out/bench_$w.json | tr '\n' ' '; echo $w
done

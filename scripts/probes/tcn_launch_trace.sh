mkdir -p gpurun_out/r06b; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/tk
rocprofv3 --kernel-trace --output-format csv -d /tmp/tk -o out -- python $R/bench.py --workload tcn --steps 1 --warmup 1 --preheat 0 --no-cpu-baseline --no-also --no-exclusive > $R/gpurun_out/r06b/tcn_trace.log 2>&1
f=$(find /tmp/tk -name "*kernel_trace.csv" | head -1)
python - $f <<'P' > $R/gpurun_out/r06b/tcn_launches.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "gemm_tap_kernel" in r["Kernel_Name"] or "gemm_wgrad_wide" in r["Kernel_Name"]]
half = sel[len(sel) // 2:]
for r in half:
    print(r["Kernel_Name"][:40], r.get("Grid_Size_X") or r.get("Grid_Size"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
P
cat $R/gpurun_out/r06b/tcn_launches.txt | head -120

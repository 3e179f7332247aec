#!/bin/bash
# dev: SQ / TA / TCP / TCC counters of the forward launches of scripts/perf_halo.py (halo-tile kernel next to the tap-major kernel on
# the same layer; separate rocprofv3 --pmc passes, counters only).   PERF_LAYERS=0 bash scripts/pmc_halo.sh
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc_halo; cd /tmp; export TMPDIR=/tmp
export PERF_FWD_ONLY=1 PERF_LAYERS=${PERF_LAYERS:-0}
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_VALU" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TAGRAM0_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET -d $R/gpurun_out/pmc_halo/p$i -o out --output-format csv -- python $R/scripts/perf_halo.py 3 > $R/gpurun_out/pmc_halo/run$i.log 2>&1
done
cd $R
python - <<'P' | tee gpurun_out/pmc_halo/summary.txt
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); calls = collections.Counter()
for f in sorted(glob.glob("gpurun_out/pmc_halo/p*/**/*counter_collection.csv", recursive=True)):
    seen = set()
    first = "/p1/" in f
    ncall = collections.Counter()
    for r in csv.DictReader(open(f)):
        n = re.sub(r"^void ", "", r["Kernel_Name"])[:48]
        if "gemm_halo" not in n and "gemm_tap" not in n: continue
        agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); ncall[n] += 1
            if first: dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; calls[n] += 1
    for n in ncall: agg[n]["_calls_" + f.split("/")[2]] = ncall[n]
for n, t in sorted(dur.items(), key=lambda kv: -kv[1]):
    a = agg[n]; c = calls[n]
    print(f"\n{n}: {c} launches, {t / c:.1f} us each (under the counters)")
    for k in sorted(a):
        if k.startswith("_calls_"): continue
        print(f"   {k:44s} {a[k] / c:16.0f} per launch")
P

#!/bin/bash
# dev: SQ / TA / TCP counters of the weight-gradient launches of scripts/perf_wgrad.py (separate rocprofv3 --pmc passes, counters only)
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/pmc; cd /tmp; export TMPDIR=/tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_MFMA" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
           "TCP_TAGRAM0_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET -d $R/gpurun_out/pmc/p$i -o out --output-format csv -- python $R/scripts/perf_wgrad.py > $R/gpurun_out/pmc/run$i.log 2>&1
done
cd $R
python - <<'P' | tee gpurun_out/pmc/wgrad_summary.txt
import csv, glob, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); calls = collections.Counter()
for f in glob.glob("gpurun_out/pmc/p*/**/*counter_collection.csv", recursive=True):
    seen = set()
    first = "p1" in f
    for r in csv.DictReader(open(f)):
        n = re.sub(r"^void ", "", r["Kernel_Name"])[:60]
        if "wgrad" not in n: continue
        key = n + "|" + r["Grid_Size"]
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if first and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); dur[key] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; calls[key] += 1
for key, t in sorted(dur.items(), key=lambda kv: -kv[1])[:8]:
    a = agg[key]; c = calls[key]
    print(f"\n{key}: {c} launches, {t / c:.1f} us each")
    cyc = t / c * 2400.0      # clocks per launch at 2.4 GHz
    for k in sorted(a):
        v = a[k] / c
        print(f"    {k:42s} {v:16.0f}   per clk {v / cyc:10.3f}")
P
rm -rf gpurun_out/pmc/p1 gpurun_out/pmc/p2 gpurun_out/pmc/p3 gpurun_out/pmc/p4

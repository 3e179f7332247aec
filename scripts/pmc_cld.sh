#!/bin/bash
# dev: SQ counters of the channels-last DConv kernels (two rocprofv3 --pmc passes, counters + kernel trace only)
mkdir -p gpurun_out/pmc_cld; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d /tmp/p1 -o out --output-format csv -- python $R/scripts/perf_cldconv.py > $R/gpurun_out/pmc_cld/run1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA -d /tmp/p2 -o out --output-format csv -- python $R/scripts/perf_cldconv.py > $R/gpurun_out/pmc_cld/run2.log 2>&1
cd $R
python - <<'P' > gpurun_out/pmc_cld/summary.txt
import csv, glob, collections, re
for d in ("/tmp/p1", "/tmp/p2"):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print("no counters in", d); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter(); seen = set()
    for r in csv.DictReader(open(fs[0])):
        n = re.sub(r"^void ", "", r["Kernel_Name"])[:40]
        agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); calls[n] += 1
    for n, a in agg.items():
        if "cl_dconv" in n or "cl_wgrad_kernel" in n:
            print(n, "calls", calls[n], {k: f"{v / calls[n]:.4g}" for k, v in sorted(a.items())})
P
cat gpurun_out/pmc_cld/summary.txt

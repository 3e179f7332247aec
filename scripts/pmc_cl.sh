#!/bin/bash
# SQ counters of the channels-last kernels at one layer's shapes: PERF_CL_LAYERS=48 bash scripts/pmc_cl.sh <outdir>
OUT=${1:-gpurun_out/pmc_cl}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT -d $R/$OUT/a -o a --output-format csv -- python $R/scripts/perf_cl.py 64 > $R/$OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU -d $R/$OUT/b -o b --output-format csv -- python $R/scripts/perf_cl.py 64 > $R/$OUT/b.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
for tag in ("a", "b"):
    fs = glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True)
    if not fs:
        print("no counter file for", tag); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(fs[0])):
        k = row["Kernel_Name"][:70]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
    for k, v in agg.items():
        if "cl_" not in k: continue
        print(tag, k)
        for c, x in sorted(v.items()): print("    %-28s %.4g" % (c, x))
PY

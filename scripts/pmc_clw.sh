#!/bin/bash
# dev: SQ counters of the channels-last weight-gradient kernel (two rocprofv3 --pmc passes, counters + kernel trace only)
mkdir -p gpurun_out/pmc_clw; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PERF_CL_LAYERS=${1:-48}
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA -d /tmp/p1 -o out --output-format csv -- python $R/scripts/perf_clw.py > $R/gpurun_out/pmc_clw/run1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES -d /tmp/p2 -o out --output-format csv -- python $R/scripts/perf_clw.py > $R/gpurun_out/pmc_clw/run2.log 2>&1
cd $R
python - <<'P' > gpurun_out/pmc_clw/summary.txt
import csv, glob, collections, re
for d in ("/tmp/p1", "/tmp/p2"):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print("no counters in", d); continue
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        n = re.sub(r"^void ", "", r["Kernel_Name"])[:34]
        if "cl_wgrad_kernel" not in n:
            continue
        key = (r["Dispatch_Id"], n, r["Grid_Size_X"] if "Grid_Size_X" in r else "")
        rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
        rows[key]["dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    seen = set()
    for (did, n, gsz), a in rows.items():
        sig = (n, gsz, round(a.get("SQ_INSTS_MFMA", a.get("SQ_INSTS_VALU", 0)) / 1e3))
        if sig in seen:
            continue
        seen.add(sig)
        print(n, "grid", gsz, {k: f"{v:.4g}" for k, v in sorted(a.items())})
P
cat gpurun_out/pmc_clw/summary.txt

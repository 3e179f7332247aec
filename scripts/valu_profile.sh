#!/bin/bash
# dev: per-kernel VALU occupancy of the Demucs step (SQ counters, one rocprofv3 --pmc pass; kernels are serialised by the profiler)
mkdir -p gpurun_out/valu; cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -d $R/gpurun_out/valu/pmc -o out --output-format csv -- python $R/bench.py --gpus 1 --steps 2 --warmup 1 --preheat 0 --no-also --no-exclusive --sink main > $R/gpurun_out/valu/run.log 2>&1
cd $R
python - <<'P' > gpurun_out/valu/summary.txt
import csv, glob, collections, re
f = glob.glob("gpurun_out/valu/pmc/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    n = re.sub(r"^void ", "", n)[:90]
    agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); dur[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; calls[n] += 1
tot = sum(dur.values())
print(f"total kernel time {tot/1e3:.1f} ms (3 steps, serialised)")
print(f"{'us/step':>9} {'calls':>5} {'VALUbusy':>8} {'wait':>6} {'valu/vmem':>9}  kernel")
for n, t in sorted(dur.items(), key=lambda kv: -kv[1])[:70]:
    a = agg[n]
    valu_busy = a["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (t * 2400) if t else 0     # quad-cycles -> cycles per SIMD over duration at 2.4 GHz
    wait = a["SQ_WAIT_ANY"] / a["SQ_WAVE_CYCLES"] if a["SQ_WAVE_CYCLES"] else 0
    vm = a["SQ_INSTS_VMEM_WR"] + a["SQ_INSTS_VMEM_RD"]
    print(f"{t/3:9.1f} {calls[n]//3:5d} {valu_busy:8.2f} {wait:6.2f} {a['SQ_INSTS_VALU']/vm if vm else 0:9.1f}  {n}")
P
head -75 gpurun_out/valu/summary.txt

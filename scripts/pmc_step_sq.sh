#!/bin/bash
# dev: SQ activity counters of every kernel of the one-stream headline step (two counter-only passes), one table sorted by wave cycles:
# which kernels are VALU-issue-bound (candidates for packed fp32 math), which wait on memory / LDS
TAG=${1:-dev}; R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/sq1 /tmp/sq2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d /tmp/sq1 -o out --output-format csv -- python $R/bench.py --one-stream --steps 1 --warmup 1 --preheat 0 --no-cpu-baseline --no-also --no-exclusive > $OUT/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d /tmp/sq2 -o out --output-format csv -- python $R/bench.py --one-stream --steps 1 --warmup 1 --preheat 0 --no-cpu-baseline --no-also --no-exclusive > $OUT/sq2.log 2>&1
cd $R
python - $OUT <<'P'
import csv, glob, collections, re, sys
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for d in ("/tmp/sq1", "/tmp/sq2"):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print("no counters in", d); continue
    seen = set()
    for r in csv.DictReader(open(fs[0])):
        n = re.sub(r"^void ", "", r["Kernel_Name"]); n = re.sub(r"\(.*", "", n)[:60]
        agg[n][r["Counter_Name"]] += float(r["Counter_Value"])
        if d == "/tmp/sq1" and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); calls[n] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))
with open(out + "/step_sq_counters.md", "w") as f:
    f.write("SQ counters per kernel, one-stream 64-clip step (2 steps profiled; sums over launches), sorted by wave cycles\n\n")
    f.write("| kernel | launches | wave cycles | VALU-active / wave cycles | any-inst-active / wave cycles | wait-any / wave cycles | VALU insts per MFMA | SALU per MFMA | LDS insts per MFMA | LDS bank-conflict / LDS-active |\n|---|---|---|---|---|---|---|---|---|---|\n")
    for n, a in rows[:70]:
        wc = a.get("SQ_WAVE_CYCLES", 0) or 1
        mf = a.get("SQ_INSTS_MFMA", 0)
        per = lambda k: (f"{a.get(k, 0) / mf:.1f}" if mf else "-")
        la = a.get("SQ_ACTIVE_INST_LDS", 0)
        f.write(f"| `{n}` | {calls[n]} | {wc:.3g} | {a.get('SQ_ACTIVE_INST_VALU', 0) / wc:.2f} | {a.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} | {a.get('SQ_WAIT_ANY', 0) / wc:.2f} | {per('SQ_INSTS_VALU')} | {per('SQ_INSTS_SALU')} | {per('SQ_INSTS_LDS')} | {(a.get('SQ_LDS_BANK_CONFLICT', 0) / la if la else 0):.2f} |\n")
print(open(out + "/step_sq_counters.md").read()[:9000])
P

#!/bin/bash
# dev: GPU busy time (union of kernel intervals over all streams) of the timed steps vs the step time -> launch-gap estimate;
# plus what runs next to the largest zero fill
ROOT=$(pwd); OUT=$ROOT/gpurun_out/gap; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -o kt -- python $ROOT/bench.py --steps 4 --warmup 3 --preheat 0 --no-cpu-baseline --no-also --no-exclusive "$@" > $OUT/kt.log 2>&1
cd $ROOT
F=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
python - "$F" <<'P' | tee $OUT/gap.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(list(rows[0].keys()))
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Grid_Size", "?")) for r in rows)
marks = [s for s, e, n, q, g in iv if "adamw_kernel" in n]
print("launches", len(iv), "adamw marks", len(marks))
for a, b in zip(marks[-4:-1], marks[-3:]):
    seg = [(s, e) for s, e, n, q, g in iv if s >= a and s < b]
    busy, cur_s, cur_e = 0, None, None
    for s, e in seg:
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"step {(b - a) / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, idle {(b - a - busy) / 1e6:.2f} ms, launches {len(seg)}")
zs = sorted(((e - s, s, e, q, g) for s, e, n, q, g in iv if n.startswith("zero_kernel") and s > marks[-3]), reverse=True)[:3]
for dur, s, e, q, g in zs:
    print(f"zero_kernel {dur / 1e3:.1f} us queue {q} grid {g}; overlapping / neighbouring kernels:")
    for s2, e2, n2, q2, g2 in iv:
        if e2 > s - 200000 and s2 < e + 100000 and (s2, e2) != (s, e):
            print(f"    [{(s2 - s) / 1e3:9.1f} .. {(e2 - s) / 1e3:9.1f}] us q{q2} {n2[:70]}")
P
rm -rf $OUT/kt

"""Histogram of ATen (torch-native) kernel launches by grid size from a rocprofv3 kernel_trace.csv."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
agg = defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"]
    if "at::" not in n:
        continue
    short = n.split("at::native::")[-1][:60] if "at::native::" in n else n[:60]
    if "CUDAFunctor_add" in n:
        short = "add"
    elif "FillFunctor" in n:
        short = "fill"
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    g = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r["Grid_Size"])
    bucket = 1 << max(0, g.bit_length() - 1)
    a = agg[(short, bucket)]
    a[0] += 1
    a[1] += dur
for (k, b), (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"{k:62s} grid>={b:10d} {c / steps:8.1f} calls/step {d / steps / 1e3:8.3f} ms/step")

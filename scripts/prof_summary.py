"""Print a per-kernel and per-family summary of a rocprofv3 --kernel-trace --stats kernel_stats.csv."""
import csv
import sys

f, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
fam = {}
for r in rows:
    n = r["Name"].replace("void ", "")
    key = ("gemm_fwd" if n.startswith("gemm_fwd") or n.startswith("gemm_thin") else
           "gemm_wgrad" if n.startswith("gemm_wgrad") else
           "lstm" if n.startswith("lstm") else
           "miopen/rocblas" if ("Cijk" in n or "LSTM" in n or "miopen" in n.lower()) else
           "gn/bn" if n.startswith(("gn_", "bn_")) else
           "fft" if n.startswith(("fft", "stft", "istft")) else
           "glu/act/add" if n.startswith(("glu", "act", "add_", "row_", "prelu")) else
           "loss" if n.startswith(("l1", "stft_loss", "sisdr")) else
           "pack" if n.startswith(("pack", "unpack")) else
           "optim" if n.startswith(("adamw", "sumsq", "clip")) else
           "aten" if "at::" in n else "other")
    fam[key] = fam.get(key, 0.0) + float(r["TotalDurationNs"])
for k, v in sorted(fam.items(), key=lambda kv: -kv[1]):
    print(f"{k:16s} {v / steps / 1e6:8.2f} ms/step {100 * v / tot:5.1f} %")
print(f"{'total':16s} {tot / steps / 1e6:8.2f} ms/step")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['TotalDurationNs']) / steps / 1e6:8.2f} ms/step "
          f"{float(r['AverageNs']) / 1e3:9.1f} us")

"""Per-family and per-kernel summary of a rocprofv3 --kernel-trace --stats kernel_stats.csv, as a markdown table; with the
PMC traffic JSON of the same command (scripts/collect_pmc.py) every kernel also gets its measured HBM bytes per launch and
the rate / fraction of the 8 TB/s peak they correspond to.  DESIGN.md section 6 is this script's output.

    python scripts/prof_summary.py <kernel_stats.csv> [steps=4] [rows=25] [pmc_traffic.json]
"""
import csv
import json
import sys

PEAK_HBM = 8.0e12

FAMILIES = [
    ("channels-last conv (fwd + dgrad)", ("cl_conv",)),
    ("channels-last weight gradient", ("cl_wgrad",)),
    ("channels-last fused DConv", ("cl_dconv",)),
    ("channels-last layout / sums / pack", ("cl_from_cm", "cl_to_cm", "cl_rowsum", "cl_pack", "cl_dgelu")),
    ("gemm forward family", ("gemm_fwd", "gemm_tap", "gemm_thin_fwd", "gemm_halo")),
    ("gemm weight gradient", ("gemm_wgrad", "gemm_thin_wgrad")),
    ("GroupNorm / BatchNorm", ("gn_", "bn_")),
    ("LSTM recurrence", ("lstm",)),
    ("GLU / activations / adds", ("glu", "act", "add_", "row_", "prelu", "channel_sum", "blstm_frames", "span_mask")),
    ("framed FFT", ("fft", "stft", "istft")),
    ("losses", ("l1", "stft_loss", "sisdr")),
    ("attention", ("localstate", "ls_mfma")),
    ("pack / unpack", ("pack", "unpack")),
    ("optimiser", ("adamw", "sumsq", "clip")),
    ("fused DConv layer", ("dconv_",)),
    ("zero fills (rfx_zero)", ("zero_kernel",)),
]


def family(name):
    n = name.replace("void ", "")
    for fam, prefixes in FAMILIES:
        if n.startswith(prefixes):
            return fam
    if "Cijk" in n or "miopen" in n.lower() or "rocblas" in n.lower():
        return "rocBLAS / MIOpen"
    if "at::" in n or "rocclr" in n or "elementwise_kernel" in n:
        return "ATen / runtime copies"
    return "other"


def _drop_first_launch_outlier(rows):
    """The profiled command starts cold (--preheat 0): a kernel's FIRST launch can carry a one-time cost inside its GPU duration
    (first touch of freshly mapped memory: cl_from_cm_kernel<float, 0> once took 25.9 ms against 60 us for its other 35 launches).
    The stats CSV only has Calls / Total / Min / Max, so: a Max above 2 ms AND above 50x the mean of the remaining launches is
    replaced by that mean.  Returns the names corrected (printed under the tables)."""
    fixed = []
    for r in rows:
        n, tot, mx = int(r["Calls"]), float(r["TotalDurationNs"]), float(r["MaxNs"])
        if n >= 4 and mx > 2e6 and mx > 50.0 * (tot - mx) / (n - 1):
            rest = (tot - mx) / (n - 1)
            fixed.append((r["Name"].replace("void ", "").split("(")[0], mx, rest))
            r["TotalDurationNs"] = str(tot - mx + rest)
            r["AverageNs"] = str((tot - mx + rest) / n)
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    return fixed


def main(path, steps=4.0, top=25, pmc=None):
    rows = list(csv.DictReader(open(path)))
    fixed = _drop_first_launch_outlier(rows)
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    traffic = json.load(open(pmc))["kernels"] if pmc else {}
    fam = {}
    for r in rows:
        f = family(r["Name"])
        a = fam.setdefault(f, [0.0, 0])
        a[0] += float(r["TotalDurationNs"]); a[1] += int(r["Calls"])
    print(f"kernel time {tot / steps / 1e6:.2f} ms per step over {sum(int(r['Calls']) for r in rows) / steps:.0f} launches "
          f"({path.split('/')[-1]}, {steps:g} steps profiled)\n")
    print("| family | ms / step | share | launches / step |\n|---|---|---|---|")
    for f, (ns, calls) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        print(f"| {f} | {ns / steps / 1e6:.2f} | {100 * ns / tot:.1f} % | {calls / steps:.0f} |")
    print("\n| kernel | launches / step | ms / step | avg us | HBM bytes / launch (PMC) | TB/s | of 8 TB/s |\n|---|---|---|---|---|---|---|")
    for r in rows[:top]:
        name = r["Name"].replace("void ", "")
        key = name.split("(")[0]
        avg = float(r["AverageNs"])
        t = traffic.get(key)
        if t:
            b = t["fetch_bytes_per_launch"] + t["write_bytes_per_launch"]
            extra = f"{b / 1e6:.1f} MB | {b / avg / 1e3:.2f} | {b / avg * 1e9 / PEAK_HBM:.2f}"
        else:
            extra = " | | "
        print(f"| `{key[:70]}` | {int(r['Calls']) / steps:.0f} | {float(r['TotalDurationNs']) / steps / 1e6:.2f} | {avg / 1e3:.1f} | {extra} |")
    for name, mx, rest in fixed:
        print(f"\n(first-launch outlier replaced by the mean of the other launches: `{name[:70]}` max {mx / 1e3:.0f} us against {rest / 1e3:.1f} us)")


if __name__ == "__main__":
    a = sys.argv
    main(a[1], float(a[2]) if len(a) > 2 else 4.0, int(a[3]) if len(a) > 3 else 25, a[4] if len(a) > 4 else None)

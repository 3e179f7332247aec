"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same bench command) into
the per-kernel HBM traffic summary bench.py reports as roofline.traffic.

gfx950 corrections (MI355X_MICROARCH.md section HBM, re-checked with scripts/pmc_calibrate.py on known byte
counts: a 1 GiB read reports FETCH_SIZE = 524 347 KB for dword and dwordx4 loads alike; a 1 GiB write reports
WRITE_SIZE = 1 048 576 KB): bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
usage: python scripts/collect_pmc.py <fetch_counter_csv> <write_counter_csv> <out_json> [steps_profiled]
"""
import collections
import csv
import json
import sys


def agg(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            d[k][0] += 1
            d[k][1] += float(row["Counter_Value"])
    return d


def main(fetch_csv, write_csv, out, steps=2):
    f, w = agg(fetch_csv, "FETCH_SIZE"), agg(write_csv, "WRITE_SIZE")
    res = {"note": "HBM-side bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 correction), "
                   "averaged over the launches of each kernel in the profiled bench run",
           "steps_profiled": steps, "kernels": {}}
    tot = 0.0
    for k in sorted(f, key=lambda k: -(2 * f[k][1] + w[k][1])):
        n = f[k][0]
        fb, wb = 2 * f[k][1] * 1024 / n, w[k][1] * 1024 / max(w[k][0], 1)
        tot += (2 * f[k][1] + w[k][1]) * 1024
        res["kernels"][k] = {"launches_per_step": n / steps, "fetch_bytes_per_launch": round(fb),
                             "write_bytes_per_launch": round(wb), "bytes_per_step": round((fb + wb) * n / steps)}
    res["total_bytes_per_step"] = round(tot / steps)
    json.dump(res, open(out, "w"), indent=1)
    print("total GB/step", tot / steps / 1e9)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 2)

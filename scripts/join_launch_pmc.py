"""Per-LAUNCH HBM traffic of the forward-family gather-GEMMs against their algorithmic bytes.

    python scripts/join_launch_pmc.py <launches.json from bench.py --dump-launches> <FETCH_SIZE counter csv> <WRITE_SIZE counter csv> [top]

The two rocprofv3 --pmc passes and the bench run execute the same deterministic launch sequence; the forward-family dispatches
(gemm_tap_kernel, gemm_tap_stream_kernel, gemm_fwd_kernel, gemm_thin_fwd_kernel) of the LAST profiled step are matched to the
launch list by order.  gfx950: bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (scripts/collect_pmc.py).
"""
import csv
import json
import sys

FAM = ("gemm_tap_kernel", "gemm_tap_stream_kernel", "gemm_fwd_kernel", "gemm_thin_fwd_kernel")


def series(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and any(f in r["Kernel_Name"] for f in FAM)]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    return [(r["Kernel_Name"].split("(")[0].replace("void ", ""), float(r["Counter_Value"])) for r in rows]


def main(lj, fcsv, wcsv, top=40):
    L = json.load(open(lj))
    f, w = series(fcsv, "FETCH_SIZE"), series(wcsv, "WRITE_SIZE")
    n = len(L)
    assert len(f) >= n and len(w) >= n and len(f) % n == 0, (len(f), len(w), n)
    f, w = f[-n:], w[-n:]
    out = []
    for d, (kn, fv), (_, wv) in zip(L, f, w):
        rd, wr = 2 * fv * 1024, wv * 1024
        out.append(dict(d, kernel=kn, read=rd, write=wr, amp=(rd + wr) / d["bytes"], excess=rd + wr - d["bytes"]))
    tot_alg, tot = sum(o["bytes"] for o in out), sum(o["read"] + o["write"] for o in out)
    print(f"{n} launches: algorithmic {tot_alg / 1e9:.1f} GB, measured {tot / 1e9:.1f} GB ({tot / tot_alg:.2f}x)\n")
    print("| kernel | M | K | N | OA x OB | alg MB | read MB | write MB | x | ms |\n|---|---|---|---|---|---|---|---|---|---|")
    for o in sorted(out, key=lambda o: -o["excess"])[:int(top)]:
        print(f"| `{o['kernel'][:34]}` | {o['M']} | {o['K']} | {o['N']} | {o['OA']} x {o['OB']} | {o['bytes'] / 1e6:.0f} | "
              f"{o['read'] / 1e6:.0f} | {o['write'] / 1e6:.0f} | {o['amp']:.2f} | {o['ms']:.3f} |")


if __name__ == "__main__":
    main(*sys.argv[1:])

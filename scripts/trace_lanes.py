"""Per-queue occupancy of a rocprofv3 kernel trace (kernel_trace.csv): for the last `steps` steps of the run, the busy time of every
HIP queue, the time both queues run kernels at once, the idle time of the device, and per kernel family the time it spends ALONE on
the device vs overlapped -- what `--stats` sums cannot show for a step that runs on two streams.

    python scripts/trace_lanes.py <kernel_trace.csv> <steps_profiled> [marker_kernel=adamw]
"""
import csv
import sys
from collections import defaultdict


def main(path, steps, marker="adamw"):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    rows.sort()
    marks = [e for s, e, n, q in rows if marker in n]
    if len(marks) < 3:
        print("marker kernel not found often enough")
        return
    t0, t1 = marks[-3], marks[-1]                 # the last two full steps
    nst = 2
    win = [(max(s, t0), min(e, t1), n, q) for s, e, n, q in rows if e > t0 and s < t1]
    span = (t1 - t0) / nst / 1e6
    # sweep
    ev = []
    for s, e, n, q in win:
        ev.append((s, 1, n, q)); ev.append((e, -1, n, q))
    ev.sort(key=lambda x: (x[0], x[1]))
    active = defaultdict(int)
    last = t0
    busy_by_q = defaultdict(float)
    both = idle = 0.0
    alone = defaultdict(float); shared = defaultdict(float)
    cur = {}
    for t, d, n, q in ev:
        dt = t - last
        if dt > 0:
            qs = [k for k, v in active.items() if v > 0]
            if not qs:
                idle += dt
            else:
                for k in qs:
                    busy_by_q[k] += dt
                if len(qs) > 1:
                    both += dt
                names = [nm for nm, c in cur.items() if c > 0]
                for nm in names:
                    (alone if len(names) == 1 else shared)[nm] += dt
        last = t
        active[q] += d
        cur[n] = cur.get(n, 0) + d
    print(f"window: {nst} steps, {span:.2f} ms / step; device idle {idle / nst / 1e6:.2f} ms / step; >1 queue busy {both / nst / 1e6:.2f} ms / step")
    for q, b in sorted(busy_by_q.items(), key=lambda kv: -kv[1]):
        print(f"  queue {q}: busy {b / nst / 1e6:.2f} ms / step")
    fam = defaultdict(lambda: [0.0, 0.0])
    for nm in set(alone) | set(shared):
        key = nm.replace("void ", "").split("(")[0][:60]
        fam[key][0] += alone.get(nm, 0.0); fam[key][1] += shared.get(nm, 0.0)
    # per kernel: launches in the window and the spread of their durations (a launch that waits for a concurrent kernel's workgroups
    # to drain shows up as a long maximum)
    dur = defaultdict(list)
    for s_, e_, n_, q_ in win:
        dur[n_.replace("void ", "").split("(")[0][:60]].append((e_ - s_) / 1e3)
    print("| kernel | launches | min us | median us | max us |\n|---|---|---|---|---|")
    for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:30]:
        v = sorted(v)
        print(f"| `{k}` | {len(v) // nst} | {v[0]:.0f} | {v[len(v) // 2]:.0f} | {v[-1]:.0f} |")
    print("| kernel | alone ms/step | overlapped ms/step |\n|---|---|---|")
    for k, (a, sh) in sorted(fam.items(), key=lambda kv: -(kv[1][0] + kv[1][1]))[:45]:
        print(f"| `{k}` | {a / nst / 1e6:.2f} | {sh / nst / 1e6:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4, *(sys.argv[3:4]))

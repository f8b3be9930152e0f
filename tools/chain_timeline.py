#!/usr/bin/env python
"""Per-queue view of the bench's steady state from a rocprofv3 --kernel-trace directory: for the last `window_ms` of the trace,
every HIP queue's busy time and idle time, and per kernel name on it the launches, mean duration and the mean idle time in front
of it (what the chain waits for).    python tools/chain_timeline.py <rocprofv3 -d dir> [window_ms=46]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    src = sys.argv[1]
    win = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 46e6
    rows = []
    for f in glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), n))
    rows.sort()
    l1 = [r[0] for r in rows if r[3].startswith("conv1_ft_wino4_kernel<1>")]
    if len(l1) >= 33:                    # two steady-state steps: 8 level-1 bulk launches per step, the last step left out
        t0, t1 = l1[-25], l1[-9]
        rows = [r for r in rows if t0 <= r[0] < t1]
        print("two steps between level-1 bulk launches: %.2f ms per step" % ((t1 - t0) / 2e6))
    else:
        t1 = rows[-1][1]
        rows = [r for r in rows if r[0] >= t1 - win]
    byq = defaultdict(list)
    for r in rows:
        byq[r[2]].append(r)
    print("window %.1f ms, %d launches" % ((t1 - rows[0][0]) / 1e6, len(rows)))
    for q, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, _, _ in rs)
        span = rs[-1][1] - rs[0][0]
        print("queue %s: %d launches, busy %.2f ms of %.2f ms span" % (q, len(rs), busy / 1e6, span / 1e6))
        st = defaultdict(lambda: [0, 0, 0])
        prev_end = None
        for s, e, _, n in rs:
            a = st[n[:44]]
            a[0] += 1
            a[1] += e - s
            if prev_end is not None:
                a[2] += max(0, s - prev_end)
            prev_end = max(prev_end or 0, e)
        for n, (c, d, g) in sorted(st.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
            print("    %-44s %4d x %7.1f us   idle in front %7.1f us   total %.2f + %.2f ms" % (n, c, d / c / 1e3, g / c / 1e3, d / 1e6, g / 1e6))


def raw(src, t_from_ms, t_len_ms):
    """raw listing: every launch that starts inside [t_from, t_from + t_len) ms after the first level-1 bulk launch of the
    third-last step"""
    rows = []
    for f in glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), n, r.get("Grid_Size", "")))
    rows.sort()
    l1 = [r[0] for r in rows if r[3].startswith("conv1_ft_wino4_kernel<1>")]
    t0 = l1[-25] + int(t_from_ms * 1e6)
    for s, e, q, n, g in rows:
        if t0 <= s < t0 + int(t_len_ms * 1e6):
            print("%9.1f us  +%7.1f us  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n[:40]))


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[2] == "raw":
        raw(sys.argv[1], float(sys.argv[3]), float(sys.argv[4]))
        sys.exit(0)
    main()

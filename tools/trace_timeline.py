#!/usr/bin/env python
"""Timeline of one training step from a rocprofv3 --kernel-trace directory: the step between the last two optimizer launches,
the span, the time at least one kernel runs (union over the queues), each queue's busy time, and the largest idle gaps with
the kernels around them.   python tools/trace_timeline.py <rocprofv3 -d dir> [marker substring]"""
import csv
import glob
import os
import sys


def main():
    src = sys.argv[1]
    markers = [sys.argv[2]] if len(sys.argv) > 2 else ["adam_table_kernel", "FusedOptimizer"]
    rows = []
    for f in glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if any(m in r[3] for m in markers)]
    # groups of consecutive marker launches = one optimizer step; take the region between the last two groups
    groups = []
    for i in marks:
        if groups and i - groups[-1][-1] < 40:
            groups[-1].append(i)
        else:
            groups.append([i])
    assert len(groups) >= 3, "need at least three optimizer steps in the trace"
    a, b = groups[-3][-1] + 1, groups[-2][-1] + 1
    step = rows[a:b]
    t0, t1 = step[0][0], max(r[1] for r in step)
    print(f"step: {len(step)} launches, span {(t1 - t0) / 1e6:.2f} ms")
    ev = sorted(step)
    busy, cur_s, cur_e, gaps = 0, ev[0][0], ev[0][1], []
    last_name = ev[0][3]
    for s, e, q, n in ev[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, last_name, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        if e >= cur_e:
            last_name = n
    busy += cur_e - cur_s
    print(f"at least one kernel running: {busy / 1e6:.2f} ms; idle {(t1 - t0 - busy) / 1e6:.2f} ms in {len(gaps)} gaps")
    per_q = {}
    for s, e, q, n in step:
        per_q.setdefault(q, [0, 0])
        per_q[q][0] += e - s
        per_q[q][1] += 1
    for q, (t, n) in sorted(per_q.items(), key=lambda kv: -kv[1][0]):
        print(f"  queue {q}: {n} launches, {t / 1e6:.2f} ms of kernel time")
    hist = {}
    for g, _, _ in gaps:
        k = "<5us" if g < 5e3 else "<20us" if g < 2e4 else "<100us" if g < 1e5 else ">=100us"
        hist.setdefault(k, [0, 0]); hist[k][0] += 1; hist[k][1] += g
    print("gaps:", {k: (v[0], round(v[1] / 1e6, 2)) for k, v in hist.items()})
    for g, before, after in sorted(gaps, reverse=True)[:12]:
        print(f"  {g / 1e3:8.1f} us after {before[:45]:45s} before {after[:45]}")
    # the step's kernels by total time (steady state: the data-dependent initialisation of the first step is not in it)
    per_k = {}
    for s, e, q, n in step:
        k = n.split("(")[0][:90]
        per_k.setdefault(k, [0, 0])
        per_k[k][0] += e - s
        per_k[k][1] += 1
    tot = sum(v[0] for v in per_k.values())
    print(f"kernels of the step: {len(step)} launches, {tot / 1e6:.2f} ms of kernel time")
    print(f"  {'kernel':90s} {'calls':>6s} {'total us':>10s} {'avg us':>8s} {'%':>6s}")
    for k, (t, c) in sorted(per_k.items(), key=lambda kv: -kv[1][0])[:30]:
        print(f"  {k:90s} {c:6d} {t / 1e3:10.1f} {t / c / 1e3:8.1f} {100.0 * t / tot:6.1f}")


if __name__ == "__main__":
    main()

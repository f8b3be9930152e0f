"""Experiment: the kernels (and memory copies) around the largest device-idle gaps of a bench trace.
    python tools/exp_gap_rows.py <rocprofv3 -d dir>"""
import csv
import glob
import os
import sys

src = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q" + r.get("Queue_Id", "?"), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:50]))
for f in glob.glob(os.path.join(src, "**", "*_memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY", r.get("Direction", "?")))
rows.sort()
last = max(i for i, r in enumerate(rows) if r[3].startswith('conv1_ft'))
rows = rows[len(rows) // 2:last]
t_end = 0
gaps = []
for i, (s, e, q, n) in enumerate(rows):
    if t_end and s - t_end > 300_000:
        gaps.append(i)
    t_end = max(t_end, e)
for gi in gaps[-3:]:
    print("---- gap before row", gi)
    base = rows[gi][0]
    for s, e, q, n in rows[max(0, gi - 12):gi + 8]:
        print(f"  {(s - base) / 1e3:10.1f} .. {(e - base) / 1e3:10.1f} us  {q:8s} {n}")

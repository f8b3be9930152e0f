#!/usr/bin/env python
"""Condense a rocprofv3 output directory into one small CSV per run (the raw traces are tens of MB).

usage: python tools/prof_summary.py <rocprofv3 -d dir> <out.csv> [--rm]

Reads every *_kernel_trace.csv (per-dispatch timestamps), *_kernel_stats.csv (rocprofv3's own --stats table)
and *_counter_collection.csv (--pmc) below the directory and writes, per kernel name:
    calls, total_ms, avg_us, min_us, max_us, pct   (from the trace; must agree with rocprofv3's stats table)
    <counter>_avg                                   (mean counter value per dispatch, one column per counter)
plus VGPR / LDS / grid of the first dispatch.  --rm deletes the raw directory afterwards.
`vgpr_rocprof` is rocprofv3's VGPR_Count column as it stands: on gfx950 it is HALF of the registers per lane the kernel allocates
(conv1_ft_wino4_kernel: 124 in round 2's trace, .vgpr_count 245 -> 248 allocated in the code object; 128 for this round's 256).
"""
import collections
import csv
import glob
import os
import shutil
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    return (name if cut < 0 else name[:cut])[:90]


def main():
    src, out = sys.argv[1], sys.argv[2]
    rm = "--rm" in sys.argv[3:]
    k = collections.OrderedDict()

    def rec(name):
        return k.setdefault(short(name), {"n": 0, "tot": 0, "min": None, "max": 0, "ctr": collections.defaultdict(list),
                                          "meta": None, "stats": None})
    for f in glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            e = rec(r["Kernel_Name"])
            e["n"] += 1
            e["tot"] += d
            e["min"] = d if e["min"] is None else min(e["min"], d)
            e["max"] = max(e["max"], d)
            if e["meta"] is None:
                e["meta"] = (r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", ""), r.get("LDS_Block_Size", ""),
                             "x".join(r.get(c, "") for c in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")),
                             r.get("Workgroup_Size_X", ""))
    for f in glob.glob(os.path.join(src, "**", "*_kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            rec(r["Name"])["stats"] = (r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
    counters = []
    seen = set()
    for f in glob.glob(os.path.join(src, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            key = (r["Dispatch_Id"], r["Counter_Name"])
            if key in seen:          # one row per (dispatch, counter, dimension instance): sum the instances
                pass
            e = rec(r["Kernel_Name"])
            e["ctr"][r["Counter_Name"]].append((r["Dispatch_Id"], float(r["Counter_Value"])))
            if r["Counter_Name"] not in counters:
                counters.append(r["Counter_Name"])
    total = sum(e["tot"] for e in k.values()) or 1
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct", "stats_calls", "stats_avg_ns",
                    "stats_pct", "vgpr_rocprof", "agpr", "lds_bytes", "grid", "wg"] + [c + "_avg_per_dispatch" for c in counters])
        for name, e in sorted(k.items(), key=lambda kv: -kv[1]["tot"]):
            row = [name, e["n"], "%.3f" % (e["tot"] / 1e6), "%.2f" % (e["tot"] / max(e["n"], 1) / 1e3),
                   "%.2f" % ((e["min"] or 0) / 1e3), "%.2f" % (e["max"] / 1e3), "%.2f" % (100.0 * e["tot"] / total)]
            row += list((e["stats"][0], e["stats"][2], e["stats"][3]) if e["stats"] else ("", "", ""))
            row += list(e["meta"] or ("", "", "", "", ""))
            for c in counters:
                per = collections.defaultdict(float)
                for disp, v in e["ctr"].get(c, []):
                    per[disp] += v
                row.append("%.1f" % (sum(per.values()) / len(per)) if per else "")
            w.writerow(row)
    if rm:
        shutil.rmtree(src, ignore_errors=True)
    print("wrote", out, "kernels:", len(k))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""profiles/traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/gpu_profile.sh (condensed by prof_summary.py).

usage: python tools/make_traffic.py <pmc_fetch.csv> <pmc_write.csv> <out.json> [tag used in the note, e.g. r01]

Units and corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: both counters are in KiB, separate
passes; FETCH_SIZE counts a 128-byte read request as 64 bytes, so it is doubled; WRITE_SIZE is taken as read."""
import csv
import json
import sys


def col(path, name):
    out = {}
    for r in csv.DictReader(open(path)):
        v = r.get(name + "_avg_per_dispatch")
        if v not in (None, ""):
            out[r["kernel"]] = (float(v), int(r["calls"]))
    return out


def main():
    fetch, write, dst = sys.argv[1:4]
    tag = sys.argv[4] if len(sys.argv) > 4 else "rNN"
    f, w = col(fetch, "FETCH_SIZE"), col(write, "WRITE_SIZE")
    # passes in the profiled run: wino4_input_kernel runs once per level and pass
    passes = {r["kernel"]: int(r["calls"]) for r in csv.DictReader(open(fetch))}.get("wino4_input_kernel", 0) // 2
    L1_STEPS, L1_PER_LAUNCH = 16, 8        # conditional steps of level 1 per pass; per launch of a pass started early (the steady state)
    res = {}
    for k in f:
        if not k.startswith("conv1_ft") or k not in w:
            continue
        per_dispatch = (2.0 * f[k][0] + w[k][0]) * 1024.0
        extra = {}
        if k.endswith("kernel<1>") and passes:
            # the profiled run mixes 4-step launches (its first pass) and 8-step launches: normalise by conditional steps
            per_step = per_dispatch * f[k][1] / (L1_STEPS * passes)
            extra = {"bytes_per_conditional_step": per_step, "mean_bytes_per_dispatch_in_profiled_run": per_dispatch,
                     "steps_per_launch": L1_PER_LAUNCH, "passes_in_profiled_run": passes}
            per_dispatch = per_step * L1_PER_LAUNCH
        res[k] = {
            **extra,
            "bytes_per_launch": per_dispatch,
            "fetch_size_kib_raw": f[k][0], "write_size_kib_raw": w[k][0], "dispatches": f[k][1],
            "note": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/{tag}_pmc_fetch.csv, "
                    f"{tag}_pmc_write.csv; tools/gpu_profile.sh, tools/make_traffic.py), per launch of 8 conditional steps x 2 networks at level 1 (the launches of a pass "
                    "started early, i.e. every pass but the first; total over the run / conditional steps x 8, because the profiled run's first pass launches 4 at a time), mean per dispatch at the other levels; "
                    "FETCH_SIZE doubled (gfx950 counts 128-B read requests as 64 B, MI355X_MICROARCH.md section HBM), "
                    "WRITE_SIZE as read; KiB -> bytes.  Infinity-Cache hits are included in FETCH_SIZE, so this is an upper "
                    "bound on HBM bytes.  Algorithmic bytes per level-1 launch (8 steps): conditioning features once (275 MB as "
                    "stored; 590 MB in the Winograd F(4x4,3x3) domain the default kernel reads) + weights (94 MB in the 6x6 "
                    "domain) + h2 slabs 839 MB = 1.52 GB (4 steps: 1.06 GB).  The excess is the conditioning data streamed again by the XCDs: an "
                    "XCD works on both networks of one step for half of the tile groups (the two workgroups of a group share "
                    "its images in the XCD's L2), so each of the 8 XCDs fetches half of the images for every 4 steps: 4 x 590 MB over the fabric per 4 steps "
                    "(8 x with one pair per XCD, the mapping before r02g); the XCDs walk the groups in step, so all but the first "
                    "read of a line hit the Infinity Cache, which FETCH_SIZE counts.  Round 3: conv1_ft_wino4_kernel<2> runs 6 steps per "
                    "launch, conv1_ft_wino_kernel<2> is the F(2x2,3x3) form of the last 4 conditional steps (ifl_set_conv1_tail)."}
    json.dump(res, open(dst, "w"), indent=1)
    print(json.dumps({k: v["bytes_per_launch"] for k, v in res.items()}))


if __name__ == "__main__":
    main()

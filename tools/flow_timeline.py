#!/usr/bin/env python
"""Timeline of the bench's steady-state steps from a rocprofv3 --kernel-trace directory of `bench.py`: per pass (encode,
decode) the span, the time the bulk kernels (conv1_ft*, conv3w F mode, wino4_input, load_cond) run, the time only chain
kernels run, the time nothing runs, and the idle gaps of the bulk stream with what ran meanwhile.
    python tools/flow_timeline.py <rocprofv3 -d dir> [n_last_steps=2]"""
import csv
import glob
import os
import sys


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def length(iv):
    return sum(e - s for s, e in iv)


def subtract(a, b):
    """a minus b, both unions"""
    out = []
    for s, e in a:
        cur = s
        for bs, be in b:
            if be <= cur or bs >= e:
                continue
            if bs > cur:
                out.append([cur, bs])
            cur = max(cur, be)
        if cur < e:
            out.append([cur, e])
    return out


def main():
    src = sys.argv[1]
    nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    rows = []
    for f in glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), n.split("(")[0]))
    rows.sort()
    is_bulk = lambda n: n.startswith("conv1_ft") or n.startswith("conv3w_kernel<0") or n.startswith("wino4_input") or "load_cond" in n or n.startswith("conv3f")
    # a pass starts with the conversion of its input (nchw_to_ring / squeeze kernels) -- use the first bulk-side kernel of
    # every pass instead: wino4_input launches come in pairs (two levels) per pass
    starts = [i for i, r in enumerate(rows) if r[3].startswith("wino4_input")]
    passes = starts[::2]
    if len(passes) < 2 * nlast + 1:
        raise SystemExit(f"only {len(passes)} passes in the trace")
    for pi in range(len(passes) - 2 * nlast - 1, len(passes) - 1):
        a, b = passes[pi], passes[pi + 1]
        seg = rows[a:b]
        t0, t1 = seg[0][0], rows[b][0]
        bulk = union([(s, min(e, t1)) for s, e, q, n in seg if is_bulk(n)])
        chain = union([(s, min(e, t1)) for s, e, q, n in seg if not is_bulk(n)])
        anyk = union([list(x) for x in bulk] + [list(x) for x in chain])
        only_chain = subtract(chain, bulk)
        idle = subtract([[t0, t1]], anyk)
        names = {}
        for s, e, q, n in seg:
            k = names.setdefault(n, [0, 0])
            k[0] += 1; k[1] += e - s
        print(f"pass {pi}: span {(t1 - t0) / 1e6:.3f} ms, {len(seg)} launches; bulk kernels running {length(bulk) / 1e6:.3f}, only chain kernels "
              f"{length(only_chain) / 1e6:.3f}, nothing {length(idle) / 1e6:.3f} ms")
        big = sorted(only_chain + idle, key=lambda x: x[0] - x[1])[:6]
        for s, e in sorted(big):
            inside = sorted({n for ks, ke, q, n in seg if ks < e and ke > s and not is_bulk(n)})
            print(f"    bulk stream idle {(e - s) / 1e3:8.1f} us at +{(s - t0) / 1e6:.3f} ms: {', '.join(x[:28] for x in inside[:5]) or '(nothing)'}")
        if pi == len(passes) - 2:
            for n, (c, d) in sorted(names.items(), key=lambda x: -x[1][1])[:12]:
                print(f"    {n[:60]:60s} {c:4d} launches {d / 1e6:8.3f} ms (sum of durations)")


if __name__ == "__main__":
    main()

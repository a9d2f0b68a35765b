#!/usr/bin/env python3
"""Aggregates a rocprofv3 --kernel-trace CSV by (kernel, grid): launches and average duration.
   python tools/trace_by_grid.py <dir-with-*_kernel_trace.csv> [top-N]"""
import collections, csv, glob, os, re, sys

def main():
    d = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"])
        name = re.sub(r"^void ", "", name).replace("sts::", "")
        key = (name[:48], r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?"))
        a = agg[key]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot = sum(a[1] for a in agg.values())
    print(f"{'kernel':48s} {'grid':>22s} {'calls':>6s} {'avg_us':>9s} {'share':>6s}")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{k[0]:48s} {k[1]+'x'+k[2]+'x'+k[3]:>22s} {a[0]:6d} {a[1]/a[0]:9.1f} {100*a[1]/tot:5.1f}%")

if __name__ == "__main__":
    main()

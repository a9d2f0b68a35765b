#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc counter_collection CSVs by (kernel, grid): per-launch average of every counter.
   python tools/pmc_by_grid.py <dir> [<dir> ...] [--match substr]"""
import collections, csv, glob, os, re, sys

def main():
    dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = None
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
        dirs = [d for d in dirs if d != match]
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    dur = collections.defaultdict(lambda: [0, 0.0])
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            seen = set()
            for r in csv.DictReader(open(f)):
                name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("sts::", "")
                if match and match not in name:
                    continue
                key = (name[:44], r.get("Grid_Size", "?"))
                a = agg[key][r["Counter_Name"]]
                a[0] += 1; a[1] += float(r["Counter_Value"])
                if "Start_Timestamp" in r and r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"])
                    dd = dur[key]; dd[0] += 1; dd[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for key, cs in sorted(agg.items(), key=lambda kv: -dur[kv[0]][1]):
        n, t = dur[key]
        print(f"{key[0]} grid={key[1]} launches={n} avg_us={t/max(1,n):.1f}")
        for c, (k, v) in sorted(cs.items()):
            print(f"    {c:34s} {v/k:16.1f}")

if __name__ == "__main__":
    main()

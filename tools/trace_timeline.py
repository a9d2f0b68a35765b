#!/usr/bin/env python3
"""Prints ONE step of a rocprofv3 --kernel-trace CSV as a timeline: start offset, duration and the idle gap to the previous
kernel's end, in dispatch order -- what the per-kernel averages cannot show (dependent-launch gaps, host stalls).
   python tools/trace_timeline.py <dir-with-*_kernel_trace.csv> [step-index-from-the-end (default 1)] [first-kernel-substr (default embed_kernel)]"""
import csv, glob, os, re, sys


def main():
    d = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    first = sys.argv[3] if len(sys.argv) > 3 else "embed_kernel"
    f = max(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
    if len(starts) < back + 1:
        print("not enough steps in the trace"); return
    i0, i1 = starts[-back - 1], starts[-back]
    t0 = int(rows[i0]["Start_Timestamp"])
    prev_end = t0
    busy = gaps = 0.0
    print(f"{'#':>4s} {'t_us':>9s} {'dur_us':>8s} {'gap_us':>7s}  kernel [grid]")
    for k, r in enumerate(rows[i0:i1]):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("sts::", "")
        gap = (s - prev_end) / 1e3
        print(f"{k:4d} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f}  {name[:60]} [{r.get('Grid_Size_X','?')}x{r.get('Grid_Size_Y','?')}x{r.get('Grid_Size_Z','?')} wg {r.get('Workgroup_Size_X','?')}]")
        busy += (e - s) / 1e3
        gaps += max(0.0, gap)
        prev_end = max(prev_end, e)
    print(f"step span {(prev_end - t0) / 1e3:.1f} us, kernel time {busy:.1f} us, idle gaps {gaps:.1f} us, {i1 - i0} kernels")


if __name__ == "__main__":
    main()

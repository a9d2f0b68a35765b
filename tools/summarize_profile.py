#!/usr/bin/env python3
"""Turns rocprofv3 output under gpurun_out/ into the committed summaries under profiles/.

  python tools/summarize_profile.py <round-tag> <kernel-trace-dir> [<pmc-fetch-dir> <pmc-write-dir> <steps-in-pmc-run> [<pmc-sq-dir>]]

Writes profiles/<tag>_kernel_stats.csv (the --stats table), and, when PMC passes are given,
profiles/<tag>_pmc.json + profiles/latest_pmc.json with the per-launch HBM traffic of the dominant
kernel family (conv_mfma_kernel).  FETCH_SIZE / WRITE_SIZE are in KB; per MI355X_MICROARCH.md (HBM
section) FETCH_SIZE on gfx950 under-reports wide coalesced reads by 2x, so the read side is doubled;
WRITE_SIZE is uncalibrated and used as reported."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(pattern):
    files = glob.glob(pattern) + glob.glob(pattern.replace(os.sep + "*" + os.sep, os.sep))   # with or without the host sub-directory
    return max(files, key=os.path.getmtime)


def counter_sum(d, family):
    f = newest(os.path.join(d, "*", "*counter_collection.csv"))
    n, tot = 0, 0.0
    for r in csv.DictReader(open(f)):
        if any(f in r["Kernel_Name"] for f in family):
            n += 1
            tot += float(r["Counter_Value"])
    return n, tot


def main():
    tag, kt = sys.argv[1], sys.argv[2]
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    st = newest(os.path.join(kt, "*", "*kernel_stats.csv"))
    shutil.copy(st, os.path.join(out, f"{tag}_kernel_stats.csv"))
    # the LDS-staged matrix-core conv: single (upsamplers), grouped (ResBlock chains) and fused-layer (narrow stages) forms
    fam = ("conv_mfma_kernel", "conv_mfma_group_kernel", "resblock_layer_kernel", "resblock_wino_kernel")
    rows = [r for r in csv.DictReader(open(st)) if any(f in r["Name"] for f in fam)]
    calls = sum(int(r["Calls"]) for r in rows)
    tot_ns = sum(float(r["TotalDurationNs"]) for r in rows)
    summary = {"tag": tag, "kernel_family": " + ".join(fam), "calls": calls, "avg_launch_us_rocprof": tot_ns / max(1, calls) / 1e3}
    if len(sys.argv) >= 6:
        nf, fetch_kb = counter_sum(sys.argv[3], fam)
        nw, write_kb = counter_sum(sys.argv[4], fam)
        summary.update({
            "fetch_size_kb_per_launch_raw": fetch_kb / max(1, nf),
            "write_size_kb_per_launch_raw": write_kb / max(1, nw),
            "hbm_bytes_per_launch_corrected": (2.0 * fetch_kb / max(1, nf) + write_kb / max(1, nw)) * 1024.0,
            "correction": "read side x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B); WRITE_SIZE as reported",
            "launches_in_pmc_run": nf,
        })
        json.dump(summary, open(os.path.join(out, "latest_pmc.json"), "w"), indent=1)
    if len(sys.argv) >= 7:      # SQ pass: MFMA-busy of the family (kernels are serialised under --pmc)
        f = newest(os.path.join(sys.argv[6], "*", "*counter_collection.csv"))
        disp = {}
        for r in csv.DictReader(open(f)):
            if not any(f in r["Kernel_Name"] for f in fam):
                continue
            d = disp.setdefault(r["Dispatch_Id"], {"t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
            d[r["Counter_Name"]] = float(r["Counter_Value"])
        busy = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for d in disp.values())
        gui = sum(d.get("GRBM_GUI_ACTIVE", 0.0) for d in disp.values()) / 8.0     # one copy per XCD
        tns = sum(d["t"] for d in disp.values())
        summary.update({
            "mfma_busy_pct_serialised": 100.0 * busy / (1024.0 * gui) if gui else None,   # 256 CUs x 4 SIMDs
            "shader_clock_ghz": gui / tns if tns else None,
            "avg_launch_us_serialised": tns / max(1, len(disp)) / 1e3,
            "hbm_gbps_serialised": summary.get("hbm_bytes_per_launch_corrected", 0.0) / (tns / max(1, len(disp))) if tns else None,
            "note_serialised": "rocprofv3 --pmc serialises dispatches: dispatches run strictly one after another here",
        })
        json.dump(summary, open(os.path.join(out, "latest_pmc.json"), "w"), indent=1)
    json.dump(summary, open(os.path.join(out, f"{tag}_summary.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()

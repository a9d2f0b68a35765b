#!/usr/bin/env python3
"""Turns rocprofv3 output (under gpurun_out/) into the committed summaries under profiles/.

  python tools/summarize_profile.py <tag> --workload "<key>" --kt <kernel-trace-dir>
         [--fetch <pmc FETCH_SIZE dir>] [--write <pmc WRITE_SIZE dir>] [--sq <pmc SQ dir>] [--flow-launches N]

<key> is bench.py's workload key ("hifigan_sdp|batch=1|phonemes=128|ragged=0").  Writes
  profiles/<tag>_kernel_stats.csv   the rocprofv3 --stats table of the kernel-trace run
  profiles/<tag>_timeline.txt       one step of that trace in dispatch order (tools/trace_timeline.py)
  profiles/<tag>_summary.json       decoder-trunk family (the matrix-core launches bench.py times) and reverse-flow launches:
                                    launches per step, average duration, HBM bytes per launch from the PMC passes, MFMA-busy
  profiles/latest_pmc.json          {"by_workload": {<key>: summary}} -- bench.py attaches a summary to its `roofline` object only
                                    when the workload key AND the kernel build id (sha256 of summertts_amd/csrc) match.

Kernel classes are found by DISPATCH ORDER inside a step: the flow_layer / flow_finish launches (round 4; before: the
`--flow-launches` convs) after a step's expand_frames_kernel are the reverse flow, the next launch is conv_pre, and everything up to
the step's last ResBlock-layer launch (sum_scale launches excluded) is the decoder trunk -- the upsamplers and the grouped / fused ResBlock layers, whichever kernel
variant the dispatcher picked for them.

FETCH_SIZE / WRITE_SIZE are in KB; per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports half the bytes of wide
coalesced reads, so the read side is doubled; WRITE_SIZE is uncalibrated and used as reported.  The two counters need separate
passes (TCC slots), and --pmc runs serialise dispatches."""
import argparse
import csv
import glob
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(d, suffix):
    files = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    if not files:
        raise SystemExit(f"no *{suffix} under {d}")
    return max(files, key=os.path.getmtime)


def kernel_build_id():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "summertts_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").replace("sts::", "")


def classify(names, flow_launches):
    """names: kernel names of ONE run in dispatch order -> list of labels ('flow', 'trunk', other)."""
    lab = ["other"] * len(names)
    # a step's acoustic half starts at its (single) expand_frames_kernel; the region runs to the next step's (the embedding
    # kernel is not a usable marker: at batch it is fused into the first projection)
    starts = [i for i, n in enumerate(names) if "expand_frames_kernel" in n] + [len(names)]
    for s, e in zip(starts[:-1], starts[1:]):
        i = s + 1
        if any("flow_layer_kernel" in names[k] for k in range(s + 1, e)):
            # round 4: the reverse flow = one launch per WaveNet layer (wn_flow.hip) + the finishing kernel
            while i < e and ("flow_layer_kernel" in names[i] or "flow_finish_kernel" in names[i] or "conv_" in names[i] and "flow_layer_kernel" in " ".join(names[i:i + 3])):
                lab[i] = "flow"
                i += 1
        else:
            nconv = 0
            while i < e and nconv < flow_launches:
                if "conv_" in names[i]:
                    lab[i] = "flow"
                    nconv += 1
                i += 1
        pre = i                                              # conv_pre
        # the trunk ends with the last ResBlock layer launch of the step (fused layer / grouped conv) -- or the sum_scale behind it where
        # the chain mean is still a launch of its own; conv_post and the tail follow
        ends = [k for k in range(pre, e) if "resblock_" in names[k] or "_group_kernel" in names[k] or "sum_scale_kernel" in names[k]]
        if not ends:
            continue
        for k in range(pre + 1, ends[-1] + 1):
            if "sum_scale_kernel" not in names[k] and "split_planes_kernel" not in names[k] and "rocclr" not in names[k]:      # (the entry split of the pre-split path is not a matrix-core launch: bench.py does not count it either, its time is inside the region)
                lab[k] = "trunk"
    return lab


def pmc_rows(d):
    """-> dispatches in order: [{name, t_ns, counters{}}]"""
    f = newest(d, "counter_collection.csv")
    disp = {}
    for r in csv.DictReader(open(f)):
        k = int(r["Dispatch_Id"])
        x = disp.setdefault(k, {"name": short(r["Kernel_Name"]), "t_ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "c": {}})
        x["c"][r["Counter_Name"]] = x["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return [disp[k] for k in sorted(disp)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--workload", required=True)
    ap.add_argument("--kt", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--sq")
    ap.add_argument("--flow-launches", type=int, default=40, help="convs of the reverse flow per step: n_flows * (2 + 2 * wn_layers) [+ n_flows cond convs]")
    a = ap.parse_args()
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    try:
        shutil.copy(newest(a.kt, "kernel_stats.csv"), os.path.join(out, f"{a.tag}_kernel_stats.csv"))
    except SystemExit:
        pass
    tl = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_timeline.py"), a.kt], capture_output=True, text=True).stdout
    open(os.path.join(out, f"{a.tag}_timeline.txt"), "w").write(tl)

    rows = sorted(csv.DictReader(open(newest(a.kt, "kernel_trace.csv"))), key=lambda r: int(r["Start_Timestamp"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    lab = classify(names, a.flow_launches)
    nsteps = max(1, sum(1 for n in names if "expand_frames_kernel" in n))
    summary = {"tag": a.tag, "workload": a.workload, "kernel_build_id": kernel_build_id(), "steps_in_trace": nsteps}
    for cls in ("trunk", "flow"):
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r, l in zip(rows, lab) if l == cls]
        kinds = sorted({n for n, l in zip(names, lab) if l == cls})
        summary[cls] = {"launches_per_step": len(d) / nsteps, "avg_launch_us_rocprof": sum(d) / max(1, len(d)),
                        "us_per_step_rocprof": sum(d) / nsteps, "kernels": kinds}

    def per_class(d, counter):
        pr = pmc_rows(d)
        pl = classify([x["name"] for x in pr], a.flow_launches)
        res = {}
        for cls in ("trunk", "flow"):
            sel = [x for x, l in zip(pr, pl) if l == cls]
            res[cls] = (len(sel), sum(x["c"].get(counter, 0.0) for x in sel), sum(x["t_ns"] for x in sel), sel)
        return res

    if a.fetch and a.write:
        fe, wr = per_class(a.fetch, "FETCH_SIZE"), per_class(a.write, "WRITE_SIZE")
        for cls in ("trunk", "flow"):
            nf, fkb, _, _ = fe[cls]
            nw, wkb, _, _ = wr[cls]
            summary[cls].update({
                "fetch_size_kb_per_launch_raw": fkb / max(1, nf), "write_size_kb_per_launch_raw": wkb / max(1, nw),
                "hbm_bytes_per_launch_corrected": (2.0 * fkb / max(1, nf) + wkb / max(1, nw)) * 1024.0,
                "launches_in_pmc_run": nf})
        summary["correction"] = "read side x2 (gfx950 FETCH_SIZE counts 128-B requests as 64 B); WRITE_SIZE as reported"
    if a.sq:
        sq = per_class(a.sq, "SQ_VALU_MFMA_BUSY_CYCLES")
        for cls in ("trunk", "flow"):
            n, busy, tns, sel = sq[cls]
            gui = sum(x["c"].get("GRBM_GUI_ACTIVE", 0.0) for x in sel) / 8.0          # one copy per XCD
            summary[cls].update({
                "mfma_busy_pct_serialised": 100.0 * busy / (1024.0 * gui) if gui else None,     # 256 CUs x 4 SIMDs
                "shader_clock_ghz": gui / tns if tns else None,
                "avg_launch_us_serialised": tns / max(1, n) / 1e3})
            if "hbm_bytes_per_launch_corrected" in summary[cls] and tns:
                summary[cls]["hbm_gbps_serialised"] = summary[cls]["hbm_bytes_per_launch_corrected"] / (tns / max(1, n))
        summary["note_serialised"] = "rocprofv3 --pmc serialises dispatches: every launch runs alone here"
    json.dump(summary, open(os.path.join(out, f"{a.tag}_summary.json"), "w"), indent=1)

    # what bench.py reads
    latest_path = os.path.join(out, "latest_pmc.json")
    try:
        latest = json.load(open(latest_path))
        if "by_workload" not in latest:
            latest = {"by_workload": {}}
    except Exception:
        latest = {"by_workload": {}}
    if "hbm_bytes_per_launch_corrected" in summary["trunk"]:
        t = summary["trunk"]
        entry = {"tag": a.tag, "kernel_build_id": summary["kernel_build_id"], "correction": summary["correction"],
                 "hbm_bytes_per_launch_corrected": t["hbm_bytes_per_launch_corrected"], "avg_launch_us_rocprof": t["avg_launch_us_rocprof"],
                 "flow": {k: summary["flow"].get(k) for k in ("launches_per_step", "us_per_step_rocprof", "hbm_bytes_per_launch_corrected",
                                                              "hbm_gbps_serialised", "mfma_busy_pct_serialised")}}
        for k in ("mfma_busy_pct_serialised", "hbm_gbps_serialised"):
            if k in t:
                entry[k] = t[k]
        latest["by_workload"][a.workload] = entry
        json.dump(latest, open(latest_path, "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""One summary line per bench.py JSON file:  python tools/bench_line.py a.json b.json ..."""
import json
import sys
for p in sys.argv[1:]:
    try:
        d = json.loads([ln for ln in open(p) if ln.startswith("{")][-1])
        s = d["stage_ms_per_step"]; r = d["roofline"]
        print(f"{p.split('/')[-1]:28s} ms/step {d['ms_per_step']:.3f} xRT {d['x_realtime_16khz']:.0f} te {s['text_encoder']:.3f} dp {s['duration']:.3f} flow {s['flow']:.3f} dec {s['decoder']:.3f} "
              f"trunk {r.get('launches_per_step', 0):.0f} launches x {r.get('avg_launch_us', 0):.1f} us = {r.get('launches_per_step', 0) * r.get('avg_launch_us', 0) / 1e3:.3f} ms, {r['achieved']:.1f} TF ({r['frac']:.3f}) sync {d.get('host_sync_wait_ms_per_step', 0):.3f}")
        for k in ("sustained", "api_call_leg", "stage_breakdown_leg"):
            if d.get(k):
                print(f"    {k}: " + ", ".join(f"{a}={b:.4g}" for a, b in d[k].items() if isinstance(b, (int, float))))
        for k, v in (d.get("configs") or {}).items():
            if "error" in v:
                print(f"    {k}: ERROR {v['error']}")
            else:
                print(f"    {k}: ms/step {v['ms_per_step']:.2f} xRT {v['x_realtime_16khz']:.0f} trunk frac {v['roofline']['frac']:.3f} stages {v.get('stage_ms_per_step')} sync {v.get('host_sync_wait_ms_per_step', 0):.3f} parity {json.dumps(v.get('parity'))[:220]}")
        if d.get("parity"):
            print("    parity:", json.dumps(d["parity"])[:400])
        if d.get("cpu_baseline"):
            print("    cpu:", json.dumps(d["cpu_baseline"])[:200])
    except Exception as e:   # noqa: BLE001
        print(p, "FAILED", e)

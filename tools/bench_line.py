#!/usr/bin/env python3
"""One summary line per bench.py JSON file:  python tools/bench_line.py a.json b.json ..."""
import json
import sys
for p in sys.argv[1:]:
    try:
        d = json.loads([ln for ln in open(p) if ln.startswith("{")][-1])
        s = d["stage_ms_per_step"]; r = d["roofline"]
        print(f"{p.split('/')[-1]:28s} ms/step {d['ms_per_step']:.3f} xRT {d['x_realtime_16khz']:.0f} te {s['text_encoder']:.3f} dp {s['duration']:.3f} flow {s['flow']:.3f} dec {s['decoder']:.3f} "
              f"trunk {r.get('launches', 0)} launches x {r.get('avg_launch_us', 0):.1f} us = {r.get('launches', 0) * r.get('avg_launch_us', 0) / 1e3:.3f} ms, {r['achieved']:.1f} TF ({r['frac']:.3f}) sync {d.get('host_sync_wait_ms_per_step', 0):.3f}")
    except Exception as e:   # noqa: BLE001
        print(p, "FAILED", e)

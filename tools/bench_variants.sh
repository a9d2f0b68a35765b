#!/bin/bash
# Runs bench.py (no CPU baseline, no pool) under several experiment knobs and prints one summary line per variant.
# usage: tools/bench_variants.sh <out-dir> "<label>|<env assignments>|<bench args>" ...
out=$1; shift
mkdir -p "$out"
for spec in "$@"; do
  label=${spec%%|*}; rest=${spec#*|}; envs=${rest%%|*}; args=${rest#*|}
  env $envs python bench.py --no-cpu-baseline --no-f32-leg --pipeline-engines 0 --steps 30 --warmup 5 $args > "$out/$label.json" 2> "$out/$label.err"
  python - "$out/$label.json" "$label" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    s = d["stage_ms_per_step"]; r = d["roofline"]
    print(f"{sys.argv[2]:28s} ms/step {d['ms_per_step']:.3f} xRT {d['x_realtime_16khz']:.0f} te {s['text_encoder']:.3f} dp {s['duration']:.3f} flow {s['flow']:.3f} dec {s['decoder']:.3f} "
          f"mfma {r['achieved']:.1f} TF ({r['frac']:.3f}) bf16 {r.get('bf16_issued_frac', 0):.3f} f32i {r['mfma_issued_frac']:.3f} sync {d['host_sync_wait_ms_per_step']:.3f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done

#!/bin/bash
# bench.py under lab knobs (lab build: make -C summertts_amd/csrc exp).  usage: tools/knob_sweep.sh "<label>|<env assignments>|<bench args>" ...
out=gpurun_out/knob_sweep; mkdir -p $out
export SUMMERTTS_HIP_LIB=summertts_amd/lib/exp_knobs/libsummertts_hip.so
for spec in "$@"; do
  label=${spec%%|*}; rest=${spec#*|}; envs=${rest%%|*}; args=${rest#*|}
  env $envs timeout 150 python bench.py --no-cpu-baseline --no-f32-leg --pipeline-engines 0 --steps 20 --warmup 5 $args > $out/$label.json 2> $out/$label.err
  python - $out/$label.json "$label" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); s = d["stage_ms_per_step"]; r = d["roofline"]
    print(f"knob {sys.argv[2]:24s} ms/step {d['ms_per_step']:.3f} dec {s['decoder']:.3f} trunk {r['avg_launch_us']*r['launches_per_step']/1e3:.3f} ms  {r['achieved']:.1f} TF  launches {r['launches_per_step']:.0f}")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done

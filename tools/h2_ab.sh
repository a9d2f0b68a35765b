#!/bin/bash
# A/B of library variants under the two-term fp16 arithmetic (one utterance, config 1): tools/h2_ab.sh base 6w3 ...
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/h2ab
for v in "$@"; do
  if [ "$v" = base ]; then lib=summertts_amd/lib/libsummertts_hip.so; else lib=summertts_amd/lib/var/libvar$v.so; fi
  SUMMERTTS_HIP_LIB=$lib timeout 100 python bench.py ${H2AB_ARGS:---config 1} --conv-math ${H2AB_MATH:-f16x2} --no-cpu-baseline --no-f32-leg --pipeline-engines 0 --steps ${H2AB_STEPS:-30} --warmup 5 > gpurun_out/h2ab/v$v.json 2> gpurun_out/h2ab/v$v.err
  python - gpurun_out/h2ab/v$v.json $v <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); s = d["stage_ms_per_step"]; r = d["roofline"]
    print(f"var {sys.argv[2]:6s} ms/step {d['ms_per_step']:.3f} dec {s['decoder']:.3f} trunk {r['avg_launch_us']*r['launches_per_step']/1e3:.3f} ms  {r['achieved']:.1f} TF")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done

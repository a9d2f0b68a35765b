#!/bin/bash
# Per-stage tile sweep of the grouped split-bf16 launches at one utterance (lab build with the knobs: make -C summertts_amd/csrc exp).
# STS_BF3_GROUP_TILE = one tile code per decoder stage ('-' = automatic); codes: 0 128x128  1 64x256  2 128x256w8  3 64x128  4 32x256
# 5 32x128  6 128x128 as 4 waves of 32x128  7 64x128 as 2 waves of 32x128  8..f = 0..7 with 32-channel staged chunks
out=gpurun_out/tile_sweep; mkdir -p $out
export SUMMERTTS_HIP_LIB=summertts_amd/lib/exp_knobs/libsummertts_hip.so
for t in "$@"; do
  STS_BF3_GROUP_TILE=$t timeout 100 python bench.py --no-cpu-baseline --no-f32-leg --pipeline-engines 0 --steps 30 --warmup 5 > $out/t$t.json 2> $out/t$t.err
  python - $out/t$t.json "$t" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); s = d["stage_ms_per_step"]; r = d["roofline"]
    print(f"tiles {sys.argv[2]:6s} ms/step {d['ms_per_step']:.3f} dec {s['decoder']:.3f} trunk {r['avg_launch_us']*r['launches_per_step']/1e3:.3f} ms  {r['achieved']:.1f} TF")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done

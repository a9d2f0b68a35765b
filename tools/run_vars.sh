mkdir -p gpurun_out/r03h
for v in ${VARS:-base 6 22 30 base 22}; do
  if [ "$v" = base ]; then lib=summertts_amd/lib/libsummertts_hip.so; else lib=summertts_amd/lib/var/libvar$v.so; fi
  SUMMERTTS_HIP_LIB=$lib timeout 100 python bench.py --no-cpu-baseline --no-f32-leg --pipeline-engines 0 --steps 30 --warmup 5 > gpurun_out/r03h/v$v.json 2> gpurun_out/r03h/v$v.err
  python - gpurun_out/r03h/v$v.json $v <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); s = d["stage_ms_per_step"]; r = d["roofline"]
    print(f"var {sys.argv[2]:5s} ms/step {d['ms_per_step']:.3f} dec {s['decoder']:.3f} trunk {r['avg_launch_us']*r['launches_per_step']/1e3:.3f} ms  {r['achieved']:.1f} TF")
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done
SUMMERTTS_HIP_LIB=summertts_amd/lib/var/libvar${TESTVAR:-22}.so timeout 200 python -m pytest tests/test_parity_gpu.py -q -x -k "golden or bf3" 2>&1 | tail -3

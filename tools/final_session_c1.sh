# the config-1 half of tools/final_session.sh: profiling passes + the full default bench line (for a change that touches no trunk kernel)
mkdir -p gpurun_out/r03z
bash tools/profile_session.sh gpurun_out/r03z/c1 > gpurun_out/r03z/c1.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03z/bench_c1.json 2> gpurun_out/r03z/bench_c1.err
python - gpurun_out/r03z/bench_c1.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); s = d["stage_ms_per_step"]; r = d["roofline"]
print("c1", f"ms/step {d['ms_per_step']:.3f} xRT {d['x_realtime_16khz']:.0f} trunk {r['achieved']:.1f} TF frac {r['frac']:.3f}", "parity", json.dumps(d.get("parity"))[:200], d["config"]["kernel_build_id"])
PY

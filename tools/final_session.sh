mkdir -p gpurun_out/r03z
bash tools/profile_session.sh gpurun_out/r03z/c1 > gpurun_out/r03z/c1.log 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r03z/bench_c1.json 2> gpurun_out/r03z/bench_c1.err
timeout 200 python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline --pipeline-engines 0 > gpurun_out/r03z/bench_c2.json 2> gpurun_out/r03z/bench_c2.err
timeout 300 python bench.py --config 3 --steps 5 --warmup 2 --cpu-reps 1 --cpu-threads 16 --pipeline-engines 0 > gpurun_out/r03z/bench_c3.json 2> gpurun_out/r03z/bench_c3.err
timeout 300 python bench.py --config 4 --steps 5 --warmup 2 --cpu-reps 1 --cpu-threads 16 --pipeline-engines 0 > gpurun_out/r03z/bench_c4.json 2> gpurun_out/r03z/bench_c4.err
STS_BENCH_FORCE_DIST=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r03z/bench_c1_rccl1rank.json 2> gpurun_out/r03z/bench_c1_rccl1rank.err
for f in c1 c2 c3 c4 c1_rccl1rank; do python - gpurun_out/r03z/bench_$f.json $f <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); s = d["stage_ms_per_step"]; r = d["roofline"]
    print(sys.argv[2], f"ms/step {d['ms_per_step']:.3f} xRT {d['x_realtime_16khz']:.0f} stages {s} trunk {r['achieved']:.1f} TF frac {r['frac']:.3f}", "parity", json.dumps(d.get("parity"))[:400])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done

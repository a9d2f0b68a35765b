# Round-end evidence at the current HEAD (one gpurun call): profiling passes of config 1, the full bench lines of configs 1-4 (each with
# cpu_baseline + parity), the one-rank RCCL line, the power / clock trace, a tile trace.  Results under gpurun_out/$TAG; copy with
# tools/collect_final.sh.     usage: bash tools/final_session.sh [tag]
TAG=${1:-r04z}
O=gpurun_out/$TAG
mkdir -p $O
bash tools/profile_session.sh $O/c1 > $O/c1.log 2>&1
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_c1.json 2> $O/bench_c1.err
timeout 400 python bench.py --config 2 --steps 5 --warmup 2 --cpu-reps 1 --cpu-threads 16 --pipeline-engines 0 > $O/bench_c2.json 2> $O/bench_c2.err
timeout 400 python bench.py --config 3 --steps 5 --warmup 2 --cpu-reps 1 --cpu-threads 16 --pipeline-engines 0 > $O/bench_c3.json 2> $O/bench_c3.err
timeout 400 python bench.py --config 4 --steps 5 --warmup 2 --cpu-reps 1 --cpu-threads 16 --pipeline-engines 0 > $O/bench_c4.json 2> $O/bench_c4.err
STS_BENCH_FORCE_DIST=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --min-seconds 0 > $O/bench_c1_rccl1rank.json 2> $O/bench_c1_rccl1rank.err
timeout 200 python tools/power_trace.py 5 > $O/power_trace.log 2>&1
# same-box A/B of the PCM hand-over (the last kernel writes into the pinned host buffer vs a download behind it)
for v in 1 0 1 0; do timeout 100 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-f32-leg --pipeline-engines 0 --min-seconds 0 --debug-set pcm_direct=$v 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('pcm_direct=$v', round(d['ms_per_step'],4), d['stage_ms_per_step'], d['host_us_per_step']['step_wall_minus_device_stages'])" >> $O/pcm_direct_ab.txt; done
cat $O/pcm_direct_ab.txt
if [ -f summertts_amd/lib/var/libvar6tt.so ]; then
  SUMMERTTS_HIP_LIB=summertts_amd/lib/var/libvar6tt.so timeout 200 python tools/tile_trace_dump.py $O/tt_b1.npz 1 > $O/tt.log 2>&1
fi
for f in c1 c2 c3 c4 c1_rccl1rank; do python - $O/bench_$f.json $f <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); s = d["stage_ms_per_step"]; r = d["roofline"]
    print(sys.argv[2], f"ms/step {d['ms_per_step']:.3f} xRT {d['x_realtime_16khz']:.0f} stages {s} trunk {r['achieved']:.1f} TF frac {r['frac']:.3f}", "sustained", (d.get("sustained") or {}).get("ms_per_step"), "parity", json.dumps(d.get("parity"))[:300], "cpu", json.dumps(d.get("cpu_baseline"))[:200])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
done

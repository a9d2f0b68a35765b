#!/bin/bash
# One parametrised GPU-box session script (replaces the per-session one-offs of rounds 3-5).
#   gpurun --timeout N -- 'bash tools/session.sh <what> [tag]'       results under gpurun_out/<tag>/
#   what: kloop     conv_h2p K-loop micro-benchmark (check + wall time per tile code + cycles per step from the tile-trace lab build)
#         ab        A/B of the step inside the engine: sts_debug_set knobs at BASELINE configs 1 / 2 / 4   (AB_SETS="h2p=0 h2p=2 ..." overrides)
#         tests     the GPU suite with the 45 slowest tests listed
#         final     round-end evidence at the current HEAD: rocprofv3 kernel trace + PMC passes of configs 1 and 4, the full bench lines of
#                   configs 1-4 (cpu_baseline + parity each), the one-rank RCCL line, power / clock trace, tile trace
cd "$(dirname "$0")/.."
WHAT=${1:-final}; TAG=${2:-r06}
O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --no-f32-leg --configs-block off --pipeline-engines 0 --min-seconds 0"
line() {   # label, bench args... -> one line of the step's figures
  local label=$1; shift
  $B "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; st = d['stage_ms_per_step']
print('%-34s ms/step %7.3f  decoder %7.3f  flow %6.3f  te %6.3f dur %6.3f | mfma region: %5.1f launches %7.1f us avg -> %7.3f ms  frac %.3f' % ('$label', d['ms_per_step'], st['decoder'], st['flow'], st['text_encoder'], st['duration'], r['launches_per_step'], r['avg_launch_us'], r['launches_per_step'] * r['avg_launch_us'] / 1e3, r['frac']))"
}
case $WHAT in
kloop)
  timeout 600 python tools/ubench/kloop_h2p.py check > $O/kloop_check.log 2>&1; echo "check rc=$?"; tail -2 $O/kloop_check.log
  timeout 900 python tools/ubench/kloop_h2p.py time > $O/kloop_time.log 2>&1; cat $O/kloop_time.log
  [ -f summertts_amd/lib/var/libvar6tt.so ] && SUMMERTTS_HIP_LIB=summertts_amd/lib/var/libvar6tt.so timeout 900 python tools/ubench/kloop_h2p.py trace > $O/kloop_trace.log 2>&1
  ;;
ab)
  for cfg in ${AB_CONFIGS:-1 2 4}; do
    line "c$cfg default" --config $cfg
    for s in ${AB_SETS:-h2p=0 h2p=2}; do line "c$cfg $s" --config $cfg --debug-set $s; done
  done 2>&1 | tee $O/ab.log
  ;;
tests)
  timeout 1500 python -m pytest tests -q -m gpu --durations=45 > $O/pytest_full.log 2>&1; echo "rc=$?"; tail -60 $O/pytest_full.log
  ;;
final)
  bash tools/profile_session.sh $O/c1 > $O/c1.log 2>&1
  bash tools/profile_session.sh $O/c4 --config 4 --steps 3 > $O/c4.log 2>&1
  timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_c1.json 2> $O/bench_c1.err
  for c in 2 3 4 5; do timeout 400 python bench.py --config $c --steps 5 --warmup 2 --cpu-reps 1 --cpu-threads 16 --pipeline-engines 0 > $O/bench_c$c.json 2> $O/bench_c$c.err; done
  STS_BENCH_FORCE_DIST=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --min-seconds 0 > $O/bench_c1_rccl1rank.json 2> $O/bench_c1_rccl1rank.err
  timeout 200 python tools/power_trace.py 5 > $O/power_trace.log 2>&1
  [ -f summertts_amd/lib/var/libvar6tt.so ] && SUMMERTTS_HIP_LIB=summertts_amd/lib/var/libvar6tt.so timeout 200 python tools/tile_trace_dump.py $O/tt_b1.npz 1 > $O/tt.log 2>&1
  for f in c1 c2 c3 c4 c5 c1_rccl1rank; do python - $O/bench_$f.json $f <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); s = d["stage_ms_per_step"]; r = d["roofline"]
    print(sys.argv[2], f"ms/step {d['ms_per_step']:.3f} xRT {d['x_realtime_16khz']:.0f} stages {s} trunk {r['achieved']:.1f} TF frac {r['frac']:.3f}", "sustained", (d.get("sustained") or {}).get("ms_per_step"), "parity", json.dumps(d.get("parity"))[:300], "cpu", json.dumps(d.get("cpu_baseline"))[:200])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
  done
  ;;
*) echo "unknown session '$WHAT'"; exit 2;;
esac

#!/bin/bash
# round 6 session 2: the pre-split path inside the engine -- parity subset, then A/B of the step (h2p off / on, tile codes) at configs 1, 2, 4
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "golden and f16x2" > gpurun_out/r06_s2_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r06_s2_pytest.log
B="python bench.py --no-cpu-baseline --no-f32-leg --configs-block off --pipeline-engines 0 --min-seconds 0"
run() {  # label, args...
  local label=$1; shift
  $B "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']; st = d['stage_ms_per_step']
print('%-34s ms/step %7.3f  decoder %7.3f  flow %6.3f  te %6.3f dur %6.3f | mfma region: %5.1f launches %7.1f us avg -> %7.3f ms  frac %.3f' % ('$label', d['ms_per_step'], st['decoder'], st['flow'], st['text_encoder'], st['duration'], r['launches_per_step'], r['avg_launch_us'], r['launches_per_step'] * r['avg_launch_us'] / 1e3, r['frac']))
"
}
for cfg in 1 2 4; do
  run "c$cfg staged (h2p=0)" --config $cfg --debug-set h2p=0
  run "c$cfg h2p auto" --config $cfg
  for t in 0 1 3 4 8 9; do
    run "c$cfg h2p tile128=$t tile256=0" --config $cfg --debug-set h2p_tile=$t
  done
  run "c$cfg h2p tile128=3 tile256=9" --config $cfg --debug-set h2p_tile=$((3 + 9*256))
  run "c$cfg h2p tile128=3 tile256=3" --config $cfg --debug-set h2p_tile=$((3 + 3*256))
  run "c$cfg h2p tile128=3 tile256=1" --config $cfg --debug-set h2p_tile=$((3 + 1*256))
done 2>&1 | tee gpurun_out/r06_s2_ab.log

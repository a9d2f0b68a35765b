# same-box A/B of library variants (tools/var_build.sh): bench lines of configs 1-4 per variant, twice
cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-f32-leg --configs-block off --pipeline-engines 0 --min-seconds 0"
line() { local label=$1; shift; "$@" 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; st = d['stage_ms_per_step']
print('%-28s ms/step %7.3f  decoder %7.3f | mfma region: %7.3f ms  frac %.3f' % ('$label', d['ms_per_step'], st['decoder'], r['launches_per_step'] * r['avg_launch_us'] / 1e3, r['frac']))"; }
for rep in 1 2; do for cfg in ${AB_CONFIGS:-1 2 3 4}; do
  line "c$cfg default" $B --config $cfg
  for v in ${AB_LIBS}; do SUMMERTTS_HIP_LIB=summertts_amd/lib/var/libvar$v.so line "c$cfg $v" $B --config $cfg; done
done; done

# session 2: (1) the test whose worker crashed under xdist, alone and serial, three times; (2) the whole GPU suite on 8 workers, no -x;
# (3) same-box A/B of what the HIP events and launch-ahead cost the timed step
O=gpurun_out/r05s2
mkdir -p $O
for i in 1 2 3; do timeout 300 python -X faulthandler -m pytest tests -m gpu -q -k "odd_T9" 2>&1 | tail -3; done > $O/odd.log 2>&1
cat $O/odd.log
( time timeout 1500 python -m pytest tests -m gpu -q -n 8 --timeout 900 ) > $O/pytest.log 2>&1
tail -12 $O/pytest.log
Q="--no-cpu-baseline --no-f32-leg --pipeline-engines 0 --steps 40 --warmup 5 --configs-block off --min-seconds 0"
for rep in 1 2; do
  for v in "2 1" "0 1" "1 1" "2 0" "0 0"; do
    set -- $v
    timeout 200 python bench.py $Q --timed-profiling $1 --debug-set launch_ahead=$2 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('timed_profiling=$1 launch_ahead=$2', 'ms/step', round(d['ms_per_step'],4), 'p50', round(d['p50_latency_ms'],4), 'api', round(d['api_call_leg']['ms_per_call'],4), 'stage_leg', round(d['stage_breakdown_leg']['ms_per_step'],4), d['stage_ms_per_step'], 'trunk_us', round(d['roofline']['avg_launch_us']*d['roofline']['launches_per_step'],1))" >> $O/ab.txt
  done
done
cat $O/ab.txt

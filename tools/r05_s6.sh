# session 6: the occupancy switches adopted (32-row tiles and the grouped 128 x 128 tile at three waves per SIMD): trunk parity tests, bench lines
O=gpurun_out/r05s6
mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "full_size_configs or near_full_scale or realistic_weight or unusual_resblock or streaming or conv_kernels_against or f16x2_conv or bf3_conv or amplitude_edge" ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
Q="--no-cpu-baseline --no-f32-leg --pipeline-engines 0 --steps 40 --warmup 5 --configs-block off --min-seconds 0"
for rep in 1 2 3; do
  timeout 200 python bench.py $Q 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c1', 'ms/step', round(d['ms_per_step'],4), 'dec', round(d['stage_ms_per_step']['decoder'],4), 'trunk_us', round(d['roofline']['avg_launch_us']*d['roofline']['launches_per_step'],1), 'frac', round(d['roofline']['frac'],4))" >> $O/ab.txt
done
for c in 2 3 4; do
  timeout 300 python bench.py $Q --config $c --steps 6 --warmup 2 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c$c', 'ms/step', round(d['ms_per_step'],4), 'dec', round(d['stage_ms_per_step']['decoder'],4), 'trunk_us', round(d['roofline']['avg_launch_us']*d['roofline']['launches_per_step'],1), 'frac', round(d['roofline']['frac'],4), 'sync', round(d['host_sync_wait_ms_per_step'],4))" >> $O/ab.txt
done
cat $O/ab.txt

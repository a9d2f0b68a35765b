# session 3: the wide-chunk upsampler tiles: parity (new test + the flow-width test), then same-box A/B of up_wide = 0 / 1 / 2 at one utterance and batch 32
O=gpurun_out/r05s3
mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q --timeout 600 -k "wide_chunk or flow_layer_kernel_at_every" ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
Q="--no-cpu-baseline --no-f32-leg --pipeline-engines 0 --steps 40 --warmup 5 --configs-block off --min-seconds 0"
for rep in 1 2; do
  for v in 0 1 2; do
    timeout 200 python bench.py $Q --debug-set up_wide=$v 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c1 up_wide=$v', 'ms/step', round(d['ms_per_step'],4), d['stage_ms_per_step']['decoder'], 'trunk_us', round(d['roofline']['avg_launch_us']*d['roofline']['launches_per_step'],1))" >> $O/ab.txt
  done
done
for v in 0 1 2; do
  timeout 300 python bench.py $Q --config 2 --steps 6 --warmup 2 --debug-set up_wide=$v 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c2 up_wide=$v', 'ms/step', round(d['ms_per_step'],4), d['stage_ms_per_step']['decoder'], 'trunk_us', round(d['roofline']['avg_launch_us']*d['roofline']['launches_per_step'],1))" >> $O/ab.txt
  timeout 300 python bench.py $Q --config 4 --steps 6 --warmup 2 --debug-set up_wide=$v 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c4 up_wide=$v', 'ms/step', round(d['ms_per_step'],4), d['stage_ms_per_step']['decoder'], 'trunk_us', round(d['roofline']['avg_launch_us']*d['roofline']['launches_per_step'],1))" >> $O/ab.txt
done
cat $O/ab.txt
# per-kernel times of the upsamplers under each setting (kernel trace, 6 steps)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for v in 0 1 2; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$v -- python bench.py $Q --steps 6 --warmup 2 --debug-set up_wide=$v > $O/kt$v.log 2>&1
  f=$(find $O/kt$v -name "*kernel_stats.csv" | head -1)
  echo "== up_wide=$v"; grep "conv_bf3_kernel" $f | cut -d, -f1-4 | cut -c1-150
done > $O/kstats.txt 2>&1
cat $O/kstats.txt

# session 7: same-box A/B of (a) the grouped 128 x 128 tile at three (default now) vs two (g2) waves per SIMD, (b) conv_post staging 16 (default now) vs 8 / 32 rows per round
O=gpurun_out/r05s7
mkdir -p $O
Q="--no-cpu-baseline --no-f32-leg --pipeline-engines 0 --steps 40 --warmup 5 --configs-block off --min-seconds 0"
for rep in 1 2 3; do
  for lib in default var6g2 var6cb8 var6cb32; do
    if [ $lib = default ]; then L=""; else L="SUMMERTTS_HIP_LIB=summertts_amd/lib/var/lib$lib.so"; fi
    env $L timeout 200 python bench.py $Q 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c1 $lib', 'ms/step', round(d['ms_per_step'],4), 'dec', round(d['stage_ms_per_step']['decoder'],4), 'trunk_us', round(d['roofline']['avg_launch_us']*d['roofline']['launches_per_step'],1))" >> $O/ab.txt
  done
done
for lib in default var6g2; do
  if [ $lib = default ]; then L=""; else L="SUMMERTTS_HIP_LIB=summertts_amd/lib/var/lib$lib.so"; fi
  for c in 4 2; do
  env $L timeout 300 python bench.py $Q --config $c --steps 6 --warmup 2 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c$c $lib', 'ms/step', round(d['ms_per_step'],4), 'dec', round(d['stage_ms_per_step']['decoder'],4), 'trunk_us', round(d['roofline']['avg_launch_us']*d['roofline']['launches_per_step'],1))" >> $O/ab.txt
  done
done
cat $O/ab.txt
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for lib in default var6g2 var6cb8 var6cb32; do
  if [ $lib = default ]; then L=""; else L="SUMMERTTS_HIP_LIB=summertts_amd/lib/var/lib$lib.so"; fi
  env $L rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$lib -- python bench.py $Q --steps 6 --warmup 2 > $O/kt_$lib.log 2>&1
  f=$(find $O/kt_$lib -name "*kernel_stats.csv" | head -1)
  echo "== $lib"; grep "conv_bf3_group\|cout1\|resblock" $f | python -c "
import sys,csv
for r in csv.reader(sys.stdin): print('  ', r[0][10:70], r[1], round(float(r[3])/1000,1))"
done > $O/kstats.txt 2>&1
cat $O/kstats.txt

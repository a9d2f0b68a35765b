# session 5: occupancy variants of the staged two-term kernels: mw1w3 = the 32-rows-per-wave tiles at three waves per SIMD; w3 = every staged tile at
# three (168 registers, a 20-byte spill in the 128-channel grouped kernel); configs 1 / 2 / 4, parity of the variants, per-kernel times
O=gpurun_out/r05s5
mkdir -p $O
Q="--no-cpu-baseline --no-f32-leg --pipeline-engines 0 --steps 40 --warmup 5 --configs-block off --min-seconds 0"
for rep in 1 2; do
  for lib in default var6mw1w3 var6w3; do
    if [ $lib = default ]; then L=""; else L="SUMMERTTS_HIP_LIB=summertts_amd/lib/var/lib$lib.so"; fi
    env $L timeout 200 python bench.py $Q 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c1 $lib', 'ms/step', round(d['ms_per_step'],4), 'dec', round(d['stage_ms_per_step']['decoder'],4), 'trunk_us', round(d['roofline']['avg_launch_us']*d['roofline']['launches_per_step'],1))" >> $O/ab.txt
  done
done
for lib in default var6mw1w3 var6w3; do
  if [ $lib = default ]; then L=""; else L="SUMMERTTS_HIP_LIB=summertts_amd/lib/var/lib$lib.so"; fi
  env $L timeout 300 python bench.py $Q --config 4 --steps 6 --warmup 2 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c4 $lib', 'ms/step', round(d['ms_per_step'],4), 'dec', round(d['stage_ms_per_step']['decoder'],4), 'trunk_us', round(d['roofline']['avg_launch_us']*d['roofline']['launches_per_step'],1))" >> $O/ab.txt
done
for lib in default var6mw1w3 var6w3; do
  if [ $lib = default ]; then L=""; else L="SUMMERTTS_HIP_LIB=summertts_amd/lib/var/lib$lib.so"; fi
  env $L timeout 300 python bench.py $Q --config 2 --steps 6 --warmup 2 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c2 $lib', 'ms/step', round(d['ms_per_step'],4), 'dec', round(d['stage_ms_per_step']['decoder'],4), 'trunk_us', round(d['roofline']['avg_launch_us']*d['roofline']['launches_per_step'],1))" >> $O/ab.txt
done
cat $O/ab.txt
# parity of the variants: the T = 128 golden + the loud fixture under f16x2 through the variant library
for lib in var6mw1w3 var6w3; do
  SUMMERTTS_HIP_LIB=summertts_amd/lib/var/lib$lib.so timeout 600 python -m pytest tests -m gpu -q --timeout 600 -k "near_full_scale or (full_size_configs and f16x2)" 2>&1 | tail -3
done > $O/pytest_var.log 2>&1
cat $O/pytest_var.log
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for lib in default var6mw1w3 var6w3; do
  if [ $lib = default ]; then L=""; else L="SUMMERTTS_HIP_LIB=summertts_amd/lib/var/lib$lib.so"; fi
  env $L rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$lib -- python bench.py $Q --steps 6 --warmup 2 > $O/kt_$lib.log 2>&1
  f=$(find $O/kt_$lib -name "*kernel_stats.csv" | head -1)
  echo "== $lib"; grep "conv_bf3" $f | python -c "
import sys,csv
for r in csv.reader(sys.stdin): print('  ', r[0][10:70], r[1], round(float(r[3])/1000,1))"
done > $O/kstats.txt 2>&1
cat $O/kstats.txt

# session 10: the reflect-padded subband conv of the iSTFT decoder families on the two-term kernels (was: exact-fp32 MFMA, 1.3 ms of the 33 ms config-4 step)
O=gpurun_out/r05s10
mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "mbb or ms_sdp or ms_fix or istft or amplitude_edge or full_size_configs or realistic_weight or streaming" ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
Q="--no-cpu-baseline --no-f32-leg --pipeline-engines 0 --configs-block off --min-seconds 0"
for rep in 1 2; do for lib in default var6norefl; do
  if [ $lib = default ]; then L=""; else L="SUMMERTTS_HIP_LIB=summertts_amd/lib/var/lib$lib.so"; fi
  env $L timeout 300 python bench.py $Q --config 4 --steps 8 --warmup 2 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('c4 $lib', 'ms/step', round(d['ms_per_step'],4), d['stage_ms_per_step'])" >> $O/ab.txt
  env $L timeout 300 python bench.py $Q --workload ms_fix --batch 64 --ragged --steps 8 --warmup 2 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('ms_fix b64 $lib', 'ms/step', round(d['ms_per_step'],4), d['stage_ms_per_step'])" >> $O/ab.txt
done; done
cat $O/ab.txt

Q="--no-cpu-baseline --no-f32-leg --pipeline-engines 0 --steps 40 --warmup 5 --configs-block off --min-seconds 0"
run() { env "$@" timeout 200 python bench.py $Q 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['stage_ms_per_step']; print('$*'.ljust(36), 'ms/step', round(d['ms_per_step'],4), 'p50', round(d['p50_latency_ms'],4), 'te', round(s['text_encoder'],3), 'dp', round(s['duration'],3), 'flow', round(s['flow'],3), 'dec', round(s['decoder'],3), 'api', round(d['api_call_leg']['ms_per_call'],4))"; }
for rep in 1 2; do
run X=1
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run HSA_ENABLE_SDMA=0
run AMD_DIRECT_DISPATCH=0
run GPU_MAX_HW_QUEUES=2
run HSA_ENABLE_INTERRUPT=0
done

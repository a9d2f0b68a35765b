mkdir -p gpurun_out/s20
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "prepared_batch or (hip_matches_reference_golden and f16x2) or (full_size_configs and f16x2 and (hifigan_sdp_T128 or mbb_fix_T96)) or request_pool or streaming_equals" 2>&1 | tail -5 > gpurun_out/s20/pytest.txt
cat gpurun_out/s20/pytest.txt
B="python bench.py --no-cpu-baseline --no-f32-leg --pipeline-engines 0 --min-seconds 0"
for i in 1 2; do
$B --steps 40 --warmup 5 > gpurun_out/s20/on_$i.json 2> gpurun_out/s20/e1
$B --steps 40 --warmup 5 --debug-set pcm_direct=0 > gpurun_out/s20/off_$i.json 2> gpurun_out/s20/e2
done
tail -3 gpurun_out/s20/e1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/s20/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); st=d.get('stage_ms_per_step') or {}
        print(f, round(d['ms_per_step'],4), {k:round(v,3) for k,v in st.items()}, d.get('host_us_per_step',{}).get('step_wall_minus_device_stages'))
    except Exception as e: print(f, 'ERR', e)
PY

mkdir -p gpurun_out/s18
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "prepared_batch or launch_ahead or (full_size_configs and f16x2 and T128)" 2>&1 | tail -5 > gpurun_out/s18/pytest.txt
cat gpurun_out/s18/pytest.txt
B="python bench.py --no-cpu-baseline --no-f32-leg --pipeline-engines 0 --min-seconds 2"
$B --steps 40 --warmup 5 > gpurun_out/s18/a.json 2> gpurun_out/s18/e1
$B --steps 40 --warmup 5 > gpurun_out/s18/b.json 2> gpurun_out/s18/e2
tail -3 gpurun_out/s18/e1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/s18/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); st=d.get('stage_ms_per_step') or {}
        print(f, round(d['ms_per_step'],4), round(d['sustained']['ms_per_step'],4), {k:round(v,3) for k,v in st.items()}, d.get('host_us_per_step'))
    except Exception as e: print(f, 'ERR', e)
PY

mkdir -p gpurun_out/s17
timeout 900 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "full_size_configs or near_full_scale or tiny or duration or text_encoder or attention or ragged or batch" 2>&1 | tail -5 > gpurun_out/s17/pytest.txt
cat gpurun_out/s17/pytest.txt
B="python bench.py --no-cpu-baseline --no-f32-leg --pipeline-engines 0 --min-seconds 0"
for i in 1 2; do
$B --steps 40 --warmup 5 > gpurun_out/s17/on_$i.json 2> gpurun_out/s17/e1
$B --steps 40 --warmup 5 --debug-set attn_reg=0 > gpurun_out/s17/off_$i.json 2> gpurun_out/s17/e2
done
tail -3 gpurun_out/s17/e1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/s17/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); st=d.get('stage_ms_per_step') or {}
        print(f, round(d['ms_per_step'],4), {k:round(v,3) for k,v in st.items()})
    except Exception as e: print(f, 'ERR', e)
PY

mkdir -p gpurun_out/s12
python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "launch_ahead" 2>&1 | tail -15 > gpurun_out/s12/tests.log
cat gpurun_out/s12/tests.log
B="python bench.py --no-cpu-baseline --no-f32-leg --pipeline-engines 0 --min-seconds 0"
for i in 1 2; do
$B --steps 40 --warmup 5 > gpurun_out/s12/la1_$i.json 2> gpurun_out/s12/e1
$B --steps 40 --warmup 5 --debug-set launch_ahead=0 > gpurun_out/s12/la0_$i.json 2> gpurun_out/s12/e2
done

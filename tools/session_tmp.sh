mkdir -p gpurun_out/s10
B="python bench.py --no-cpu-baseline --no-f32-leg --pipeline-engines 0"
$B --steps 30 --warmup 5 > gpurun_out/s10/ff1.json 2> gpurun_out/s10/e1
$B --steps 30 --warmup 5 --debug-set flow_fused=0 > gpurun_out/s10/ff0.json 2> gpurun_out/s10/e2
$B --steps 10 --warmup 3 --config 2 > gpurun_out/s10/c2_ff1.json 2> gpurun_out/s10/e3
$B --steps 10 --warmup 3 --config 2 --debug-set flow_fused=0 > gpurun_out/s10/c2_ff0.json 2> gpurun_out/s10/e4
$B --steps 10 --warmup 3 --config 4 > gpurun_out/s10/c4_ff1.json 2> gpurun_out/s10/e5
$B --steps 10 --warmup 3 --config 4 --debug-set flow_fused=0 > gpurun_out/s10/c4_ff0.json 2> gpurun_out/s10/e6
python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "(full_size_configs and f16x2) or near_full_scale or batch_equals_single or forced_durations or long_utterance" 2>&1 | tail -8 > gpurun_out/s10/tests.log
cat gpurun_out/s10/tests.log

set -x
mkdir -p gpurun_out/s4
B="python bench.py --no-cpu-baseline --no-f32-leg --pipeline-engines 0 --steps 30 --warmup 5"
python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "near_full_scale or (full_size_configs and (T128 or batch8_hifigan)) or streaming_equals" 2>&1 | tail -4 > gpurun_out/s4/tests.log
$B > gpurun_out/s4/rb3_claim.json 2> gpurun_out/s4/e1
$B --debug-set tile_claim=0 > gpurun_out/s4/rb3_noclaim.json 2> gpurun_out/s4/e2
SUMMERTTS_HIP_LIB=summertts_amd/lib/var/libvar6rb2.so $B > gpurun_out/s4/rb2_claim.json 2> gpurun_out/s4/e3
SUMMERTTS_HIP_LIB=summertts_amd/lib/var/libvar6rb2.so $B --debug-set tile_claim=0 > gpurun_out/s4/rb2_noclaim.json 2> gpurun_out/s4/e4
$B --config 2 > gpurun_out/s4/c2_rb3_claim.json 2> gpurun_out/s4/e5
$B --config 2 --debug-set tile_claim=0 > gpurun_out/s4/c2_rb3_noclaim.json 2> gpurun_out/s4/e6
cat gpurun_out/s4/tests.log

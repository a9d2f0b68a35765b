mkdir -p gpurun_out/s13
B="python bench.py --no-cpu-baseline --no-f32-leg --pipeline-engines 0 --min-seconds 0"
for i in 1 2; do
$B --steps 40 --warmup 5 > gpurun_out/s13/rpf1_$i.json 2> gpurun_out/s13/e1
SUMMERTTS_HIP_LIB=summertts_amd/lib/var/libvar6rpf0.so $B --steps 40 --warmup 5 > gpurun_out/s13/rpf0_$i.json 2> gpurun_out/s13/e2
done
$B --steps 10 --warmup 3 --config 2 > gpurun_out/s13/c2_rpf1.json 2> gpurun_out/s13/e3
SUMMERTTS_HIP_LIB=summertts_amd/lib/var/libvar6rpf0.so $B --steps 10 --warmup 3 --config 2 > gpurun_out/s13/c2_rpf0.json 2> gpurun_out/s13/e4
python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "(full_size_configs and f16x2 and (T128 or batch8_hifigan)) or near_full_scale" 2>&1 | tail -3

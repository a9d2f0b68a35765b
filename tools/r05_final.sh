# Round-5 evidence at the freeze build (one gpurun call): rocprofv3 passes of config 1, a tile trace, the bench lines of configs 2-4 (each with parity +
# cpu_baseline), the one-rank RCCL line, the native multi-device line with three emulated ranks, the power / clock trace.   bash tools/r05_final.sh [tag]
TAG=${1:-r05z}
O=gpurun_out/$TAG
mkdir -p $O
bash tools/profile_session.sh $O/c1 --configs-block off --min-seconds 0 > $O/c1.log 2>&1
if [ -f summertts_amd/lib/var/libvar6tt.so ]; then
  SUMMERTTS_HIP_LIB=summertts_amd/lib/var/libvar6tt.so timeout 200 python tools/tile_trace_dump.py $O/tt_b1.npz 1 > $O/tt.log 2>&1
fi
for c in 2 3 4; do
  timeout 500 python bench.py --config $c --steps 5 --warmup 2 --cpu-reps 1 --cpu-threads 16 --pipeline-engines 0 > $O/bench_c$c.json 2> $O/bench_c$c.err
done
STS_BENCH_FORCE_DIST=1 timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --min-seconds 0 --configs-block off > $O/bench_c1_rccl1rank.json 2> $O/bench_c1_rccl1rank.err
STS_TEST_HOOKS=1 STS_BENCH_RCCL_LIB=$PWD/tests/fake_rccl/libfake_rccl.so timeout 300 python bench.py --gpus 3 --multi native --share-gpu --steps 5 --warmup 2 > $O/bench_multi_native_3emulated.json 2> $O/bench_multi_native_3emulated.err
timeout 200 python bench.py --gpus 1 --multi native --steps 10 --warmup 3 > $O/bench_multi_native_1rank.json 2> $O/bench_multi_native_1rank.err
timeout 200 python tools/power_trace.py 5 > $O/power_trace.log 2>&1
for f in c2 c3 c4 c1_rccl1rank; do python tools/bench_line.py $O/bench_$f.json; done
tail -3 $O/bench_multi_native_3emulated.json | cut -c1-1500
tail -3 $O/bench_multi_native_1rank.json | cut -c1-800
tail -12 $O/power_trace.log

# the default bench line (what the driver runs) with the PMC summary of this build attached, then the whole GPU suite serially (as the driver runs it)
O=gpurun_out/r05zb
mkdir -p $O
( time timeout 600 python bench.py ) > $O/bench_c1.json 2> $O/bench_c1.err
python tools/bench_line.py $O/bench_c1.json
( time timeout 1700 python -m pytest tests -m gpu -x -q ) > $O/pytest_serial.log 2>&1
tail -6 $O/pytest_serial.log

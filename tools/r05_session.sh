# Round-5 GPU session (one gpurun call): the GPU suite on several workers (the oracle legs are host-bound), then bench lines.
#   usage: bash tools/r05_session.sh <tag> [pytest -k expression | "all" | "none"] [bench: default|quick|none]
TAG=${1:-r05s}
SEL=${2:-all}
BENCH=${3:-default}
O=gpurun_out/$TAG
mkdir -p $O
if [ "$SEL" != "none" ]; then
  if [ "$SEL" = "all" ]; then
    ( time timeout 1500 python -m pytest tests -m gpu -q -n 8 -x --timeout 900 ) > $O/pytest.log 2>&1
  else
    ( time timeout 1500 python -m pytest tests -m gpu -q -n 8 --timeout 900 -k "$SEL" ) > $O/pytest.log 2>&1
  fi
  tail -15 $O/pytest.log
fi
if [ "$BENCH" = "default" ]; then
  ( time timeout 600 python bench.py ) > $O/bench_c1.json 2> $O/bench_c1.err
  tail -3 $O/bench_c1.err
elif [ "$BENCH" = "quick" ]; then
  ( time timeout 300 python bench.py --no-cpu-baseline --no-f32-leg --pipeline-engines 0 --steps 30 --warmup 5 --configs-block off ) > $O/bench_q.json 2> $O/bench_q.err
  tail -3 $O/bench_q.err
fi
for f in $O/bench_*.json; do python tools/bench_line.py $f 2>/dev/null || true; done

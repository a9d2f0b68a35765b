#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( time STS_TEST_HOOKS=1 timeout 900 python tests/fake_rccl/three_ranks.py ) > gpurun_out/r06_s4_three_ranks.log 2>&1
python - <<'P'
import json
t = open('gpurun_out/r06_s4_three_ranks.log').read()
for ln in t.splitlines():
    if ln.startswith('{'):
        for c in json.loads(ln)['checks']: print(c['ok'], c['name'][:70], '|', c['detail'][:80])
print(t[-300:])
P
AB_CONFIGS="1" AB_SETS="chain_streams=1 chain_streams=2 chain_streams=3 chain_streams=12 chain_streams=15 h2p=2" bash tools/session.sh ab r06s4
AB_CONFIGS="2 4" AB_SETS="chain_streams=15 h2p=0" bash tools/session.sh ab r06s4b
python bench.py --no-cpu-baseline --no-f32-leg --configs-block off --min-seconds 0 --pipeline-engines 4 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], 'p50', d['p50_latency_ms']); print(json.dumps(d['request_pool'], indent=0)[:1500])"

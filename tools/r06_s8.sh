#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
AB_CONFIGS="1" AB_SETS="h2p=3 h2p=4" bash tools/session.sh ab r06s8
AB_CONFIGS="2 4" AB_SETS="h2p=5" bash tools/session.sh ab r06s8b
for t in 0 1 3 4 8; do AB_CONFIGS="4" AB_SETS="" ; python bench.py --no-cpu-baseline --no-f32-leg --configs-block off --pipeline-engines 0 --min-seconds 0 --config 4 --debug-set h2p=5 --debug-set h2p_tile=$((t + 3*256)) 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; print('c4 h2p=5 tile128=$t', round(d['ms_per_step'],3), 'frac', round(r['frac'],3))"; done

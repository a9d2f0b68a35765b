#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
( time STS_TEST_HOOKS=1 timeout 900 python tests/fake_rccl/three_ranks.py > /dev/null ) 2>&1 | tail -4
AB_CONFIGS="4 5" AB_SETS="tail_fused=0" bash tools/session.sh ab r06s5
bash tools/session.sh tests r06s5 | tail -55

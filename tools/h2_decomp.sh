#!/bin/bash
# K loop of the two-term fp16 128 x 128 tile with parts of a step compiled out (STS_EXP: 64 no MFMAs, 16 no LDS reads, 2 no weight
# loads, 82 none of the three) -- lab builds: VAR_TAG=e<N> VAR_EXTRA="-DSTS_TILE_TRACE -DSTS_EXP=<N>" tools/var_build.sh 6
cd "$(dirname "$0")/.."
for n in 0 64 16 2 82; do
  echo "== STS_EXP $n"
  TT_H2=1 SUMMERTTS_HIP_LIB=summertts_amd/lib/var/libvar6e$n.so timeout 200 python tools/tile_trace_conv.py 2>&1 | grep "f16x2" | grep -E " (1|512|4096) workgroups" | grep "k=11"
done

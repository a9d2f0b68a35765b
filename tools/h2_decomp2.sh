#!/bin/bash
cd "$(dirname "$0")/.."
for n in ${H2_DECOMP_SET:-18 31}; do
  echo "== STS_EXP $n"
  TT_H2=1 SUMMERTTS_HIP_LIB=summertts_amd/lib/var/libvar6e$n.so timeout 200 python tools/tile_trace_conv.py 2>&1 | grep -E " (1|512|4096) workgroups" | grep "k=11"
done

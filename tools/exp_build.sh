#!/bin/bash
# Builds timing-experiment variants of the library (conv.hip compiled with -DSTS_EXP=<mask>) into
# summertts_amd/lib/exp/libexp<mask>.so.  Results of these variants are WRONG by design (parts of the kernel
# are switched off); they only answer "how much time does this part cost".  Use with
#   SUMMERTTS_HIP_LIB=summertts_amd/lib/exp/libexp<mask>.so python tools/conv_bench.py ...
set -e
cd "$(dirname "$0")/.."
mkdir -p summertts_amd/lib/exp
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form"
# EXP_FILE selects the translation unit the mask applies to (conv | conv_bf3 | col_layer); the library is lib<file><mask>.so for col_layer
X=${EXP_FILE:-conv}
for m in "$@"; do
  /opt/rocm/bin/hipcc $F -DSTS_EXP=$m -c summertts_amd/csrc/$X.hip -o summertts_amd/lib/exp/$X$m.o &
done
wait
O=summertts_amd/lib/obj
OTHER=""; for f in conv conv_bf3 col_layer; do [ "$f" = "$X" ] || OTHER="$OTHER $O/$f.o"; done
P=$([ "$X" = conv ] && echo exp || echo $X)
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o summertts_amd/lib/exp/lib$P$m.so summertts_amd/lib/exp/$X$m.o $OTHER $O/misc_kernels.o $O/model.o $O/engine.o $O/capi.o $O/pool.o $O/multi.o $O/synthesizer_trn.o -pthread
  rm summertts_amd/lib/exp/$X$m.o
done
ls -la summertts_amd/lib/exp

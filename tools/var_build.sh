#!/bin/bash
# Builds A/B variants of the library: conv_bf3.hip compiled with -DSTS_VAR=<mask> into summertts_amd/lib/var/libvar<mask>$VAR_TAG.so
# (every variant computes the same results; the masks are documented at STS_VAR in conv_bf3.hip).  Use with
#   SUMMERTTS_HIP_LIB=summertts_amd/lib/var/libvar<mask>$VAR_TAG.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
mkdir -p summertts_amd/lib/var
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form"
for m in "$@"; do
  /opt/rocm/bin/hipcc $F $VAR_EXTRA -DSTS_VAR=$m -c summertts_amd/csrc/conv_bf3.hip -o summertts_amd/lib/var/conv_bf3_$m$VAR_TAG.o &
done
wait
O=summertts_amd/lib/obj
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o summertts_amd/lib/var/libvar$m$VAR_TAG.so summertts_amd/lib/var/conv_bf3_$m$VAR_TAG.o $O/persist.o $O/conv.o $O/col_layer.o $O/misc_kernels.o $O/model.o $O/engine.o $O/capi.o $O/pool.o $O/multi.o $O/synthesizer_trn.o -pthread -ldl
  rm summertts_amd/lib/var/conv_bf3_$m$VAR_TAG.o
done
ls -la summertts_amd/lib/var

#!/bin/bash
# Builds A/B variants of the library: ONE translation unit (VAR_SRC, default conv_bf3.hip) compiled with -DSTS_VAR=<mask> $VAR_EXTRA, linked with
# the default build's other objects into summertts_amd/lib/var/libvar<mask>$VAR_TAG.so.  VAR_SRC may list several sources ("a.hip b.hip").
# (every STS_VAR mask computes the same results; the masks are documented at STS_VAR in conv_bf3_dev.hpp.)  Use with
#   SUMMERTTS_HIP_LIB=summertts_amd/lib/var/libvar<mask>$VAR_TAG.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
mkdir -p summertts_amd/lib/var
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value -mllvm -amdgpu-mfma-vgpr-form"
SRCS=${VAR_SRC:-"conv_bf3.hip conv_bf3_group.hip resblock_bf3.hip"}
O=summertts_amd/lib/obj
for m in "$@"; do
  for s in $SRCS; do
    FF="$F"; [ "$s" = conv_h2p.hip ] && FF="${F% -mllvm -amdgpu-mfma-vgpr-form}"     # (its 128 x 128 tiles need the AGPR half: Makefile FLAGS_conv_h2p)
    /opt/rocm/bin/hipcc $FF $VAR_EXTRA -DSTS_VAR=$m -c summertts_amd/csrc/$s -o summertts_amd/lib/var/${s%.hip}_$m$VAR_TAG.o &
  done
done
wait
for m in "$@"; do
  objs=""; skip=""
  for s in $SRCS; do objs="$objs summertts_amd/lib/var/${s%.hip}_$m$VAR_TAG.o"; skip="$skip ${s%.hip}.o"; done
  rest=""
  for o in $O/*.o; do b=$(basename $o); case " $skip " in *" $b "*) ;; *) rest="$rest $o";; esac; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o summertts_amd/lib/var/libvar$m$VAR_TAG.so $objs $rest -pthread -ldl
  rm -f $objs
done
ls -la summertts_amd/lib/var

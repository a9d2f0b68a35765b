#!/bin/bash
# round 6 session 1: pre-split conv kernels -- correctness, wall time per tile code, K-loop cycles per step
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/ubench/kloop_h2p.py check > gpurun_out/r06_s1_check.log 2>&1
echo "check rc=$?"
tail -3 gpurun_out/r06_s1_check.log
timeout 900 python tools/ubench/kloop_h2p.py time > gpurun_out/r06_s1_time.log 2>&1
echo "time rc=$?"
SUMMERTTS_HIP_LIB=summertts_amd/lib/var/libvar6tt.so timeout 900 python tools/ubench/kloop_h2p.py trace > gpurun_out/r06_s1_trace.log 2>&1
echo "trace rc=$?"
cat gpurun_out/r06_s1_time.log

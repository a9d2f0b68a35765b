#!/bin/bash
# One profiling session on the GPU box: kernel trace (+ --stats) and the three PMC passes (FETCH_SIZE, WRITE_SIZE, SQ busy) of
# bench.py for one workload.  Counters are collected in their own runs (no trace domains combined with --pmc).
#   tools/profile_session.sh <out-dir under gpurun_out> <bench args...>
# Summarise afterwards with tools/summarize_profile.py (locally: the CSVs merge back with gpurun_out/).
out=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
B="python bench.py --no-cpu-baseline --no-f32-leg --pipeline-engines 0 --configs-block off --min-seconds 0 $*"     # (only the K timed steps + the stage leg of ONE workload in the trace)
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- $B --steps 6 --warmup 3 > "$out/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- $B --steps 3 --warmup 1 > "$out/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -- $B --steps 3 --warmup 1 > "$out/write.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$out/sq" -- $B --steps 3 --warmup 1 > "$out/sq.log" 2>&1
du -sh "$out"

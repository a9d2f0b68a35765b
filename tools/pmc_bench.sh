#!/bin/bash
# SQ counter passes over bench.py, summarised per (kernel, grid) (GPU box).
#   tools/pmc_bench.sh <out-dir under gpurun_out> <match> <bench args...>      (environment knobs pass through)
out=$1; match=$2; shift 2
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
B="python bench.py --no-cpu-baseline --no-f32-leg --pipeline-engines 0 --steps 3 --warmup 1 $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d "$out/p1" -- $B > "$out/p1.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d "$out/p2" -- $B > "$out/p2.log" 2>&1
python tools/pmc_by_grid.py "$out/p1" "$out/p2" --match "$match" > "$out/summary.txt" 2>&1
cat "$out/summary.txt"

#!/bin/bash
# SQ counter passes over tools/conv_bench.py for one shape / one kernel variant (GPU box).
#   tools/pmc_conv.sh <out-dir under gpurun_out> <frames> <modes> <shape...>
out=$1; frames=$2; modes=$3; shift 3
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
mkdir -p "$out"
export CONV_BENCH_FRAMES=$frames CONV_BENCH_MODES=$modes
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d "$out/p1" -- python tools/conv_bench.py "$@" > "$out/p1.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv -d "$out/p2" -- python tools/conv_bench.py "$@" > "$out/p2.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVES TCP_PENDING_STALL_CYCLES_sum --output-format csv -d "$out/p3" -- python tools/conv_bench.py "$@" > "$out/p3.log" 2>&1
python tools/pmc_by_grid.py "$out/p1" "$out/p2" "$out/p3" --match bf3 > "$out/summary.txt" 2>&1
tail -n 3 "$out/p1.log"
cat "$out/summary.txt"

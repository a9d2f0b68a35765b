cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/rb_pmc; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-f32-leg --pipeline-engines 0 --configs-block off --min-seconds 0 --config 2 --steps 2 --warmup 1"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU SQ_INSTS_SALU" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCC_HIT_sum" "TCC_MISS_sum TCC_REQ_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout 280 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- $B > $O/p$i.log 2>&1
  echo "pass $i ($set) rc=$?"
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/rb_pmc/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        key = "fused64" if "resblock_bf3_kernel<1, 2, 2, 2, 1>" in n else ("fused32" if "resblock_bf3_kernel<1, 1, 2, 4, 1>" in n else ("h2p" if "conv_h2p_group" in n else None))
        if key: acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]; print("   %-34s n=%3d mean %.4g" % (c, len(v), sum(v) / len(v)))
PY

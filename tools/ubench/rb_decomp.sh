# Decomposition of the fused narrow-stage ResBlock kernel (resblock_bf3.hip, lab switches STS_RB_EXP): rocprofv3 average duration of its launches
# at BASELINE configs[2] (and [1]) with pieces of the K loops compiled out.  TIMING only: the variants compute wrong results.
#   for e in 1 2 3 4 7; do VAR_SRC=resblock_bf3.hip VAR_EXTRA=-DSTS_RB_EXP=$e VAR_TAG=e$e tools/var_build.sh 0; done;  gpurun -- bash tools/ubench/rb_decomp.sh
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/rb_decomp; mkdir -p $O
for cfg in ${CFGS:-2 1}; do
for v in default e1 e2 e3 e4 e7; do
  lib=""; [ $v != default ] && lib=summertts_amd/lib/var/libvar0$v.so
  rm -rf /tmp/rbp; SUMMERTTS_HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rbp -- python bench.py --no-cpu-baseline --no-f32-leg --pipeline-engines 0 --configs-block off --min-seconds 0 --config $cfg --steps 3 --warmup 2 > $O/$v.log 2>&1
  f=$(find /tmp/rbp -name "*kernel_stats.csv" | head -1)
  python - $f "c$cfg $v" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "resblock_bf3" in r["Name"]]
print("%-12s" % sys.argv[2], "  ".join("%s calls %s avg %.1f us" % (r["Name"].split("kernel")[1][:18], r["Calls"], float(r["AverageNs"]) / 1e3) for r in rows))
PY
done; done

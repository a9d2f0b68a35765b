# Code-freeze session (one gpurun call): the full GPU test suite, then -- only if it is green -- the round-end evidence (tools/final_session.sh).
#   usage: bash tools/freeze_session.sh [tag]
TAG=${1:-r04z}
mkdir -p gpurun_out/$TAG
timeout 1000 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_gpu_full.txt 2>&1
tail -6 gpurun_out/$TAG/pytest_gpu_full.txt
# (the summary line is not the last one: library log lines written at exit follow it)
if ! grep -qE "^[0-9]+ passed" gpurun_out/$TAG/pytest_gpu_full.txt || grep -qE "^(FAILED|ERROR)|[0-9]+ (failed|error)" gpurun_out/$TAG/pytest_gpu_full.txt; then echo "GPU suite not green: evidence session skipped"; exit 1; fi
bash tools/final_session.sh $TAG

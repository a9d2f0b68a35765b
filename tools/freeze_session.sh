# Code-freeze session (one gpurun call): the full GPU test suite, then -- only if it is green -- the round-end evidence (tools/final_session.sh).
#   usage: bash tools/freeze_session.sh [tag]
TAG=${1:-r04z}
mkdir -p gpurun_out/$TAG
timeout 1000 python -m pytest tests -m gpu -x -q > gpurun_out/$TAG/pytest_gpu_full.txt 2>&1
tail -6 gpurun_out/$TAG/pytest_gpu_full.txt
if ! tail -3 gpurun_out/$TAG/pytest_gpu_full.txt | grep -q " passed" || tail -3 gpurun_out/$TAG/pytest_gpu_full.txt | grep -q "failed\|error"; then echo "GPU suite not green: evidence session skipped"; exit 1; fi
bash tools/final_session.sh $TAG

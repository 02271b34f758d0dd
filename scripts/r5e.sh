# round 5: option stale_factor -- test and study
set -eu
cd "${GRAFT_REPO_ROOT:?not on a GPU box: GRAFT_REPO_ROOT is unset}"
. scripts/r5_lib.sh
mkdir -p gpurun_out
run 300 gpurun_out/r5e_tests.log python -m pytest -x -q tests/test_gpu_operators.py
tail -5 gpurun_out/r5e_tests.log
run 400 gpurun_out/r5e_stale.txt python scripts/r5_stale_factor.py
cut -c1-230 gpurun_out/r5e_stale.txt; tail -5 gpurun_out/r5e_stale.err

# round 5: CG2CR1 parity test
set -eu
cd "${GRAFT_REPO_ROOT:?not on a GPU box: GRAFT_REPO_ROOT is unset}"
. scripts/r5_lib.sh
mkdir -p gpurun_out
run 600 gpurun_out/r5h_cr.log python -m pytest -q tests/test_gpu_parity.py -k cg2cr1
tail -25 gpurun_out/r5h_cr.log

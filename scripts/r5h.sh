# round 5: CG2CR1 parity, transient march and CSR export
set -eu
cd "${GRAFT_REPO_ROOT:?not on a GPU box: GRAFT_REPO_ROOT is unset}"
. scripts/r5_lib.sh
mkdir -p gpurun_out
run 600 gpurun_out/r5h_cr.log python -m pytest -q tests/test_gpu_parity.py tests/test_gpu_dynamic.py -k "cg2cr1 or CG2CR1 or csr"
tail -25 gpurun_out/r5h_cr.log

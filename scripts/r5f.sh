# round 5: the whole GPU suite (call audit of the C-ABI entry points on)
set -eu
cd "${GRAFT_REPO_ROOT:?not on a GPU box: GRAFT_REPO_ROOT is unset}"
. scripts/r5_lib.sh
mkdir -p gpurun_out
FEMO_CALL_AUDIT=gpurun_out/r5_entry_point_calls.json run 1100 gpurun_out/r5f_tests.log python -m pytest tests -q -m gpu --durations=8 -x
tail -25 gpurun_out/r5f_tests.log

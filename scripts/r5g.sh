# round 5: idle quarter of diagonal tiles skipped in k_trailing_mfma -- fuzz, goldens, level table, factorisation time
set -eu
cd "${GRAFT_REPO_ROOT:?not on a GPU box: GRAFT_REPO_ROOT is unset}"
. scripts/r5_lib.sh
mkdir -p gpurun_out
run 600 gpurun_out/r5g_tests.log python -m pytest -x -q tests/test_gpu_schedules.py tests/test_gpu_goldens.py tests/test_gpu_building_blocks.py
tail -3 gpurun_out/r5g_tests.log
run 300 gpurun_out/r5g_levels.txt python scripts/r2_levels.py wing1m
cat gpurun_out/r5g_levels.txt

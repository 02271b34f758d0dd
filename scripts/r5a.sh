# round 5: the strip kernel of the HBM-bound rank-k updates (option strip_cnt) -- schedule fuzz, per-level tables with and without
set -eu
cd "${GRAFT_REPO_ROOT:?not on a GPU box: GRAFT_REPO_ROOT is unset}"
. scripts/r5_lib.sh
mkdir -p gpurun_out
run 600 gpurun_out/r5a_sched.log python -m pytest tests/test_gpu_schedules.py -x -q
tail -3 gpurun_out/r5a_sched.log
run 300 gpurun_out/r5a_levels_off.txt python scripts/r2_levels.py wing1m strip_cnt=0
for d in 1 2; do run 300 gpurun_out/r5a_levels_d$d.txt python scripts/r2_levels.py wing1m strip_cnt=256 strip_depth=$d; done
for d in 1 2; do head -8 gpurun_out/r5a_levels_d$d.txt | cut -c1-60; done

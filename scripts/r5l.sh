# round 5: do the look-ahead schedules fail because five streams share four hardware queues?  (ROCm maps HIP streams onto
# GPU_MAX_HW_QUEUES = 4 hardware queues by default; streams that share one serialise)
set -eu
cd "${GRAFT_REPO_ROOT:?not on a GPU box: GRAFT_REPO_ROOT is unset}"
. scripts/r5_lib.sh
mkdir -p gpurun_out
run 300 gpurun_out/r5l_base.txt python scripts/r2_levels.py wing1m
run 300 gpurun_out/r5l_q8_base.txt env GPU_MAX_HW_QUEUES=8 python scripts/r2_levels.py wing1m
run 300 gpurun_out/r5l_q8_da.txt env GPU_MAX_HW_QUEUES=8 python scripts/r2_levels.py wing1m diag_ahead=1
run 300 gpurun_out/r5l_q8_spa.txt env GPU_MAX_HW_QUEUES=8 python scripts/r2_levels.py wing1m super_panel_ahead=1
run 300 gpurun_out/r5l_q8_both.txt env GPU_MAX_HW_QUEUES=8 python scripts/r2_levels.py wing1m lookahead_cnt=64 super_panel_cnt=0
run 300 gpurun_out/r5l_q8_split.txt env GPU_MAX_HW_QUEUES=8 python scripts/r2_levels.py wing1m split_cnt=16
for f in base q8_base q8_da q8_spa q8_both q8_split; do echo $f; tail -1 gpurun_out/r5l_$f.txt; done

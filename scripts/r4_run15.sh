cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 300 gpurun_out/r4n_levels_a.txt python scripts/r2_levels.py wing1m
run 300 gpurun_out/r4n_levels_b.txt python scripts/r2_levels.py wing1m
run 600 gpurun_out/r4n_tests.log python -m pytest tests -q -m gpu -x -k "schedules or goldens or multifrontal or super_panel"
cat gpurun_out/r4n_levels_a.txt; tail -4 gpurun_out/r4n_levels_b.txt; tail -3 gpurun_out/r4n_tests.log

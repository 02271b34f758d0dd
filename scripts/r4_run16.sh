cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 500 gpurun_out/r4o_corun.txt python scripts/r4_corun.py wing1m plate250k
cat gpurun_out/r4o_corun.txt; tail -3 gpurun_out/r4o_corun.err

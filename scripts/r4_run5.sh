cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 900 gpurun_out/r4e_plan_sweep.txt python scripts/r4_plan_sweep.py wing1m
run 300 gpurun_out/r4e_test4.log python -m pytest tests/test_gpu_distributed.py -q -m gpu -k "four_ranks"
cat gpurun_out/r4e_plan_sweep.txt; tail -3 gpurun_out/r4e_test4.log

cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 900 gpurun_out/r4t_tests.log python -m pytest tests -q -m gpu
tail -5 gpurun_out/r4t_tests.log

cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 900 gpurun_out/r4f_tests.log python -m pytest tests -q -m gpu --durations=5 -x
bash scripts/r4_rocprof.sh > gpurun_out/r4f_rocprof.log 2>&1
tail -4 gpurun_out/r4f_tests.log; tail -30 gpurun_out/r4f_rocprof.log

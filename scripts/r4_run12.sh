cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 1000 gpurun_out/r4l_fuzz.txt python scripts/fuzz_schedules.py 404 80
tail -3 gpurun_out/r4l_fuzz.txt; tail -3 gpurun_out/r4l_fuzz.err

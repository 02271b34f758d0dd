cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 900 gpurun_out/r4_amdahl_weak.json python scripts/r3_amdahl.py wing1m weak
grep "^|" gpurun_out/r4_amdahl_weak.err

cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 1100 gpurun_out/r4w_cpu_full.txt python scripts/cpu_baseline_full.py wing1m gpurun_out/r4_cpu_baseline_wing1m.json 3
tail -6 gpurun_out/r4w_cpu_full.txt

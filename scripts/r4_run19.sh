cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
for w in plate250k wing1m wing4m; do
run 900 gpurun_out/r4r_sweep_$w.txt python scripts/r4_ab_sweep.py $w
cat gpurun_out/r4r_sweep_$w.txt
done

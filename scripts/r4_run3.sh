cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 400 gpurun_out/r4c_ab.txt python scripts/r4_ab.py wing1m "rows_fine_wg=0,narrow_fine_wg=0" "rows_fine_wg=96,narrow_fine_wg=0" "rows_fine_wg=256,narrow_fine_wg=0" "rows_fine_wg=768,narrow_fine_wg=0" "rows_fine_wg=96,narrow_fine_wg=64" "rows_fine_wg=96,narrow_fine_wg=128" "rows_fine_wg=256,narrow_fine_wg=256"
run 300 gpurun_out/r4c_sweeps.txt python scripts/r3_sweeps.py wing1m
run 400 gpurun_out/r4c_bench_wing1m.json python bench.py --steps 20 --warmup 3 --no-cpu-baseline
cat gpurun_out/r4c_ab.txt

# round 4, second pass: tests with the extended-precision goldens, the equilibration experiment, the fine rows / narrow kernels on and off
cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 600 gpurun_out/r4b_tests.log python -m pytest tests -q -m gpu --durations=8
run 200 gpurun_out/r4b_levels_base.txt python scripts/r2_levels.py wing1m rows_fine_wg=0 narrow_fine_wg=0
run 200 gpurun_out/r4b_levels_rows.txt python scripts/r2_levels.py wing1m rows_fine_wg=768 narrow_fine_wg=0
run 200 gpurun_out/r4b_levels_both.txt python scripts/r2_levels.py wing1m rows_fine_wg=768 narrow_fine_wg=512
run 200 gpurun_out/r4b_levels_big.txt python scripts/r2_levels.py wing1m rows_fine_wg=4096 narrow_fine_wg=2048
run 400 gpurun_out/r4b_equilibrate.txt python scripts/r4_equilibrate.py
run 400 gpurun_out/r4b_bench_wing1m.json python bench.py --steps 20 --warmup 3 --no-cpu-baseline
tail -6 gpurun_out/r4b_tests.log; grep -h "factor_ms\|class totals" gpurun_out/r4b_levels_*.txt

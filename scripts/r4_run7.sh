cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 900 gpurun_out/r4g_tests.log python -m pytest tests -q -m gpu --durations=5
run 300 gpurun_out/r4g_bench_wing1m.json python bench.py --steps 20 --warmup 3 --no-cpu-baseline
run 1100 gpurun_out/r4g_cpu_full.txt python scripts/cpu_baseline_full.py wing1m gpurun_out/r4_cpu_baseline_wing1m.json 3
tail -5 gpurun_out/r4g_tests.log; tail -4 gpurun_out/r4g_cpu_full.txt

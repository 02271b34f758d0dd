# round 4, first pass on the GPU box: tests, the two bench workloads, the per-level table
cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 600 gpurun_out/r4a_tests.log python -m pytest tests -q -m gpu --durations=8
run 400 gpurun_out/r4a_bench_wing1m.json python bench.py --steps 20 --warmup 3
run 400 gpurun_out/r4a_bench_dyn.json python bench.py --workload plate500k_dynamic
run 200 gpurun_out/r4a_levels.txt python scripts/r2_levels.py wing1m
tail -5 gpurun_out/r4a_tests.log; tail -c 600 gpurun_out/r4a_bench_wing1m.err; tail -c 600 gpurun_out/r4a_bench_dyn.err

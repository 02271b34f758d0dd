# round 4, fourth pass: the whole GPU suite (CG1CG1, four ranks at full size), the strong-scaling composition with the new plan, both bench workloads
cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 900 gpurun_out/r4d_tests.log python -m pytest tests -q -m gpu --durations=8
run 600 gpurun_out/r4d_amdahl.json python scripts/r3_amdahl.py wing1m
run 400 gpurun_out/r4d_levels.txt python scripts/r2_levels.py wing1m
run 500 gpurun_out/r4d_bench_wing1m.json python bench.py --steps 40 --warmup 5
run 400 gpurun_out/r4d_bench_dyn.json python bench.py --workload plate500k_dynamic
run 300 gpurun_out/r4d_bench_plate250k.json python bench.py --workload plate250k --steps 40 --warmup 5 --no-cpu-baseline
tail -12 gpurun_out/r4d_tests.log; tail -c 400 gpurun_out/r4d_amdahl.err

# Final measurement pass of round 4 on the GPU box: full GPU tests (with the call audit of the C-ABI entry points), the bench
# workloads as the driver runs them, the per-level tables, the larger skins, then the rocprofv3 passes (scripts/r4_rocprof.sh).
# Everything lands in gpurun_out/; what is judged is copied into profiles/ afterwards.
set -eu
cd "${GRAFT_REPO_ROOT:?not on a GPU box: GRAFT_REPO_ROOT is unset}"
. scripts/r4_lib.sh
mkdir -p gpurun_out
FEMO_CALL_AUDIT=gpurun_out/r4_entry_point_calls.json run 900 gpurun_out/r4z_tests.log python -m pytest tests -q -m gpu --durations=5
tail -3 gpurun_out/r4z_tests.log
run 600 gpurun_out/r4z_bench_wing1m.json python bench.py
run 300 gpurun_out/r4z_bench_plate250k.json python bench.py --workload plate250k
run 400 gpurun_out/r4z_dynamic_500k.json python bench.py --workload plate500k_dynamic
run 400 gpurun_out/r4z_bench_uskin1m.json python bench.py --workload uskin1m --steps 40 --no-keep-numbering-leg
run 300 gpurun_out/r4z_levels_wing1m.txt python scripts/r2_levels.py wing1m
run 300 gpurun_out/r4z_sweeps_wing1m.txt python scripts/r3_sweeps.py wing1m
run 300 gpurun_out/r4z_smoke.txt python -c "import __graft_entry__ as g; g.smoke()"
# a step of the profile pass that is killed at its limit stops this script as well: no further GPU step on that box
bash scripts/r4_rocprof.sh > gpurun_out/r4z_rocprof.log 2>&1 || { echo 'rocprof pass stopped'; exit 1; }
for w in wing4m wing8m wing16m wing32m wing48m; do
  run 500 gpurun_out/r4z_bench_$w.json python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-keep-numbering-leg
done
tail -c 400 gpurun_out/r4z_bench_wing1m.json; tail -3 gpurun_out/r4z_rocprof.log; tail -2 gpurun_out/r4z_smoke.txt

# Final measurement pass of round 6 on the GPU box: the whole GPU suite (with the call audit of the C-ABI entry points), the bench
# workloads as the driver runs them (with the committed counter passes of scripts/r6_rocprof.sh in place, so that roofline.traffic is
# filled), the per-level tables, the triangle-rule study, the multi-right-hand-side sweeps, the schedule fuzz.
# Everything lands in gpurun_out/; what is judged is copied into profiles/ afterwards.
set -eu
cd "${GRAFT_REPO_ROOT:?not on a GPU box: GRAFT_REPO_ROOT is unset}"
. scripts/r6_lib.sh
mkdir -p gpurun_out
FEMO_CALL_AUDIT=gpurun_out/r6_entry_point_calls.json run 900 gpurun_out/r6z_tests.log python -m pytest tests -q -m gpu --durations=5
tail -3 gpurun_out/r6z_tests.log
run 600 gpurun_out/r6z_bench_wing1m.json python bench.py
run 300 gpurun_out/r6z_bench_plate250k.json python bench.py --workload plate250k
run 400 gpurun_out/r6z_dynamic_500k.json python bench.py --workload plate500k_dynamic
run 400 gpurun_out/r6z_bench_uquad1m.json python bench.py --workload uquad1m --steps 40
run 400 gpurun_out/r6z_bench_uskin1m.json python bench.py --workload uskin1m --steps 40 --no-keep-numbering-leg
run 400 gpurun_out/r6z_bench_wing1m_tri.json python bench.py --workload wing1m_tri --steps 40 --no-keep-numbering-leg
run 300 gpurun_out/r6z_levels_wing1m.txt python scripts/r2_levels.py wing1m
run 300 gpurun_out/r6z_sweeps_wing1m.txt python scripts/r3_sweeps.py wing1m
run 300 gpurun_out/r6z_sweeps_nrhs.txt python scripts/r6_sweeps_nrhs.py wing1m
run 400 gpurun_out/r6z_quadrature_uskin1m.txt python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k triangle_rule_sensitivity
run 300 gpurun_out/r6z_smoke.txt python -c "import __graft_entry__ as g; g.smoke()"
run 600 gpurun_out/r6z_fuzz.txt python scripts/fuzz_schedules.py 5 60
tail -c 400 gpurun_out/r6z_bench_wing1m.json; tail -2 gpurun_out/r6z_smoke.txt; tail -1 gpurun_out/r6z_fuzz.txt

# round 5: the 6 x 6 rule on the unstructured quadrilateral skin (goldens at n = 6 and n = 5, quadrature steps, properties, bench line); wing1m unchanged
set -eu
cd "${GRAFT_REPO_ROOT:?not on a GPU box: GRAFT_REPO_ROOT is unset}"
. scripts/r5_lib.sh
mkdir -p gpurun_out
run 900 gpurun_out/r5c_tests.log python -m pytest -x -q -s "tests/test_gpu_fullsize.py::test_full_size_properties[uquad1m]" tests/test_gpu_fullsize.py::test_quadrature_rule_sensitivity_at_config3 tests/test_gpu_goldens.py::test_parity_triple_on_the_unstructured_and_triangle_skins tests/test_gpu_goldens.py::test_parity_triple_against_the_one_million_dof_golden tests/test_gpu_parity.py
tail -15 gpurun_out/r5c_tests.log
run 500 gpurun_out/r5c_bench_uquad1m.json python bench.py --workload uquad1m --steps 40
run 500 gpurun_out/r5c_bench_wing1m.json python bench.py --steps 40 --no-cpu-baseline
tail -c 300 gpurun_out/r5c_bench_uquad1m.err

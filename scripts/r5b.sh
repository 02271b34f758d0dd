# round 5: the unstructured quadrilateral skin (golden, properties, quadrature rule, bench line) and config 5 at the parity setting
set -eu
cd "${GRAFT_REPO_ROOT:?not on a GPU box: GRAFT_REPO_ROOT is unset}"
. scripts/r5_lib.sh
mkdir -p gpurun_out
run 900 gpurun_out/r5b_tests.log python -m pytest -x -q -s tests/test_gpu_goldens.py -k "unstructured or product_default" "tests/test_gpu_fullsize.py::test_full_size_properties[uquad1m]" "tests/test_gpu_fullsize.py::test_quadrature_rule_sensitivity_at_config3[uquad1m]"
tail -12 gpurun_out/r5b_tests.log
run 500 gpurun_out/r5b_bench_uquad1m.json python bench.py --workload uquad1m --steps 40
run 500 gpurun_out/r5b_dynamic_500k.json python bench.py --workload plate500k_dynamic
tail -c 1500 gpurun_out/r5b_bench_uquad1m.err; tail -c 600 gpurun_out/r5b_dynamic_500k.err

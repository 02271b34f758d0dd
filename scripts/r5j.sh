# round 5: first outer panel gathered from the children inside the diagonal-block and row kernels (option panel_gather = fewest fronts
# of a level that does so): schedule / operator / parity tests, per-level tables with 0 (off), 512, 1 (every level)
set -eu
cd "${GRAFT_REPO_ROOT:?not on a GPU box: GRAFT_REPO_ROOT is unset}"
. scripts/r5_lib.sh
mkdir -p gpurun_out
run 600 gpurun_out/r5j_tests.log python -m pytest tests/test_gpu_schedules.py tests/test_gpu_operators.py tests/test_gpu_parity.py tests/test_gpu_building_blocks.py -q -m gpu -x
tail -3 gpurun_out/r5j_tests.log
for w in 512 1 0; do
  run 300 gpurun_out/r5j_levels_pg$w.txt python scripts/r2_levels.py wing1m panel_gather=$w
done
for w in 512 0; do
  run 400 gpurun_out/r5j_bench_pg$w.json env FEMO_OPTIONS=panel_gather=$w python bench.py --no-cpu-baseline
done
for w in 512 1 0; do tail -3 gpurun_out/r5j_levels_pg$w.txt; done
for w in 512 0; do python - <<PY
import json
d=json.loads(open('gpurun_out/r5j_bench_pg$w.json').read().strip().splitlines()[-1])
print('panel_gather=$w', d['value'], d['ms_per_step'], d['forward_ms'], d['adjoint_ms'], d['forward_split_ms'])
PY
done

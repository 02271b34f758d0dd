# round 5: W form of the wide levels' sweeps (option sweep_w): correctness on the schedule tests, per-level sweep and factorisation
# tables with and without, bench with and without
set -eu
cd "${GRAFT_REPO_ROOT:?not on a GPU box: GRAFT_REPO_ROOT is unset}"
. scripts/r5_lib.sh
mkdir -p gpurun_out
run 600 gpurun_out/r5i_tests.log python -m pytest tests/test_gpu_schedules.py tests/test_gpu_operators.py tests/test_gpu_parity.py -q -m gpu -x
tail -3 gpurun_out/r5i_tests.log
for w in 1 0; do
  run 300 gpurun_out/r5i_sweeps_w$w.txt python scripts/r3_sweeps.py wing1m sweep_w=$w
  run 300 gpurun_out/r5i_levels_w$w.txt python scripts/r2_levels.py wing1m sweep_w=$w
  run 400 gpurun_out/r5i_bench_w$w.json env FEMO_OPTIONS=sweep_w=$w python bench.py --no-cpu-baseline
done
for w in 1 0; do tail -4 gpurun_out/r5i_sweeps_w$w.txt; tail -c 300 gpurun_out/r5i_bench_w$w.err; python - <<PY
import json
d=json.loads(open('gpurun_out/r5i_bench_w$w.json').read().strip().splitlines()[-1])
print('sweep_w=$w', d['value'], d['ms_per_step'], d['forward_ms'], d['adjoint_ms'], d['forward_split_ms'], d['preconditioner_apply'])
PY
done

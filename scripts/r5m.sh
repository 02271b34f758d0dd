# round 5: front-centric assembly (option assemble_fc): the whole GPU suite with it on, per-level tables and bench lines with and without
set -eu
cd "${GRAFT_REPO_ROOT:?not on a GPU box: GRAFT_REPO_ROOT is unset}"
. scripts/r5_lib.sh
mkdir -p gpurun_out
run 900 gpurun_out/r5m_tests.log python -m pytest tests -q -m gpu -x
tail -3 gpurun_out/r5m_tests.log
for w in 1 0; do
  run 300 gpurun_out/r5m_levels_fc$w.txt python scripts/r2_levels.py wing1m assemble_fc=$w
  run 400 gpurun_out/r5m_bench_wing1m_fc$w.json env FEMO_OPTIONS=assemble_fc=$w python bench.py --no-cpu-baseline
  run 400 gpurun_out/r5m_bench_uskin1m_fc$w.json env FEMO_OPTIONS=assemble_fc=$w python bench.py --workload uskin1m --steps 40 --no-keep-numbering-leg --no-cpu-baseline
  run 400 gpurun_out/r5m_dynamic_fc$w.json env FEMO_OPTIONS=assemble_fc=$w python bench.py --workload plate500k_dynamic --no-cpu-baseline
done
for w in 1 0; do tail -3 gpurun_out/r5m_levels_fc$w.txt
python - <<PY
import json
for wl in ('bench_wing1m', 'bench_uskin1m'):
    d=json.loads(open('gpurun_out/r5m_%s_fc$w.json' % wl).read().strip().splitlines()[-1])
    print('assemble_fc=$w', wl, d['value'], d['ms_per_step'], d['forward_ms'], d['adjoint_ms'], d['forward_split_ms'], d['factorisation_profile_ms']['front_assemble'], d['factorisation_profile_ms']['memset'])
d=json.loads(open('gpurun_out/r5m_dynamic_fc$w.json').read().strip().splitlines()[-1]); print('assemble_fc=$w dynamic', d['value'], d['ms_per_step'])
PY
done

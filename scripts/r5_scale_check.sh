# the front-centric assembly at larger sizes on one GPU (4 M and 16 M DOF: 32 k and 131 k leaf fronts), with and without
set -eu
cd "${GRAFT_REPO_ROOT:?not on a GPU box: GRAFT_REPO_ROOT is unset}"
. scripts/r5_lib.sh
mkdir -p gpurun_out
for w in wing4m wing16m; do for fc in 1 0; do
  run 500 gpurun_out/r5s_${w}_fc$fc.json env FEMO_OPTIONS=assemble_fc=$fc python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline --no-keep-numbering-leg
  python - <<PY
import json
d=json.loads(open('gpurun_out/r5s_${w}_fc$fc.json').read().strip().splitlines()[-1])
print('$w assemble_fc=$fc', d['config']['ndof'], round(d['value']/1e6,2), 'M DOF/s forward', round(d['forward_ms'],2), 'adjoint', round(d['adjoint_ms'],2), d['config'].get('pcg_iterations_forward'), d['config'].get('true_relres_forward'), d['factorisation_profile_ms']['front_assemble'], d['factorisation_profile_ms']['memset'])
PY
done; done

# The capacity cases of round 4 (profiles/r4_bench_wing{4,16,48}m.json) with the round-6 sources: 4.1 M, 16.3 M and 48.7 M DOF on one GPU.
set -eu
cd "${GRAFT_REPO_ROOT:?not on a GPU box: GRAFT_REPO_ROOT is unset}"
. scripts/r6_lib.sh
mkdir -p gpurun_out
for w in wing4m wing16m wing48m; do
  run 500 gpurun_out/r6_bench_$w.json python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-keep-numbering-leg
  tail -c 300 gpurun_out/r6_bench_$w.err
done

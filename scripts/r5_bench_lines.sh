# the bench lines of profiles/r5_bench_*.json, with the committed counter passes (profiles/pmc_wing1m.json) in place so that
# roofline.traffic is filled
set -eu
cd "${GRAFT_REPO_ROOT:?not on a GPU box: GRAFT_REPO_ROOT is unset}"
. scripts/r5_lib.sh
mkdir -p gpurun_out
run 600 gpurun_out/r5z_bench_wing1m.json python bench.py
run 300 gpurun_out/r5z_bench_plate250k.json python bench.py --workload plate250k
run 400 gpurun_out/r5z_bench_uquad1m.json python bench.py --workload uquad1m --steps 40
run 400 gpurun_out/r5z_bench_uskin1m.json python bench.py --workload uskin1m --steps 40 --no-keep-numbering-leg
tail -c 300 gpurun_out/r5z_bench_wing1m.json

# round 5: fused sweeps of the wide levels (option sweep_fuse) -- per-level sweep tables with and without, read modes
set -eu
cd "${GRAFT_REPO_ROOT:?not on a GPU box: GRAFT_REPO_ROOT is unset}"
. scripts/r5_lib.sh
mkdir -p gpurun_out
run 300 gpurun_out/r5d_sweeps_off.txt python scripts/r3_sweeps.py wing1m sweep_fuse=0
for m in 0 1 2; do run 300 gpurun_out/r5d_sweeps_m$m.txt python scripts/r3_sweeps.py wing1m sweep_fuse=1 sweep_read_mode=$m; done
tail -2 gpurun_out/r5d_sweeps_off.txt; for m in 0 1 2; do grep -E "^    4 |^   13 |total" gpurun_out/r5d_sweeps_m$m.txt; done

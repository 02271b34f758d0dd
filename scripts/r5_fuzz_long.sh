# a longer schedule fuzz on the final sources: ten seeds of 120 random option sets each (two meshes per set)
set -eu
cd "${GRAFT_REPO_ROOT:?not on a GPU box: GRAFT_REPO_ROOT is unset}"
. scripts/r5_lib.sh
mkdir -p gpurun_out
: > gpurun_out/r5_fuzz_long.txt
for seed in 11 12 13 14 15 16 17 18 19 20; do
  run 400 gpurun_out/r5_fuzz_s$seed.txt python scripts/fuzz_schedules.py $seed 120
  echo "seed $seed: $(tail -1 gpurun_out/r5_fuzz_s$seed.txt)" | tee -a gpurun_out/r5_fuzz_long.txt
done

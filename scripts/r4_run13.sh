cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
for seed in 11 12 13; do
  run 900 gpurun_out/r4m_fuzz_$seed.txt python scripts/fuzz_schedules.py $seed 300
  tail -1 gpurun_out/r4m_fuzz_$seed.txt
done

cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 900 gpurun_out/r4i_tests.log python -m pytest tests -q -m gpu -x
tail -3 gpurun_out/r4i_tests.log
for w in wing4m wing8m wing16m wing32m wing48m; do
  run 500 gpurun_out/r4i_bench_$w.json python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-keep-numbering-leg
  tail -c 300 gpurun_out/r4i_bench_$w.err
done

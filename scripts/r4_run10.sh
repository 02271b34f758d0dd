cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
for w in wing32m wing48m; do
  run 500 gpurun_out/r4i_bench_$w.json python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-keep-numbering-leg
  tail -c 300 gpurun_out/r4i_bench_$w.err
done

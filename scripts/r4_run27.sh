cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 600 gpurun_out/r4x_amdahl.json python scripts/r3_amdahl.py wing1m
cp gpurun_out/r4x_amdahl.err gpurun_out/r4x_amdahl.md
bash scripts/r4_rocprof.sh > gpurun_out/r4x_rocprof.log 2>&1
run 300 gpurun_out/r4x_bench_wing1m.json python bench.py --steps 40
tail -12 gpurun_out/r4x_amdahl.md; tail -c 300 gpurun_out/r4x_bench_wing1m.json

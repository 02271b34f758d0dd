cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 900 gpurun_out/r4_last_tests.log python -m pytest tests -x -q -m gpu
tail -2 gpurun_out/r4_last_tests.log
run 300 gpurun_out/r4_last_smoke.txt python -c "import __graft_entry__ as g; g.build(); g.smoke()"
tail -1 gpurun_out/r4_last_smoke.txt
run 600 gpurun_out/r4_last_bench.json python bench.py
tail -c 500 gpurun_out/r4_last_bench.json

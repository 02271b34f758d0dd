cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 600 gpurun_out/r4q_tests.log python -m pytest tests/test_gpu_building_blocks.py -q -m gpu
tail -60 gpurun_out/r4q_tests.log

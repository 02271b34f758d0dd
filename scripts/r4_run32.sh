cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 600 gpurun_out/r4z_tests_fullsize.log python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k "full_size_properties"
tail -3 gpurun_out/r4z_tests_fullsize.log

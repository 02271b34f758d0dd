cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 800 gpurun_out/r4_shape_full.log python -m pytest tests/test_gpu_fullsize.py -q -m gpu -k shape_gradient_at --durations=2
tail -25 gpurun_out/r4_shape_full.log

cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 600 gpurun_out/r4_golden_c5.log python -m pytest tests/test_gpu_goldens.py -q -m gpu -k config5
tail -30 gpurun_out/r4_golden_c5.log

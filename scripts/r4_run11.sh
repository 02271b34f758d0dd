cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 600 gpurun_out/r4k_tests.log python -m pytest tests/test_gpu_parity.py tests/test_gpu_dynamic.py -q -m gpu -k "cg1cg1 or march_and_adjoint"
tail -40 gpurun_out/r4k_tests.log

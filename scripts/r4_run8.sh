cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 400 gpurun_out/r4h_precond_nquad.txt python scripts/r4_precond_nquad.py
run 900 gpurun_out/r4h_tests.log python -m pytest tests -q -m gpu -x
cat gpurun_out/r4h_precond_nquad.txt; tail -3 gpurun_out/r4h_tests.log

cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
FEMO_CALL_AUDIT=gpurun_out/r4_entry_point_calls.json run 900 gpurun_out/r4_all_tests.log python -m pytest tests -q -m gpu --durations=8
tail -14 gpurun_out/r4_all_tests.log

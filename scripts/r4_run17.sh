cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
export FEMO_CALL_AUDIT=gpurun_out/r4_entry_point_calls.json
run 900 gpurun_out/r4p_tests.log python -m pytest tests -q -m gpu
tail -3 gpurun_out/r4p_tests.log
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_entry_point_calls.json"))
print(len(d), "entry points;", sum(v == 0 for v in d.values()), "never called:", [k for k, v in d.items() if v == 0])
PY

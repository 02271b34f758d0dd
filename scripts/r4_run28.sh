cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 600 gpurun_out/r4y_tests.log python -m pytest tests/test_gpu_dynamic.py tests/test_gpu_fullsize.py -q -m gpu -x -k "march or config5 or dynamic or forward_mode"
tail -3 gpurun_out/r4y_tests.log
run 400 gpurun_out/r4y_dynamic_500k.json python bench.py --workload plate500k_dynamic --no-cpu-baseline
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4y_dynamic_500k.json").read().strip().splitlines()[-1])
print("ms per time step", d["config"]["ms_per_time_step"], "profile", d["factorisation_profile_ms"], "factor once", d["factor_once"]["march_ms"])
PY

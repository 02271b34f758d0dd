cd $GRAFT_REPO_ROOT && . scripts/r4_lib.sh && mkdir -p gpurun_out
run 900 gpurun_out/r4u_tests.log python -m pytest tests -q -m gpu -x
tail -3 gpurun_out/r4u_tests.log
run 400 gpurun_out/r4u_bench_wing1m.json python bench.py --steps 20 --warmup 3 --no-cpu-baseline
run 400 gpurun_out/r4u_bench_uskin1m.json python bench.py --workload uskin1m --steps 20 --warmup 3 --no-cpu-baseline --no-keep-numbering-leg
python - <<'PY'
import json
for w in ("wing1m", "uskin1m"):
    d = json.loads(open(f"gpurun_out/r4u_bench_{w}.json").read().strip().splitlines()[-1])
    print(w, "forward", round(d["forward_ms"], 2), "step", round(d["ms_per_step"], 2), "MDOF/s", round(d["value"] / 1e6, 1), "its", d["config"]["pcg_iterations_forward"], d["config"]["pcg_iterations_adjoint"], d["frontal"], "roofline", d["roofline"]["bound"], round(d["roofline"]["frac"], 3), d["setup_s"])
PY
tail -3 gpurun_out/r4u_bench_uskin1m.err

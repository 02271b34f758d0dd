"""The full CPU-baseline protocol of BASELINE.md section 3 / SURVEY.md section 8d on the SAME mesh as the GPU headline:
single core and all host cores, "as the reference runs it" and "best effort", medians of >= 5 repeats (SuperLU, which
takes minutes at 1 M DOF, once per core count), CPU model and core count in the record.

    python scripts/cpu_baseline_full.py [wing1m|plate250k] [out.json] [repeats]

Too long for bench.py's default run (its cpu_baseline leg times a bounded part of this); the result of this script is
committed under profiles/ and cited by DESIGN.md."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bench import make_workload                                  # noqa: E402
from femo_alpha_amd.solver.symbolic import build_plan           # noqa: E402
from oracle import cpu_baseline as cb                            # noqa: E402
from oracle.rm_shell_oracle import ShellOracle                   # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "wing1m"
    out = sys.argv[2] if len(sys.argv) > 2 else None
    repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    m, fields, marker, desc = make_workload(which)
    o = ShellOracle(m, penalty_facets=m.penalty_facets(marker))
    o.set_fields(h=fields["thickness"], E=fields["E"], nu=fields["nu"], rho=fields["density"], f=fields["F_solid"])
    t0 = time.perf_counter()
    plan = build_plan(m, 12)
    symbolic_s = time.perf_counter() - t0
    ncores = cb.host_cores()
    rec = dict(workload=f"{which}: {desc}", cpu_model=cb.cpu_model_name(), host_cores=ncores,
               cores_visible=len(os.sched_getaffinity(0)), symbolic_s=symbolic_s,
               kind="port", protocol="1 warm-up + median of N repeats per phase; SuperLU factorisation timed once per core count")
    print(f"{which}: {m.ndof} DOF, {ncores} host cores ({rec['cpu_model']}), symbolic analysis {symbolic_s:.1f} s", flush=True)
    for cores in (1, ncores):
        print(f"--- {cores} core(s)", flush=True)
        rec[f"cores_{cores}"] = cb.measure(o, plan, cores, repeats=repeats, superlu=True, log=lambda *a: print(*a, flush=True))
        if out:
            json.dump(rec, open(out, "w"), indent=1)
    for k in (f"cores_1", f"cores_{ncores}"):
        r = rec[k]
        print(k, "best effort %.0f DOF/s (%.2f s), as the reference runs it %.0f DOF/s (%.1f s), adjoint set-up %.1f s" % (
            r["best_effort"]["dof_per_s"], r["best_effort"]["forward_s"], r["as_reference"]["dof_per_s"],
            r["as_reference"]["forward_s"], r["as_reference"]["adjoint_setup_s"]))


if __name__ == "__main__":
    main()

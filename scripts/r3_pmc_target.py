"""Target program of the counter passes (scripts/r3_rocprof.sh): one mesh, two warm-up factorisations, then ONE factorisation
and a few operator applications -- the last `launches` rank-k updates in dispatch order are that factorisation's.
With an output path: the instrumented factorisation instead (no counters), writing per-launch level / time / flops / compulsory
bytes of the rank-k updates as JSON.
    python3 scripts/r3_pmc_target.py [wing1m] [meta.json]"""
import json, os, re, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import make_workload
from femo_alpha_amd.backend import ShellContext

which = sys.argv[1] if len(sys.argv) > 1 else "wing1m"
meta = sys.argv[2] if len(sys.argv) > 2 else None
m, fields, marker, desc = make_workload(which)
c = ShellContext(m)
for k, v in fields.items():
    c.set_field(k, v)
c.set_penalty_facets(m.penalty_facets(marker))
plan = c.enable_frontal(12)
c.set_solver(preconditioner=2, rtol=1e-10, maxit=30, check_every=1)
c.factorize(); c.factorize()
if meta is None:
    c.factorize()
    c.bench_kernel("apply", 5)
    c.sync()
else:
    c.set_option("profile_verbose", 1)
    sys.stderr.flush()
    tmp = tempfile.TemporaryFile(mode="w+b")
    old = os.dup(2); os.dup2(tmp.fileno(), 2)
    c.factorize_profile()
    os.dup2(old, 2); os.close(old)
    tmp.seek(0)
    rows = []
    for line in tmp.read().decode().splitlines():
        mm = re.match(r"prof level (\d+) class 2 ([\d.]+) us flops ([\d.e+-]+) bytes ([\d.e+-]+)", line)
        if mm:
            L = int(mm[1])
            rows.append(dict(level=L, fronts=len(plan.level_nodes[L]), us=float(mm[2]), flops=float(mm[3]), compulsory_bytes=float(mm[4])))
    json.dump(dict(workload=which, launches=rows), open(meta, "w"), indent=1)
    print(f"{len(rows)} rank-k update launches")

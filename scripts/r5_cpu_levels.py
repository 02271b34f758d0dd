"""Why the CPU restatement's factorisation is SLOWER at 64 threads than at 16: every tree level timed under
several (OpenMP threads over the fronts, BLAS threads inside a front) choices.    python scripts/r5_cpu_levels.py [wing1m]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
which = sys.argv[1] if len(sys.argv) > 1 else "wing1m"
sys.argv = [sys.argv[0]]
from threadpoolctl import threadpool_limits, threadpool_info
from bench import make_workload
from femo_alpha_amd.solver.symbolic import build_plan
from oracle import cpu_baseline as cb
from oracle.cpu_baseline import _i, _l, _d
from oracle.rm_shell_oracle import ShellOracle

m, fields, marker, desc = make_workload(which)
o = ShellOracle(m, penalty_facets=m.penalty_facets(marker))
o.set_fields(h=fields["thickness"], E=fields["E"], nu=fields["nu"], rho=fields["density"], f=fields["F_solid"])
cs = cb.CpuShell(o)
every = cb.host_cores(cap=64)
print("cores this process may use:", len(os.sched_getaffinity(0)), "-> at most", every, "threads;", [(i["internal_api"], i["num_threads"], i.get("threading_layer")) for i in threadpool_info()])
mf = cb.CpuMultifrontal(cs, build_plan(m, m.recommended_leaf_size()), every)
mf.factorize()                                  # first touch of the fronts by all threads
lib = cs.lib


def level(lev, omp, blas):
    with threadpool_limits(limits=blas):
        t0 = time.perf_counter()
        rc = lib.cpu_fronts_factor_level(lev.size, _i(lev), _i(mf.nf), _i(mf.npiv), _l(mf.front_off), _l(mf.dof_off), _i(mf.left), _i(mf.right),
                                         _i(mf.up_map), _d(mf.F), mf.ptr["potrf"], mf.ptr["trsm"], mf.ptr["syrk"], omp)
        assert rc == 0
        return time.perf_counter() - t0


for T in sorted({16, 32, every}):
    if T > every:
        continue
    mf.nthreads = T
    asm, fac = mf.factorize()
    print(f"\n{T} threads: CpuMultifrontal.factorize() assembly {asm:.3f} s, factorisation {fac:.3f} s")
    print("level fronts  max nf |   omp=min(n,T) blas=1 |  omp=1 blas=T | omp=1 blas=16 | omp=1 blas=8 | omp=min(n,T/4) blas=4 (nested)")
    tot = np.zeros(5)
    for L, lev in enumerate(mf.levels):
        # every variant needs the level's input state: re-assemble / re-factor the levels below (only the level's own time is read)
        ts = []
        for omp, blas in ((min(lev.size, T), 1), (1, T), (1, min(16, T)), (1, min(8, T)), (max(1, min(lev.size, T // 4)), 4)):
            mf.nthreads = T
            mf.s.lib  # noqa
            # restore the fronts below: factorising the whole tree again is the simplest correct way
            rc = None
            asm_, _ = 0, 0
            # assemble + levels < L
            mf_levels = mf.levels
            mf.levels = mf_levels[:L]
            mf.factorize()
            mf.levels = mf_levels
            ts.append(level(lev, omp, blas))
        tot += ts
        print(f"{L:5d} {lev.size:6d} {int(mf.nf[lev].max()):7d} | " + " | ".join(f"{t * 1e3:12.1f}" for t in ts), flush=True)
    print("sum (ms)             | " + " | ".join(f"{t * 1e3:12.1f}" for t in tot))

"""Experiment: do two independent factorisations that run side by side on one GPU (two contexts, two streams, two host threads)
finish sooner than one after the other?  If the lower tree levels are bound by latency and occupancy rather than by HBM, a second
stream of independent launches fills the idle slots."""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from femo_alpha_amd.backend import ShellContext


def make(name):
    m, fields, marker, _ = bench.make_workload(name)
    c = ShellContext(m)
    for k, v in fields.items():
        c.set_field(k, v)
    c.set_penalty_facets(m.penalty_facets(marker))
    c.enable_frontal(12)
    c.set_solver(preconditioner=2, rtol=1e-10, maxit=50, check_every=1)
    c.factorize()
    return c


def loop(c, n):
    for _ in range(n):
        c.factorize()


for name in sys.argv[1:] or ["wing1m", "plate250k"]:
    a, b = make(name), make(name)
    n = 20
    torch.cuda.synchronize(); t0 = time.perf_counter(); loop(a, n); torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    ths = [threading.Thread(target=loop, args=(c, n)) for c in (a, b)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize(); t2 = time.perf_counter() - t0
    print(f"{name}: one context {t1 / n * 1e3:.2f} ms per assembly + factorisation; two side by side {t2 / n * 1e3:.2f} ms per pair "
          f"= {t2 / t1:.2f} x one (2.00 = no overlap)", flush=True)
    a.close(); b.close()

"""Overhead of the Python-side distributed driver: the 1-rank DistributedShell on the 1M-DOF wing against the native
single-GPU path (bench.py: forward 22-23 ms, adjoint 4.2 ms)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from femo_alpha_amd.mesh import wing_skin_mesh
from femo_alpha_amd.parallel import Comm, DistributedShell

m = wing_skin_mesh(116, 580, span=6.0).renumbered()[0]
ds = DistributedShell(m, Comm(None), bc_marker=lambda x: np.less(x[1], 1e-9), leaf_size=12, device=0)
ds.rtol = 1e-10
fields = dict(thickness=np.array([1.27e-3]), E=np.array([73.1e9]), nu=np.array([0.33]), density=np.array([2780.0]),
              F_solid=np.tile([0.0, 0.0, -2780.0 * 1.27e-3 * 9.81], (m.nn, 1)))
ds.set_fields(**fields)
import gc
if os.environ.get('NOGC'): gc.disable()
for rep in range(24):
    ds.set_fields(thickness=fields["thickness"])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    it, rr = ds.solve_state()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    g, it2, rr2 = ds.total_gradient("compliance", "thickness")
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"rep {rep}: forward {1e3 * (t1 - t0):.2f} ms ({it} it, {rr:.1e})  adjoint {1e3 * (t2 - t1):.2f} ms ({it2} it)", flush=True)

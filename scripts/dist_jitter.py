"""Where does the 1-rank distributed driver lose 14 ms every third step?  Phase times per step."""
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
eng = ds.eng


def T():
    torch.cuda.synchronize(); return time.perf_counter()


for rep in range(12):
    t0 = T(); ds.set_fields(thickness=fields["thickness"])
    t1 = T(); eng.load("b"); ds._sum_top("b")
    t2 = T(); eng.factor(0, ds.nl, True)
    t3 = T(); eng.factor(ds.nl, ds.nlev, False); ds.factored = True
    t4 = T(); it, rr = ds._pcg("b", "state")
    t5 = T()
    print(f"rep {rep}: set_fields {1e3*(t1-t0):.2f} load {1e3*(t2-t1):.2f} factor_local {1e3*(t3-t2):.2f} factor_top {1e3*(t4-t3):.2f} pcg {1e3*(t5-t4):.2f}", flush=True)

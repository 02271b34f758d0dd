"""Phase timing of the distributed forward solve (rehearsal: several ranks may share one GPU through gloo)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from femo_alpha_amd.mesh import wing_skin_mesh
from femo_alpha_amd.parallel import Comm, DistributedShell

dist.init_process_group(backend=os.environ.get("FEMO_BENCH_BACKEND", "gloo"))
rank, world = dist.get_rank(), dist.get_world_size()
ns = int(os.environ.get("FEMO_BENCH_NS", "580"))
m = wing_skin_mesh(116, ns * world, span=6.0 * world * ns / 580.0).renumbered()[0]
ds = DistributedShell(m, Comm(dist), bc_marker=lambda x: np.less(x[1], 1e-9), device=0)
ds.rtol = 1e-10
ds.set_fields(thickness=np.array([1.27e-3]), E=np.array([73.1e9]), nu=np.array([0.33]), density=np.array([2780.0]),
              F_solid=np.tile([0.0, 0.0, -34.6], (m.nn, 1)))
eng, info = ds.eng, ds.info


def t():
    torch.cuda.synchronize(); dist.barrier(); return time.perf_counter()


for rep in range(3):
    t0 = t(); eng.factor(0, ds.nl, True)
    t1 = t()
    sizes = info["schur_sizes"]; cap = max(sizes) ** 2
    mine = eng.new_tensor(cap); eng.schur_get(info["root_front"], mine)
    t2 = t()
    blocks = ds.comm.allgather(mine)
    t3 = t()
    for q, blk in enumerate(blocks):
        if q != rank:
            eng.block_set(info["stub_fronts"][q], blk[: sizes[q] ** 2].contiguous())
    t4 = t()
    eng.factor(ds.nl, ds.nlev, False)
    t5 = t()
    ds.factored = True
    it, rr = ds.solve_state()
    t6 = t()
    if rank == 0:
        print(f"rep {rep}: local factor {1e3*(t1-t0):.1f} | schur_get {1e3*(t2-t1):.1f} | allgather {1e3*(t3-t2):.1f} | block_set {1e3*(t4-t3):.1f} | "
              f"top factor {1e3*(t5-t4):.1f} | pcg {1e3*(t6-t5):.1f} ms  its {it} rr {rr:.1e}", flush=True)
dist.destroy_process_group()

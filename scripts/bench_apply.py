import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femo_alpha_amd.mesh import plate_mesh, wing_skin_mesh
from femo_alpha_amd.backend import ShellContext
for name, m in (("plate 10x50", plate_mesh(2, 10, 10, 50)), ("plate 58x290", plate_mesh(2, 10, 58, 290)), ("wing 116x580", wing_skin_mesh(116, 580))):
    c = ShellContext(m)
    c.set_field("thickness", [0.01]); c.set_field("E", [1e9]); c.set_field("nu", [0.3])
    t = c.bench_kernel("apply", 200)
    b = 16.0 * m.ndof + 340.0 * m.nel
    print(f"{name}: ndof {m.ndof} apply {t*1e3:.1f} us  -> {b/t/1e6:.1f} GB/s algorithmic", flush=True)
    c.close()

import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femo_alpha_amd.mesh import plate_mesh, wing_skin_mesh
from femo_alpha_amd.backend import ShellContext
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
m = plate_mesh(2, 10, 58, 290) if which == "c2" else wing_skin_mesh(116, 580)
c = ShellContext(m)
for a in sys.argv[2:]:                       # schedule options, key=value
    c.set_option(a.split("=")[0], float(a.split("=")[1]))
c.set_field("thickness", [0.05]); c.set_field("E", [1e9]); c.set_field("nu", [0.3])
t = time.time(); info = c.enable_csr(); print(f"device map (radix sort of the contributions, once per mesh): {time.time()-t:.2f}s nnz={info['nnz']} contributions={m.nel * m.ldof ** 2}", flush=True)
for r in range(3):
    t = time.time(); K = c.assemble_csr(); dt = time.time() - t
    alg = 8.0 * info["nnz"] + 340.0 * m.nel
    tk = c.csr_timing
    print(f"assemble wall {dt*1e3:.1f} ms (incl. copy out): element matrices {tk['element_matrices_ms']:.2f} ms, scatter {tk['scatter_ms']:.2f} ms"
          f" -> {alg/((tk['element_matrices_ms']+tk['scatter_ms'])*1e-3)/1e9:.1f} GB/s algorithmic (B_asm = {alg/1e6:.0f} MB)", flush=True)

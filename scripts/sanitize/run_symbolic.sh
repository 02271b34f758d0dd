# csrc/symbolic.cpp (host C++) under AddressSanitizer + UndefinedBehaviorSanitizer, on the CPU: a small C++ harness builds plans of a
# quadrilateral and of an unstructured triangle skin for every axis rule, several leaf sizes and forced depths.  (GPU sanitizers are not
# available on the pool; this is the host library only.)      bash scripts/sanitize/run_symbolic.sh
set -e
cd "$(dirname "$0")/../.."
python3 - <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
from femo_alpha_amd.mesh import wing_skin_mesh, unstructured_skin_mesh
for name, m in (("quad", wing_skin_mesh(24, 60)), ("tri", unstructured_skin_mesh(20, 50))):
    xc = m.nodes[m.cells]; cent = xc.mean(axis=1); cext = xc.max(axis=1) - xc.min(axis=1)
    np.ascontiguousarray(m.cell_p2, dtype=np.int32).tofile(f'/tmp/sym_{name}_p2.bin')
    np.ascontiguousarray(cent).tofile(f'/tmp/sym_{name}_cent.bin'); np.ascontiguousarray(cext).tofile(f'/tmp/sym_{name}_cext.bin')
    np.ascontiguousarray(m.cell_dofs(), dtype=np.int32).tofile(f'/tmp/sym_{name}_dofs.bin')
    open(f'/tmp/sym_{name}_meta.txt', 'w').write(f"{m.nel} {m.nP2} {m.nV} {m.cell_p2.shape[1]} {m.cell_dofs().shape[1]}\n")
PY
g++ -O1 -g -std=c++17 -fopenmp -fsanitize=address,undefined -fno-omit-frame-pointer -I include -o /tmp/sym_harness scripts/sanitize/symbolic_harness.cpp femo_alpha_amd/csrc/symbolic.cpp
ASAN_OPTIONS=detect_leaks=1 OMP_NUM_THREADS=4 /tmp/sym_harness

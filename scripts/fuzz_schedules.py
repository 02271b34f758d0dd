"""Schedule fuzz: random combinations of the factorisation / sweep options on two medium meshes; every combination must give
the reference combination's solution (1e-9) in the same number of iterations.  Catches ordering and aliasing mistakes that a
single default schedule hides."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from femo_alpha_amd.backend import ShellContext
from femo_alpha_amd.mesh import plate_mesh, wing_skin_mesh

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncomb = int(sys.argv[2]) if len(sys.argv) > 2 else 24
CLAMP = lambda x: np.less(x[0], 3e-16)
cases = [("plate 64x64", plate_mesh(2.0, 5.0, 64, 64), CLAMP, False),
         ("wing 48x160 strong BC", wing_skin_mesh(48, 160, shuffle=True).renumbered()[0], lambda x: np.less(x[1], 1e-9), True)]


def solve(m, marker, strong, leaf, pre, post):
    c = ShellContext(m)
    r = np.random.default_rng(1)
    c.set_field("thickness", 0.02 * (1 + 0.3 * r.uniform(-1, 1, m.nn)))
    for k, v in (("E", [7e10]), ("nu", [0.3]), ("density", [2700.0])):
        c.set_field(k, v)
    c.set_field("F_solid", r.uniform(-1, 1, (m.nn, 3)))
    if strong:
        c.set_strong_dofs(m.locate_dofs_geometrical(marker))
    else:
        c.set_penalty_facets(m.penalty_facets(marker))
    for k, v in post.items():
        c.set_option(k, v)
    c.enable_frontal(leaf, **pre)
    c.set_solver(preconditioner=2, rtol=1e-12, maxit=30, check_every=1)
    c.factorize(); c.factorize()
    c.set_field("thickness", c.get_field("thickness"))      # the solve below factorises again, from cold: the path of option "sweep_ahead"
    it, rr = c.solve_state(True)
    w = c.get_state()
    g, it2, _ = c.total_gradient("compliance", "thickness")
    c.close()
    return it, w, g


bad = 0
for name, m, marker, strong in cases:
    it0, w0, g0 = solve(m, marker, strong, 8, {}, dict(super_panel=0, fused_schur=0, lookahead=0))
    print(f"{name}: {m.ndof} DOF, reference {it0} iterations", flush=True)
    for k in range(ncomb):
        post = dict(trailing=int(rng.integers(0, 3)), left_min=int(rng.choice([1, 8, 64, 4096])), left_max=int(rng.choice([16, 2048, 100000])),
                    super_panel=int(rng.choice([0, 200, 256, 384, 512])), super_panel_cnt=int(rng.choice([4, 64, 100000])),
                    super_panel_ahead=int(rng.integers(0, 2)), diag_ahead=int(rng.integers(0, 2)), rows_preload_wg=int(rng.choice([0, 64, 100000])), rows_fine_wg=int(rng.choice([0, 64, 96, 100000])), narrow_fine_wg=int(rng.choice([0, 16, 128, 100000])), narrow_split=int(rng.choice([1, 3, 4, 8])), narrow_split_wg=int(rng.choice([8, 1024, 100000])), super_tiles=int(rng.integers(0, 2)), super_tiles_min=int(rng.choice([1, 3, 8])), split_cnt=int(rng.choice([0, 16, 100000])), split_groups=int(rng.choice([2, 3, 4, 8])), fuse_rows=int(rng.integers(0, 2)), diag_t=int(rng.integers(0, 3)), sweep_ahead=int(rng.integers(0, 4)), sweep_graph=int(rng.integers(0, 2)), fuse_rows_cnt=int(rng.choice([1, 512])), fuse_rows_np=int(rng.choice([128, 256, 100000])), diag_v1_cnt=int(rng.choice([1, 512])), lookahead=int(rng.integers(0, 2)), lookahead_cnt=int(rng.choice([4, 16, 1000])),
                    fused_schur=int(rng.integers(0, 2)), diag_v1=int(rng.integers(0, 3)), big_tiles=int(rng.integers(0, 2)), big_min_wg=int(rng.choice([1, 64, 512])), grid_chunk=int(rng.choice([3, 17, 65535])), xinv_small_cnt=int(rng.choice([0, 32, 100000])),
                    strip_cnt=int(rng.choice([0, 1, 256])), strip_kmax=int(rng.choice([64, 128, 160])), sweep_fuse=int(rng.integers(0, 2)), sweep_w=int(rng.integers(0, 2)), assemble_fc=int(rng.choice([0, 1, 2])))
        pre = dict(wide_cnt=int(rng.choice([0, 8, 512, 100000])), wide_np=int(rng.choice([64, 512])), swork_slots=int(rng.choice([2, 100, 8192])))
        leaf = int(rng.choice([4, 8, 12, 20]))
        try:
            it, w, g = solve(m, marker, strong, leaf, pre, post)
            ew = np.abs(w - w0).max() / np.abs(w0).max(); eg = np.abs(g - g0).max() / np.abs(g0).max()
            ok = ew < 1e-9 and eg < 1e-8 and it <= it0 + 1
        except Exception as e:              # noqa: BLE001
            ok, ew, eg, it = False, -1, -1, repr(e)[:120]
        bad += not ok
        print(f"  {'ok  ' if ok else 'FAIL'} it {it} dw {ew:.1e} dg {eg:.1e} leaf {leaf} {pre} {post}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)

"""Random combinations of the factorisation / sweep schedule options (left- / right-looking thresholds, super-panels with
and without the second-stream update, look-ahead, fused extend-add, launch chunking, wide-level thresholds, scratch chunking,
leaf size): whatever the schedule, the factor is the same -- same iteration count, same displacement and gradient as the
plainest schedule.  (scripts/fuzz_schedules.py runs the same check over hundreds of combinations.)"""
import numpy as np
import pytest

from femo_alpha_amd.mesh import plate_mesh, wing_skin_mesh

pytestmark = pytest.mark.gpu


def _solve(m, marker, strong, leaf, pre, post):
    from femo_alpha_amd.backend import ShellContext
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
    c.factorize(); c.factorize()                     # twice: stream order must also hold across factorisations
    c.set_field("thickness", c.get_field("thickness"))      # ... and the solve factorises once more, from cold (the path of "sweep_ahead")
    it, _ = c.solve_state(True)
    w = c.get_state()
    g, _, _ = c.total_gradient("compliance", "thickness")
    c.close()
    return it, w, g


@pytest.mark.parametrize("case,seed", [("plate", 11), ("wing_strong", 12)])
def test_random_schedules_give_the_same_solution(case, seed):
    if case == "plate":
        m, marker, strong = plate_mesh(2.0, 5.0, 64, 64), (lambda x: np.less(x[0], 3e-16)), False
    else:
        m, marker, strong = wing_skin_mesh(32, 96, shuffle=True).renumbered()[0], (lambda x: np.less(x[1], 1e-9)), True
    it0, w0, g0 = _solve(m, marker, strong, 8, {}, dict(super_panel=0, fused_schur=0, lookahead=0))
    rng = np.random.default_rng(seed)
    for _ in range(8):
        post = dict(trailing=int(rng.integers(0, 3)), left_min=int(rng.choice([1, 8, 64, 4096])), left_max=int(rng.choice([16, 2048, 100000])),
                    super_panel=int(rng.choice([0, 200, 256, 384, 512])), super_panel_cnt=int(rng.choice([4, 64, 100000])),
                    super_panel_ahead=int(rng.integers(0, 2)), diag_ahead=int(rng.integers(0, 2)), rows_preload_wg=int(rng.choice([0, 64, 100000])), rows_fine_wg=int(rng.choice([0, 64, 100000])), narrow_fine_wg=int(rng.choice([0, 16, 100000])), narrow_split=int(rng.choice([1, 3, 4, 8])), narrow_split_wg=int(rng.choice([8, 1024, 100000])), super_tiles=int(rng.integers(0, 2)), super_tiles_min=int(rng.choice([1, 3, 8])), split_cnt=int(rng.choice([0, 16, 100000])), split_groups=int(rng.choice([2, 3, 4, 8])), fuse_rows=int(rng.integers(0, 2)), diag_t=int(rng.integers(0, 3)), sweep_ahead=int(rng.integers(0, 4)), sweep_graph=int(rng.integers(0, 2)), fuse_rows_cnt=int(rng.choice([1, 512])), fuse_rows_np=int(rng.choice([128, 256, 100000])), diag_v1_cnt=int(rng.choice([1, 512])), lookahead=int(rng.integers(0, 2)), lookahead_cnt=int(rng.choice([4, 16, 1000])),
                    fused_schur=int(rng.integers(0, 2)), diag_v1=int(rng.integers(0, 3)), big_tiles=int(rng.integers(0, 2)), big_min_wg=int(rng.choice([1, 64, 512])), grid_chunk=int(rng.choice([3, 17, 65535])), xinv_small_cnt=int(rng.choice([0, 32, 100000])),
                    strip_cnt=int(rng.choice([0, 1, 256])), strip_kmax=int(rng.choice([64, 128, 160])), sweep_fuse=int(rng.integers(0, 2)), sweep_w=int(rng.integers(0, 2)), assemble_fc=int(rng.choice([0, 1, 2])))
        pre = dict(wide_cnt=int(rng.choice([0, 8, 512, 100000])), wide_np=int(rng.choice([64, 512])), swork_slots=int(rng.choice([2, 100, 8192])))
        leaf = int(rng.choice([4, 8, 12, 20]))
        what = f"leaf {leaf} {pre} {post}"
        try:
            it, w, g = _solve(m, marker, strong, leaf, pre, post)
        except Exception as exc:                         # name the option set that broke the factorisation
            raise AssertionError(f"{what}: {exc}") from exc
        assert it <= it0 + 1, what
        assert np.abs(w - w0).max() < 1e-9 * np.abs(w0).max(), what
        assert np.abs(g - g0).max() < 1e-8 * np.abs(g0).max(), what


@pytest.mark.parametrize("sp", [200, 384])
def test_super_panel_between_multiples_of_the_outer_panel(sp):
    """A super-panel width that is not a multiple of 128 is rounded DOWN by the schedule; the decision to let the first
    rank-k update gather its columns from the children must use the rounded value too (200 -> 128: plain right-looking,
    where a gathering update after every panel would discard the earlier panels' updates)."""
    m, marker = plate_mesh(2.0, 5.0, 64, 64), (lambda x: np.less(x[0], 3e-16))
    it0, w0, g0 = _solve(m, marker, False, 8, {}, dict(super_panel=0, fused_schur=0, lookahead=0))
    it, w, g = _solve(m, marker, False, 8, {}, dict(trailing=2, super_panel=sp, super_panel_cnt=100000, fused_schur=1, lookahead=0))
    assert it <= it0 + 1
    assert np.abs(w - w0).max() < 1e-9 * np.abs(w0).max()
    assert np.abs(g - g0).max() < 1e-8 * np.abs(g0).max()


@pytest.mark.parametrize("sp,fused,case", [(256, 0, "plate"), (512, 1, "plate"), (384, 1, "wing_strong"), (512, 0, "wing_strong")])
def test_diagonal_look_ahead_inside_super_panels(sp, fused, case):
    """Option "diag_ahead": inside a super-panel only the two row tiles and three update tiles the next diagonal block needs stay
    on the main stream, the rest runs on a second stream beside that block.  Every level is forced onto the super-panel
    schedule here (all fronts of more than one panel go through the split), and with the look-ahead on the rows take the
    kernel that preloads S and the narrow updates are cut into four K slices; same factor as the plain schedule."""
    if case == "plate":
        m, marker, strong = plate_mesh(2.0, 5.0, 64, 64), (lambda x: np.less(x[0], 3e-16)), False
    else:
        m, marker, strong = wing_skin_mesh(32, 96, shuffle=True).renumbered()[0], (lambda x: np.less(x[1], 1e-9)), True
    it0, w0, g0 = _solve(m, marker, strong, 8, {}, dict(super_panel=0, fused_schur=0, lookahead=0))
    res = {}
    for da in (0, 1):
        res[da] = _solve(m, marker, strong, 8, {}, dict(trailing=2, super_panel=sp, super_panel_cnt=100000, fused_schur=fused, lookahead=0, diag_ahead=da, rows_preload_wg=100000 * da, narrow_split=1 + 3 * da, narrow_split_wg=100000))
        it, w, g = res[da]
        assert it <= it0 + 1
        assert np.abs(w - w0).max() < 1e-9 * np.abs(w0).max()
        assert np.abs(g - g0).max() < 1e-8 * np.abs(g0).max()
    assert np.abs(res[0][1] - res[1][1]).max() < 1e-10 * np.abs(w0).max()


def test_lighter_quadrature_rule_of_the_front_assembly():
    """Option "precond_nquad": the fronts are assembled with fewer Gauss points than the operator (wing skin: 5 x 5).  The factor
    is then the factor of a slightly different matrix -- still a preconditioner: the solution is the same (PCG iterates on the
    operator's residual), it only takes more iterations, which is why the option is off by default (profiles/r4_precond_nquad.txt)."""
    m, marker = wing_skin_mesh(32, 96, shuffle=True).renumbered()[0], (lambda x: np.less(x[1], 1e-9))
    assert m.recommended_nquad() == 5
    it0, w0, g0 = _solve(m, marker, True, 8, {}, {})
    it4, w4, g4 = _solve(m, marker, True, 8, {}, dict(precond_nquad=4))
    it3, w3, g3 = _solve(m, marker, True, 8, {}, dict(precond_nquad=3))
    assert it0 <= it4 <= it3 and it3 > it0
    for w, g in ((w4, g4), (w3, g3)):
        assert np.abs(w - w0).max() < 1e-9 * np.abs(w0).max()
        assert np.abs(g - g0).max() < 1e-8 * np.abs(g0).max()


@pytest.mark.parametrize("kind", ["quad", "tri"])
def test_node_order_of_the_fronts_gives_the_same_solution(kind):
    """Rows of a front in ascending node id (node_order 0, rounds 1-5) or along the separators (node_order 1, the default from round 6:
    a child's Schur block lands in long runs of consecutive parent rows): any order of the pivots inside a front is a valid elimination
    order, so iteration counts and solutions agree; and with node_order 1 ``up_map`` is increasing along every child's boundary rows."""
    from femo_alpha_amd.backend import ShellContext
    from femo_alpha_amd.mesh import quads_to_triangles
    from femo_alpha_amd.solver.symbolic import build_plan
    m = wing_skin_mesh(32, 96, shuffle=True).renumbered()[0]
    if kind == "tri":
        m = quads_to_triangles(m)
    marker = lambda x: np.less(x[1], 1e-9)
    res = {}
    for order in (0, 1):
        plan = build_plan(m, 8, node_order=order)
        if order == 1:
            for t in range(plan.ntree):
                up = plan.up_map[plan.dof_off[t] + plan.npiv[t]: plan.dof_off[t + 1]]
                assert np.all(np.diff(up) > 0), t
        c = ShellContext(m)
        r = np.random.default_rng(1)
        c.set_field("thickness", 0.02 * (1 + 0.3 * r.uniform(-1, 1, m.nn)))
        for k, v in (("E", [7e10]), ("nu", [0.3]), ("density", [2700.0])):
            c.set_field(k, v)
        c.set_field("F_solid", r.uniform(-1, 1, (m.nn, 3)))
        c.set_penalty_facets(m.penalty_facets(marker))
        c.enable_frontal(plan=plan)
        c.set_solver(preconditioner=2, rtol=1e-12, maxit=30, check_every=1)
        it, _ = c.solve_state(True)
        g, it2, _ = c.total_gradient("compliance", "thickness")
        res[order] = (it, c.get_state(), g)
        c.close()
    assert abs(res[0][0] - res[1][0]) <= 1
    assert np.abs(res[0][1] - res[1][1]).max() < 1e-9 * np.abs(res[1][1]).max()
    assert np.abs(res[0][2] - res[1][2]).max() < 1e-8 * np.abs(res[1][2]).max()


def test_sweep_ahead_gives_the_same_solution():
    """Option "sweep_ahead": the forward sweep of the first preconditioner application through the lower levels, started by the
    factorisation beside the chain of the top levels -- the same iteration count and (to the rounding of the sweeps' atomics) the same
    solution as the plain order, for every number of top levels left to the solve, also when it exceeds the tree's height, with a
    right-hand side of zero (nothing to iterate: the started sweep is joined all the same) and in the transient march."""
    from femo_alpha_amd.backend import ShellContext
    m, marker = wing_skin_mesh(32, 96, shuffle=True).renumbered()[0], (lambda x: np.less(x[1], 1e-9))
    res = {}
    for ahead in (0, 1, 2, 3, 50):
        c = ShellContext(m)
        r = np.random.default_rng(1)
        h = 0.02 * (1 + 0.3 * r.uniform(-1, 1, m.nn))
        c.set_field("thickness", h)
        for k, v in (("E", [7e10]), ("nu", [0.3]), ("density", [2700.0])):
            c.set_field(k, v)
        c.set_field("F_solid", r.uniform(-1, 1, (m.nn, 3)))
        c.set_penalty_facets(m.penalty_facets(marker))
        c.set_option("sweep_ahead", ahead)
        c.enable_frontal(8)
        c.set_solver(preconditioner=2, rtol=1e-12, maxit=30, check_every=1)
        out = []
        for rep in range(3):                              # cold solves one after the other: events and streams are reused
            c.set_field("thickness", h * (1 + 0.01 * rep))
            it, _ = c.solve_state(True)
            out.append((it, c.get_state()))
        c.set_field("F_solid", np.zeros((m.nn, 3)))       # zero load: factorisation, no iteration
        it0, _ = c.solve_state(True)
        assert it0 == 0 and np.all(c.get_state() == 0.0)
        res[ahead] = out
        c.close()
    for ahead in (1, 2, 3, 50):
        for (it_a, w_a), (it_0, w_0) in zip(res[ahead], res[0]):
            assert it_a == it_0
            assert np.abs(w_a - w_0).max() < 1e-10 * np.abs(w_0).max(), ahead

#!/usr/bin/env python3
"""Benchmark of the RM-shell forward + adjoint hot path on MI355X.

Contract:  python bench.py --gpus N --steps K --warmup W   (N>1 under torch.distributed.run)
prints ONE JSON line on rank 0.

A *step* is one pass of the hot path over one synthetic design: a cold-start forward solve
(operator set-up + PCG solve of K w = F, "assembly+solve") followed by the adjoint gradient
d compliance / d thickness (dJ/dw, one more solve, (dR/dh)^T lambda, dJ/dh).  Inputs are resident
in HBM before the timed region; nothing is cached across steps (zero initial guess, thickness
re-uploaded outside the timed region only when it changes -- it does not).

``value`` = DOF/s of the forward part = ndof * K * N / (time spent in forward solves), the
BASELINE.json metric "DOF/s (assembly+solve)"; ``adjoint_ms`` is the adjoint-gradient wall-clock
per step, the second half of that metric.  ``ms_per_step`` covers the whole step.
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
# fp64 peak: the guide lists FP32 vector/matrix = 157.3 TFLOP/s (64 FLOP/clk/SIMD); CDNA4 runs fp64 FMA and
# v_mfma_f64_16x16x4_f64 at half that rate -> 78.6 TFLOP/s (MI355X datasheet: FP64 vector = FP64 matrix = 78.6)
FP64_PEAK_TFLOPS = 78.6
# fp64 operations k_apply4 executes per quadrature point of a quad cell without mesh motion, counted from the kernel's loop body
# (femo_alpha_amd/csrc/shell_device.h; an FMA = 2, a divide / square root = 1): geometry + derivative of the normal 176,
# three field interpolations 24, constitutive coefficients 32, strains_q (two 2x2 derivative maps of 13 nodes, six reduced
# 3-vectors, nine strains) 390, stress_of 21, strains_T_q 382  ->  1025; per cell 16 points + the DPP quad reduction and the
# p.Ap partial (390).  k_gather_sum adds 2 flops per (cell, local DOF).
APPLY_FLOPS_PER_QP = 1025.0
APPLY_FLOPS_PER_CELL_EXTRA = 390.0 + 2.0 * 39.0


def apply_flops(nel, nq=16):
    return nel * (nq * APPLY_FLOPS_PER_QP + APPLY_FLOPS_PER_CELL_EXTRA)


def spmv_roofline(apply_ms, ndof, nel, traffic, where="", nq=16):
    """The north star's SpMV (k_apply4 + k_gather_sum) against BOTH roofs: HBM on the algorithmic bytes of SURVEY.md section 8d
    (B_spmv,ebe = 16 B/DOF + 340 B/cell) and the fp64 vector ALU on the counted flops.  Counter traffic equals the algorithmic
    bytes, so the HBM fraction is not what binds; the VALU fraction is the achieved share of the binding roof."""
    alg_bytes = 16.0 * ndof + 340.0 * nel
    gbs = alg_bytes / (apply_ms * 1e-3) / 1e9
    fl = apply_flops(nel, nq)
    tf = fl / (apply_ms * 1e-3) / 1e12
    return {"bound": "hbm", "kernel": "k_apply4 (matrix-free CG2xCG1 shell operator)" + where,
            "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": apply_ms,
            "algorithmic_flops_per_launch": fl, "achieved_fp64_valu_TFLOPs": tf, "frac_fp64_valu": tf / FP64_PEAK_TFLOPS,
            "binding": "fp64 vector ALU latency at two waves per SIMD (242 VGPRs); neither roof is reached"}


def trailing_roofline(prof, traffic, traffic_by_class=None):
    """The dominant kernel of the step: the rank-k updates of the multifrontal Cholesky, k_trailing_mfma / k_trailing_fine (fp64 MFMA).
    Flops and compulsory bytes are summed over the launches from each launch's own K / column ranges (femo_hip.hip, count_trailing):
    lower triangles only, C read + written once, the factor rows of the K panel read once.  One kernel, two roofs: a launch whose
    algorithmic flops / compulsory bytes lie above the ridge of the chip (78.6 TFLOP/s / 8 TB/s = 9.8 flop per byte) can be bounded by
    the matrix cores, one below it is bounded by HBM whatever the kernel does (the small-K updates of the lower tree levels).  The
    instrumented factorisation classifies every launch.  Returns (all launches, the class that takes more time, the other class):
    the top-level ``roofline`` of the JSON line is ALL launches against the MFMA peak (one definition for every round), with the two
    classes inside it (``by_binding_roof``) and beside it (``roofline_trailing_mfma_class`` / ``roofline_trailing_hbm_class``)."""
    tr = prof["trailing"]
    tf_all = prof["trailing_flops"] / (tr["ms"] * 1e-3) / 1e12
    both = {}
    for key, bound in (("trailing_mfma_bound", "mfma"), ("trailing_hbm_bound", "hbm")):
        p = prof[key]
        if p["launches"] <= 0 or p["ms"] <= 0:
            continue
        tf = p["flops"] / (p["ms"] * 1e-3) / 1e12
        gbs = p["bytes"] / (p["ms"] * 1e-3) / 1e9
        o = {"bound": bound, "kernel": "k_trailing_mfma / k_trailing_fine (fp64 rank-k updates of the multifrontal Cholesky): the launches "
                                       + ("above" if bound == "mfma" else "below") + " the ridge of 9.8 flop per compulsory byte",
             "achieved": tf if bound == "mfma" else gbs, "peak": FP64_PEAK_TFLOPS if bound == "mfma" else HBM_PEAK_GBS,
             "unit": "TFLOP/s" if bound == "mfma" else "GB/s",
             "frac": tf / FP64_PEAK_TFLOPS if bound == "mfma" else gbs / HBM_PEAK_GBS, "traffic": None,
             "algorithmic_flops_per_launch": p["flops"] / p["launches"], "algorithmic_bytes_per_launch": p["bytes"] / p["launches"],
             "avg_launch_ms": p["ms"] / p["launches"], "launches_per_factorisation": p["launches"], "ms_per_factorisation": p["ms"],
             "achieved_TFLOPs": tf, "achieved_compulsory_GBs": gbs}
        both[bound] = o
    allk = {"bound": "mfma", "kernel": "k_trailing_mfma / k_trailing_fine / k_schur_strip (fp64 rank-k updates of the multifrontal Cholesky), ALL launches",
            "achieved": tf_all, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf_all / FP64_PEAK_TFLOPS, "traffic": traffic,
            "algorithmic_flops_per_launch": prof["trailing_flops"] / tr["launches"],
            "algorithmic_bytes_per_launch": prof["trailing_bytes"] / tr["launches"],
            "avg_launch_ms": tr["ms"] / tr["launches"], "launches_per_factorisation": tr["launches"], "ms_per_factorisation": tr["ms"],
            "what": "all rank-k launches of one factorisation against the fp64 MFMA peak, whichever roof binds the single launch: the same "
                    "definition every round (rounds 1-3: `roofline`; round 4: `roofline_trailing_all_launches`).  `by_binding_roof` splits the "
                    "launches by the roof that binds them (flops / compulsory bytes above or below the ridge of 9.8)"}
    for o in both.values():                 # counter bytes per launch of each class (scripts/r4_pmc_levels.py), when the committed passes are current
        o["traffic"] = (traffic_by_class or {}).get(o["bound"])
    allk["by_binding_roof"] = both
    main = max(both.values(), key=lambda o: o["ms_per_factorisation"]) if both else None
    other = [o for o in both.values() if o is not main]
    return allk, main, (other[0] if other else None)


def pmc_traffic(workload):
    """HBM bytes per launch from the committed rocprofv3 counter passes (scripts/aggregate_pmc.py; separate FETCH_SIZE and
    WRITE_SIZE runs, gfx950 correction 2*FETCH+WRITE) -- counters cannot be read from inside bench.py.  Accepted only when the
    file carries the digest of the library sources that produced it: after a kernel change the committed bytes are stale."""
    from femo_alpha_amd import _build
    pmc = os.path.join(ROOT, "profiles", f"pmc_{workload}.json")
    if not os.path.exists(pmc):
        return None, None, {}
    pj = json.load(open(pmc))
    if pj.get("source_digest") != _build.source_digest():
        print(f"warning: {os.path.relpath(pmc, ROOT)} was measured on other kernel sources (digest mismatch): roofline.traffic = null; "
              "re-run scripts/r5_rocprof.sh", file=sys.stderr)
        return None, None, {}
    # per-class figures of the rank-k updates (scripts/r4_pmc_levels.py): launches above / below the ridge of the chip
    by_class = {"mfma": pj.get("trailing_mfma_bound_hbm_bytes_per_launch"), "hbm": pj.get("trailing_hbm_bound_hbm_bytes_per_launch")}
    return pj.get("apply_hbm_bytes_per_launch"), pj.get("trailing_hbm_bytes_per_launch"), by_class


def make_workload(name, renumber=True, timings=None, tri=None):
    """(mesh, fields, clamp marker, description).  ``renumber``: cells and vertices reordered for locality at input
    (ShellMesh.renumbered), as dolfinx reorders every mesh it reads; False keeps the generator's numbering -- for the wing
    skin a random shuffle of cells and vertices.  ``timings`` receives mesh_s / renumber_s.
    ``tri``: the triangulation of the unstructured skins' point set (femo_alpha_amd.mesh.skin_triangulation) when it must not be
    recomputed -- the goldens of uskin1m / uquad1m carry theirs, so that their tests do not depend on the installed qhull."""
    from femo_alpha_amd.mesh import plate_mesh, wing_skin_mesh
    t_start = time.perf_counter()
    if name == "plate250k":      # BASELINE.json configs[1]: flat plate, 250k DOF, thickness design variable
        m = plate_mesh(2.0, 10.0, 58, 290)
        rng = np.random.default_rng(0)
        fields = dict(thickness=0.1 * (1 + 0.2 * rng.uniform(-1, 1, m.nn)), E=[1e8], nu=[0.3], density=[10.0],
                      F_solid=np.tile([0.0, 0.0, 5.0], (m.nn, 1)))
        marker = lambda x: np.less(x[0], 3e-16)
        desc = "flat plate 2x10, 58x290 CG2xCG1 quads, 255438 DOF, nodal thickness 0.1*(1+0.2U), clamped x=0 (penalty 1e15)"
    elif name == "wing1m":       # BASELINE.json configs[2]
        m = wing_skin_mesh(116, 580)
        fields = dict(thickness=[1.27e-3], E=[73.1e9], nu=[0.33], density=[2780.0],
                      F_solid=np.tile([0.0, 0.0, -2780.0 * 1.27e-3 * 9.81], (m.nn, 1)))
        marker = lambda x: np.less(x[1], 1e-9)
        desc = "synthetic wing skin 116x580 quads (cambered, tapered, twisted, jittered, renumbered), 1015470 DOF"
    elif re.fullmatch(r"wing[1-9][0-9]*m", name):   # beyond BASELINE: wing<k>m = k times the span on one GPU (k = 4 .. 48; fronts ~4.7 GB per 1M DOF)
        mult = int(name[4:-1])
        m = wing_skin_mesh(116, 580 * mult, span=6.0 * mult)
        fields = dict(thickness=[1.27e-3], E=[73.1e9], nu=[0.33], density=[2780.0],
                      F_solid=np.tile([0.0, 0.0, -2780.0 * 1.27e-3 * 9.81], (m.nn, 1)))
        marker = lambda x: np.less(x[1], 1e-9)
        desc = f"synthetic wing skin 116x{580 * mult} quads, {m.ndof} DOF"
    elif name == "wing1m_tri":   # SURVEY.md section 8d, config 3, triangle variant: 183 x 365 x 2 = 133 590 triangles, 1 006 863 DOF
        from femo_alpha_amd.mesh import quads_to_triangles
        m = quads_to_triangles(wing_skin_mesh(183, 365))
        fields = dict(thickness=[1.27e-3], E=[73.1e9], nu=[0.33], density=[2780.0],
                      F_solid=np.tile([0.0, 0.0, -2780.0 * 1.27e-3 * 9.81], (m.nn, 1)))
        marker = lambda x: np.less(x[1], 1e-9)
        desc = f"synthetic wing skin 183x365 quads split into {m.nel} CG2xCG1 triangles (cambered, tapered, twisted, jittered, renumbered), {m.ndof} DOF"
    elif name == "uskin1m":      # the config-3 surface with an UNSTRUCTURED triangulation (Delaunay of jittered points): same vertices, same DOF count
        from femo_alpha_amd.mesh import unstructured_skin_mesh
        m = unstructured_skin_mesh(116, 580, tri=tri)
        fields = dict(thickness=[1.27e-3], E=[73.1e9], nu=[0.33], density=[2780.0],
                      F_solid=np.tile([0.0, 0.0, -2780.0 * 1.27e-3 * 9.81], (m.nn, 1)))
        marker = lambda x: np.less(x[1], 1e-9)
        desc = f"synthetic wing skin, unstructured: {m.nel} CG2xCG1 triangles (Delaunay of 117 x 581 jittered points, renumbered), {m.ndof} DOF"
    elif name == "uquad1m":      # an UNSTRUCTURED ALL-QUADRILATERAL skin of the config-3 surface and size: what the reference's real wings are
        from femo_alpha_amd.mesh import unstructured_quad_skin_mesh
        m = unstructured_quad_skin_mesh(47, 239, tri=tri)
        fields = dict(thickness=[1.27e-3], E=[73.1e9], nu=[0.33], density=[2780.0],
                      F_solid=np.tile([0.0, 0.0, -2780.0 * 1.27e-3 * 9.81], (m.nn, 1)))
        marker = lambda x: np.less(x[1], 1e-9)
        desc = (f"synthetic wing skin, unstructured quadrilaterals: {m.nel} CG2xCG1 quads (Delaunay of 48 x 240 jittered points, every triangle "
                f"cut into three kites; vertex valences 3..9, shuffled numbering), {m.ndof} DOF")
    elif name == "plate8k":      # BASELINE.json configs[0] (plumbing size)
        m = plate_mesh(2.0, 10.0, 10, 50)
        fields = dict(thickness=[0.1], E=[1e8], nu=[0.3], density=[10.0], F_solid=np.tile([0.0, 0.0, 5.0], (m.nn, 1)))
        marker = lambda x: np.less(x[0], 3e-16)
        desc = "flat plate 2x10, 10x50 quads, 8046 DOF"
    else:
        raise SystemExit(f"unknown workload {name}")
    t_mesh = time.perf_counter()
    if renumber:
        # input stage, outside the timed region: cells and vertices reordered for locality, as dolfinx reorders every
        # mesh it reads (the reference maps back through original_cell_index); per-vertex inputs follow the permutation
        m, vperm, _ = m.renumbered()
        for k, v in fields.items():
            v = np.asarray(v)
            if v.ndim == 2 and v.shape[0] == vperm.size or v.ndim == 1 and v.size == vperm.size:
                fields[k] = v[vperm]
        desc += "; solver-side numbering: Morton order of the cells (ShellMesh.renumbered)"
    if timings is not None:
        timings["mesh_s"] = t_mesh - t_start
        timings["renumber_s"] = time.perf_counter() - t_mesh
    return m, fields, marker, desc


def cpu_baseline(m, fields, marker, leaf, budget_s=25.0, workload="wing1m", nquad=None):
    """The "reference CPU path" timed on this box's host cores (rank 0, N = 1 only): kind "port" -- the reference itself
    needs FEniCSx/PETSc and cannot run here, so this is oracle/cpu_baseline.py, the repository's float64 restatement of
    its algorithm: C++/OpenMP element assembly + a multifrontal Cholesky on dense fronts with LAPACK/BLAS (what MUMPS is
    algorithmically), on the SAME mesh, fields and quadrature rule as the GPU run.  Bounded sample: the best-effort forward
    solve (1 assembly + 1 factorisation + 3 solves), median of up to 3 runs inside ``budget_s``, at 16 threads (the CPU share of a
    one-GPU box) and, where the job is GRANTED more, at every core (at most 64: scipy's OpenBLAS is built for 64 threads) -- the
    control group's CPU quota counts, not the affinity mask: a one-GPU box lists 256 hardware threads and grants 16, and round 4's
    "all cores" leg of 64 threads was throttled by the scheduler into being 5 x slower than 16 (``cpu_quota_cores``,
    ``throttled_periods_during_cpu_legs`` in the output; profiles/r5_cpu_levels.txt).  ``value`` is the FASTEST leg; plus one
    single-core run if the budget allows.  The full protocol of BASELINE.md section 3 (both core counts, "as the reference runs it" with SuperLU,
    5 repeats) is scripts/cpu_baseline_full.py -> profiles/."""
    from femo_alpha_amd.solver.symbolic import build_plan
    from oracle import cpu_baseline as cb
    from oracle.rm_shell_oracle import ShellOracle
    t_begin = time.perf_counter()
    throttled0 = cb.cpu_throttled_periods()
    o = ShellOracle(m, nquad=nquad, penalty_facets=m.penalty_facets(marker))
    o.set_fields(h=fields["thickness"], E=fields["E"], nu=fields["nu"], rho=fields["density"], f=fields["F_solid"])
    plan = build_plan(m, leaf)
    cs = cb.CpuShell(o)
    share, every = cb.host_cores(), cb.host_cores(cap=64)
    b = cs.load_vector(every)
    mf = cb.CpuMultifrontal(cs, plan, share)          # one set of fronts (15 GB at 1 M DOF) for every thread count

    def forward(cores):
        mf.nthreads = cores
        t0 = time.perf_counter()
        asm, fac = mf.factorize()
        w = mf.solve(b)
        for _ in range(2):                           # two refinement steps on the true residual: three solves, all executed
            w = w + mf.solve(b - cs.apply_K(w, cores))
        t1 = time.perf_counter()
        return t1 - t0, asm, fac, w

    def adjoint(w, cores):
        # best effort, like the forward leg: quadrature sweeps in C++/OpenMP (no assembled dR/dh), the factor reused, the
        # triangular sweeps level-parallel; one refinement step on the true residual, as the GPU path does (2 PCG iterations)
        t0 = time.perf_counter()
        rhs, dJdh = cs.dcompliance(w, cores)
        lam = mf.solve(rhs)
        lam = lam + mf.solve(rhs - cs.apply_K(lam, cores))
        g = cs.drdfield_T("h", w, lam, cores, scale=-1.0, out=dJdh)
        return time.perf_counter() - t0, g

    def leg(cores, budget):
        t_leg = time.perf_counter()
        thr0 = cb.cpu_throttled_periods()
        forward(cores)                                                          # warm-up (page faults of the fronts)
        runs = []
        # a leg the scheduler throttled (the control group's counter moved) gets two more runs, and the fastest run is reported beside the
        # median: a throttled CPU leg flatters the GPU / CPU ratio
        while len(runs) < 3 or (len(runs) < 5 and (cb.cpu_throttled_periods() or 0) > (thr0 or 0)):
            if runs and time.perf_counter() - t_leg + runs[-1][0] >= budget:
                break
            runs.append(forward(cores))
        runs.sort(key=lambda r: r[0])
        tot, asm, fac, w = runs[len(runs) // 2]
        adjoint(w, cores)
        adj = float(np.median([adjoint(w, cores)[0] for _ in range(3)]))
        thr1 = cb.cpu_throttled_periods()
        return dict(value=m.ndof / tot, cores=cores, forward_s=tot, forward_s_fastest_run=runs[0][0], value_fastest_run=m.ndof / runs[0][0],
                    assemble_s=asm, factor_s=fac, adjoint_ms=adj * 1e3, runs=len(runs),
                    throttled_periods=None if thr0 is None or thr1 is None else thr1 - thr0)

    def as_the_reference_runs_it(cores):
        """EXECUTED, once, in this run: the reference's Newton loop never meets its tolerances and runs 3 iterations
        = 3 x (Jacobian assembly into CSR + direct factorisation + solve) + 4 residual evaluations (utils_dolfinx.py:438-468), and on the
        first evaluation assembles 7 derivative matrices and factorises once more for the adjoint (state_operation.py:260-296).  The direct
        solver is the multifrontal Cholesky (half the flops of the reference's LU: this favours the CPU)."""
        mf.nthreads = cores
        t0 = time.perf_counter()
        w = np.zeros(m.ndof)
        r = -b
        for _ in range(3):
            K = cs.assemble_K(cores)
            mf.factorize()
            w = w - mf.solve(r)
            r = K @ w - b
        r = K @ w - b                                                           # the 4th residual (convergence check of the last iteration)
        t_fwd = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(3):                                                      # dR/du, A (two Jacobian assemblies) + dR/df and dR/duhat counted as one
            K = cs.assemble_K(cores)
        for name in ("h", "E", "nu"):
            cs.assemble_drdfield(name, w, cores)
        mf.factorize()
        t_adj = time.perf_counter() - t0
        del K
        return dict(value=m.ndof / t_fwd, cores=cores, forward_s=t_fwd, adjoint_setup_s=t_adj, executed=True, runs=1,
                    what="3 x (CSR assembly + front assembly + multifrontal Cholesky + solve) + 4 residuals; adjoint set-up: 7 matrix assemblies "
                         "(3 Jacobian-sized, dR/dh, dR/dE, dR/dnu) + 1 factorisation")

    legs = [leg(share, budget_s)]
    if every > share:
        legs.append(leg(every, budget_s))
    best = max(legs, key=lambda l: l["value"])
    out = dict(value=best["value"], unit="DOF/s", cores=best["cores"], kind="port", cpu_model=cb.cpu_model_name(),
               sample=f"the workload itself ({m.ndof} DOF, {o.nquad} x {o.nquad} Gauss points): C++/OpenMP front assembly {best['assemble_s']:.2f} s + "
                      f"multifrontal Cholesky (LAPACK/BLAS) {best['factor_s']:.2f} s + 3 level-parallel triangular solves with 2 matrix-free "
                      f"residuals (all executed) = {best['forward_s']:.2f} s, median of {best['runs']} after 1 warm-up, {best['cores']} threads "
                      f"(the faster of {[l['cores'] for l in legs]} threads); adjoint gradient with the same factor (C++/OpenMP quadrature of "
                      f"dJ/du, dJ/dh and (dR/dh)^T lambda, 2 solves + 1 residual) {best['adjoint_ms'] / 1e3:.3f} s, median of 3",
               forward_s=best["forward_s"], adjoint_ms=best["adjoint_ms"], **cpu_allowance(cb, throttled0))
    for l in legs:
        out["gpu_share_16_threads" if l["cores"] == share else "all_cores"] = l
    out["value_fastest_run"] = max(l["value_fastest_run"] for l in legs)
    if time.perf_counter() - t_begin < 2 * budget_s + 20.0:
        out["as_the_reference_runs_it"] = as_the_reference_runs_it(best["cores"])
    if time.perf_counter() - t_begin + 1.2 * share * legs[0]["forward_s"] < 2 * budget_s + 30.0:    # one single-core run, if it fits (same, already touched, fronts)
        t1, a1, f1, _ = forward(1)
        out["single_core"] = dict(value=m.ndof / t1, forward_s=t1, assemble_s=a1, factor_s=f1, cores=1)
    for full in (os.path.join("profiles", f"r4_cpu_baseline_{workload}.json"), os.path.join("profiles", f"r2_cpu_baseline_{workload}.json")):
        if os.path.exists(os.path.join(ROOT, full)):
            out["full_protocol"] = full
            break
    return out


def cpu_allowance(cb, throttled0):
    """What "all cores" means for this job: the hardware threads the affinity mask lists, the CPUs the control group grants time on
    (a one-GPU box of this pool: 256 listed, 16 granted -- threads beyond the quota are throttled, so the quota IS every core this
    job has), and how many scheduling periods the CPU legs were throttled in (0 when the thread count respects the quota)."""
    t1 = cb.cpu_throttled_periods()
    return dict(visible_hardware_threads=os.cpu_count(), cpu_quota_cores=cb.cpu_quota_cores(),
                throttled_periods_during_cpu_legs=None if t1 is None or throttled0 is None else t1 - throttled0)


def dynamic_case(nx=82, ny=410, nsteps=100):
    """BASELINE config 5: plate 2 x 10, 82 x 410 quads (508 734 DOF), E = 1e8, nu = 0.3, rho = 10, h = 0.1,
    quadrature degree 3 for the strain energies, strong clamp at x = 0, 100 midpoint / Newmark steps over T2 = 2.86 with the
    1-cosine gust f_z(t) = 0.1 V_p (1 - cos(2 pi (t - T0) / T1)), V_p = 50, T0 = 0.02, T1 = 0.12
    (ex_simple_dynamic_shell_opt.py:45-95)."""
    from femo_alpha_amd.mesh import plate_mesh
    mesh = plate_mesh(2.0, 10.0, nx, ny)
    dt = 2.86 / nsteps
    tt = np.arange(nsteps + 1) * dt
    fz = np.where((tt >= 0.02) & (tt <= 0.14), 0.1 * 50 * (1 - np.cos(2 * np.pi * (tt - 0.02) / 0.12)), 0.0)
    F = np.zeros((nsteps + 1, mesh.nn, 3)); F[:, :, 2] = fz[:, None]
    return mesh, dt, F.reshape(nsteps + 1, -1)


def cpu_baseline_dynamic(mesh, dt, F, leaf, nsteps):
    """CPU column of config 5, kind "port": oracle/cpu_baseline.py::dynamic_march -- the oracle's dynamic_history with the C++/OpenMP
    element kernels and the LAPACK/BLAS multifrontal Cholesky, one direct solve per time step like the reference.  Bounded sample:
    3 time steps with the operator re-assembled and re-factorised before every step (the reference's procedure and BASELINE's
    wording) and 8 steps on one factorisation, at 16 threads and at every core (at most 64); steps cost the same along the march."""
    from femo_alpha_amd.solver.symbolic import build_plan
    from oracle import cpu_baseline as cb
    from oracle.rm_shell_oracle import ShellOracle
    throttled0 = cb.cpu_throttled_periods()
    strong = mesh.locate_dofs_geometrical(lambda x: np.isclose(x[0], 0.0, atol=1e-6))
    o = ShellOracle(mesh, strong_dofs=strong, nred=2)
    o.set_fields(h=0.1, E=1e8, nu=0.3, rho=10.0)
    cs = cb.CpuShell(o)
    share, every = cb.host_cores(), cb.host_cores(cap=64)
    mf = cb.CpuMultifrontal(cs, build_plan(mesh, leaf), share)
    legs = []
    for cores in ([share] if every == share else [share, every]):
        cb.dynamic_march(cs, mf, F, dt, 1, cores, True)                       # warm-up: page faults of the fronts
        t0 = time.perf_counter()
        _, tr = cb.dynamic_march(cs, mf, F, dt, 3, cores, True)
        t_re = (time.perf_counter() - t0) / 3
        t0 = time.perf_counter()
        _, to = cb.dynamic_march(cs, mf, F, dt, 8, cores, False)
        t_once = (time.perf_counter() - t0 - to["assemble"] - to["factor"]) / 8
        legs.append(dict(cores=cores, s_per_time_step_reassembled=t_re, s_per_time_step_factor_once=t_once,
                         value=mesh.ndof / t_re, phases_reassembled_s={k: v / 3 for k, v in tr.items()}))
    best = max(legs, key=lambda l: l["value"])
    out = dict(value=best["value"], unit="DOF/s", cores=best["cores"], kind="port", cpu_model=cb.cpu_model_name(),
               sample=f"3 of the {nsteps} time steps of the workload itself ({mesh.ndof} DOF) with the step operator 2/dt^2 M + K/2 re-assembled "
                      f"and re-factorised before every step (C++/OpenMP front assembly + LAPACK/BLAS multifrontal Cholesky + right-hand side + "
                      f"one direct solve = {best['s_per_time_step_reassembled']:.3f} s per time step), and 8 steps on one factorisation "
                      f"({best['s_per_time_step_factor_once']:.3f} s per time step); {best['cores']} threads (the faster of "
                      f"{[l['cores'] for l in legs]})",
               time_steps_per_s=1.0 / best["s_per_time_step_reassembled"],
               time_steps_per_s_factor_once=1.0 / best["s_per_time_step_factor_once"], **cpu_allowance(cb, throttled0))
    for l in legs:
        out["gpu_share_16_threads" if l["cores"] == share else "all_cores"] = l
    return out


def golden_distance_dynamic(W, mesh):
    """Relative distance of a config-5 march (W: time levels x ndof) from tests/golden/config5_plate500k_dynamic.npz -- the CPU
    restatement's march with every step's solve refined to 1e-14: the tip deflection at every time level and the samples of the last
    state.  None when the golden is not there or belongs to another case."""
    path = os.path.join(ROOT, "tests", "golden", "config5_plate500k_dynamic.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    if int(g["ndof"]) != mesh.ndof or int(g["nsteps"]) + 1 != W.shape[0]:
        return None
    hist, ref = W[:, 3 * int(g["tip_vertex"]) + 2], g["tip_history"]
    return {"tip_history": float(np.abs(hist - ref).max() / np.abs(ref).max()),
            "last_state_samples": float(np.abs(W[-1][g["sample_index"]] - g["w_last_sample"]).max() / float(g["w_last_maxabs"]))}


def main_dynamic(args, torch):
    """``--workload plate500k_dynamic`` = BASELINE config 5.  A bench step is ONE MARCH of 100 time steps from zero initial
    conditions with the step operator re-assembled and re-factorised before every time step -- "re-assembly per step", as BASELINE
    words it and as the reference runs it (solveNonlinear_mod, nonlinear_utils.py:210-233 from plate_sim.py:281-361).

    ``value`` is measured AT THE PARITY SETTING: the Krylov tolerance under which tests/test_gpu_goldens.py::
    test_config5_march_against_the_full_size_golden passes 1e-8 (rtol 1e-13: two applications of the factor per time step, the
    second one a refinement step on the matrix-free residual).  The product default (PlateSim rtol 1e-8: ONE application per time
    step, what the reference's single Newton iteration with LU is) rides along as ``product_default_one_application`` with its
    measured distance from the golden; the operator does not change along a march, so the product factorises once per thickness:
    ``factor_once`` at both settings.  value = ndof x time steps / time."""
    from femo_alpha_amd.dynamic_rm_shell.plate_sim import PlateSim
    nsteps = 100
    steps = 3 if args.steps is None else args.steps
    warmup = 1 if args.warmup is None else args.warmup
    RTOL_PARITY, RTOL_PRODUCT = 1e-13, 1e-8
    t0 = time.perf_counter()
    mesh, dt, F = dynamic_case(nsteps=nsteps)
    args.leaf = mesh.recommended_leaf_size() if args.leaf is None else args.leaf
    ps = PlateSim(mesh, 1e8, 0.3, 10.0, dt, nsteps, quad_deg=3, leaf_size=args.leaf, rtol=RTOL_PARITY)
    thickness = np.full(mesh.nn, 0.1)
    ps.update_t(thickness)
    ps.update_f_history(F)
    ctx = ps.ctx
    ps._newmark()
    ctx.newmark_set_forces(ps._force_history())               # inputs resident in HBM before the timed region
    ctx.newmark_set_constant_load(None)
    setup_s = time.perf_counter() - t0

    def march(reassemble, rtol):
        ctx.set_solver(preconditioner=2, rtol=rtol, maxit=50, check_every=1)
        ps.update_t(thickness)                                # a new design: the factorisation is stale
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        info = ctx.newmark_march(nsteps, reassemble)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, info

    for _ in range(warmup):
        march(True, RTOL_PARITY)
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    rows = [march(True, RTOL_PARITY) for _ in range(steps)]
    torch.cuda.synchronize()
    t_total = time.perf_counter() - t_start
    W = ctx.newmark_history(0)
    dist_parity = golden_distance_dynamic(W, mesh)
    tip = float(np.abs(W[-1, 2:mesh.ndof_u:3]).max())
    its = [i for i, _ in rows[-1][1]]
    march(False, RTOL_PARITY)
    t_once = float(np.median([march(False, RTOL_PARITY)[0] for _ in range(max(steps, 3))]))
    # the product default: one application of the factor per time step
    march(True, RTOL_PRODUCT)
    prod = [march(True, RTOL_PRODUCT) for _ in range(max(steps, 3))]
    t_prod = float(np.median([r[0] for r in prod]))
    dist_prod = golden_distance_dynamic(ctx.newmark_history(0), mesh)
    march(False, RTOL_PRODUCT)
    t_prod_once = float(np.median([march(False, RTOL_PRODUCT)[0] for _ in range(max(steps, 3))]))
    # dominant kernel of the march as worded: the rank-k updates of the per-step factorisation (one instrumented factorisation of
    # the step operator, HIP event pairs on the context's stream); of the factor-once march: the triangular sweeps (HBM)
    prof = ctx.factorize_profile()
    roof, roof_main, roof_other = trailing_roofline(prof, None)
    sw = np.min([ctx.sweep_profile() for _ in range(3)], axis=0)
    fac_bytes = float(np.sum(ctx.plan.nf.astype(np.float64) * ctx.plan.npiv) * 8)
    out = {
        "metric": "DOF/s (assembly+solve) of the transient RM-shell step: DOF x time steps per second, operator re-assembled and "
                  "re-factorised before every time step (BASELINE config 5 as worded), at the solver setting whose history matches the "
                  "full-size golden to 1e-8",
        "value": mesh.ndof * nsteps * steps / t_total, "unit": "DOF/s", "n_gpus": 1, "steps": steps, "warmup": warmup,
        "ms_per_step": t_total / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "plate500k_dynamic: plate 2x10, 82x410 CG2xCG1 quads, 508734 DOF, 100 midpoint/Newmark steps (dt = 0.0286), "
                               "1-cosine gust, strong clamp at x = 0, strain quadrature degree 3; one bench step = one march of 100 time steps",
                   "ndof": mesh.ndof, "cells": mesh.nel, "time_steps_per_march": nsteps,
                   "time_steps_per_s": nsteps * steps / t_total, "ms_per_time_step": t_total / steps / nsteps * 1e3,
                   "solves_per_time_step": "PCG to rtol 1e-13 on the matrix-free residual, preconditioned by the exact factor of the step "
                                           "operator: two applications of the factor per time step (the parity setting of "
                                           "tests/test_gpu_goldens.py::test_config5_march_against_the_full_size_golden)",
                   "pcg_iterations_max": max(its), "rtol": RTOL_PARITY, "tip_deflection_last_level": tip,
                   "distance_from_golden": dist_parity, "parallelism": "single"},
        "factor_once": {"march_ms": t_once * 1e3, "time_steps_per_s": nsteps / t_once, "dof_steps_per_s": mesh.ndof * nsteps / t_once,
                        "what": "the same march (parity setting) with the step operator factorised once per thickness: the operator "
                                "does not change along the march"},
        "product_default_one_application": {
            "rtol": RTOL_PRODUCT, "pcg_iterations_max": max(i for i, _ in prod[-1][1]),
            "what": "PlateSim's default: PCG stops after the FIRST application of the exact factor -- one direct solve per time step, as the "
                    "reference's single Newton iteration with LU (nonlinear_utils.py:220-229)",
            "march_ms_reassembled": t_prod * 1e3, "time_steps_per_s_reassembled": nsteps / t_prod,
            "dof_steps_per_s_reassembled": mesh.ndof * nsteps / t_prod,
            "march_ms_factor_once": t_prod_once * 1e3, "time_steps_per_s_factor_once": nsteps / t_prod_once,
            "distance_from_golden": dist_prod},
        "roofline": roof,
        **{f"roofline_trailing_{o['bound']}_class": o for o in (roof_main, roof_other) if o is not None},
        "roofline_sweeps": {"bound": "hbm", "kernel": "triangular sweeps of one preconditioner application (the solve of a factor-once time step)",
                            "achieved": 2 * fac_bytes / (sw.sum() * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": 2 * fac_bytes / (sw.sum() * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                            "algorithmic_bytes_per_application": 2 * fac_bytes, "ms_per_application": float(sw.sum())},
        "factorisation_profile_ms": {k: v["ms"] for k, v in prof.items() if isinstance(v, dict)},
        "setup_s": {"mesh_context_symbolic_upload_s": setup_s},
    }
    if not args.no_cpu_baseline:
        ctx.close()
        out["cpu_baseline"] = cpu_baseline_dynamic(mesh, dt, F, args.leaf, nsteps)
    print(json.dumps(out))
    ctx.close()


def main_distributed(args, rank, local_rank, world, torch, dist):
    """N > 1.  ``--scaling strong`` (default; BASELINE config 4 as written): the SAME 1 015 470-DOF wing skin split into N
    element partitions (the N subtrees at depth log2 N of the nested-dissection tree), one per GPU.  ``--scaling weak``: N
    times the span, N times the cells.  Collectives over RCCL (backend nccl): one all-reduce of the replicated separator
    entries per operator / preconditioner application, one packed scalar all-reduce per dot, one all-gather of the
    subtree roots' Schur complements (packed lower triangles) per factorisation.

    The line's ``value`` is the scaling asked for; the OTHER scaling rides along as a sub-object (``weak`` beside a strong run,
    ``strong`` beside a weak one) from a shorter timed region, because the two say different things: 1 M DOF is a small problem for
    eight MI355X (the builder's own composition bounds strong scaling at 2.3 x, weak at 6.3 x; profiles/r4_amdahl_wing1m*.md) --
    no curve has been measured.

    ``--rehearsal-engine module:Class`` (tests only): the ranks run that engine in place of the HIP one, on the CPU over gloo --
    launcher, rendezvous, collectives and the JSON line are exercised without a GPU; the line says so and is not a measurement."""
    from femo_alpha_amd.mesh import wing_skin_mesh
    from femo_alpha_amd.parallel import Comm, DistributedShell
    if world & (world - 1):
        raise SystemExit("the element partition needs a power-of-two number of GPUs")
    if args.workload != "wing1m":
        raise SystemExit("the multi-GPU bench runs the wing-skin workload")
    rehearsal = args.rehearsal_engine is not None
    Engine = None
    if rehearsal:
        import importlib
        mod, cls = args.rehearsal_engine.split(":")
        Engine = getattr(importlib.import_module(mod), cls)
    nc = int(os.environ.get("FEMO_BENCH_NC", "116"))               # chordwise / spanwise cells (116 x 580 = the 1M-DOF config); rehearsals shrink them
    ns = int(os.environ.get("FEMO_BENCH_NS", "580"))
    marker = lambda x: np.less(x[1], 1e-9)
    comm = Comm(dist)
    shared_gpu = (not rehearsal) and torch.cuda.device_count() < world        # rehearsal: several ranks on one card
    dev = "cpu" if rehearsal else "cuda"
    sync = (lambda: None) if rehearsal else torch.cuda.synchronize

    def barrier():
        sync()
        dist.barrier()
        sync()

    def run_case(scaling, steps, warmup):
        mult = world if scaling == "weak" else 1
        m = wing_skin_mesh(nc, ns * mult, span=6.0 * mult * ns / 580.0).renumbered()[0]
        leaf = m.recommended_leaf_size() if args.leaf is None else args.leaf
        nq = m.recommended_nquad() if args.nquad is None else args.nquad          # the rule of the WHOLE mesh on every rank
        factory = (lambda sub, plan, info: Engine(sub, plan, info, nquad=nq)) if rehearsal else None
        ds = DistributedShell(m, comm, bc_marker=marker, leaf_size=leaf, device=0 if shared_gpu else local_rank, nquad=args.nquad,
                              engine_factory=factory)
        ds.rtol = args.rtol
        fields = dict(thickness=np.array([1.27e-3]), E=np.array([73.1e9]), nu=np.array([0.33]), density=np.array([2780.0]),
                      F_solid=np.tile([0.0, 0.0, -2780.0 * 1.27e-3 * 9.81], (m.nn, 1)))
        ds.set_fields(**fields)

        def step():
            ds.set_fields(thickness=fields["thickness"])         # new design: factorisation stale
            sync()
            t0 = time.perf_counter()
            it, rr = ds.solve_state()
            sync()
            t1 = time.perf_counter()
            g, it2, rr2 = ds.total_gradient("compliance", "thickness")
            sync()
            t2 = time.perf_counter()
            return (t1 - t0, t2 - t1, it, rr, it2, rr2)

        for _ in range(warmup):
            step()
        barrier()
        t_start = time.perf_counter()
        rows = [step() for _ in range(steps)]
        barrier()
        t_total = time.perf_counter() - t_start
        tt = torch.tensor([t_total, sum(r[0] for r in rows), sum(r[1] for r in rows)], device=dev, dtype=torch.float64)
        comm.allreduce_(tt, op="max")
        t_total, t_fwd, t_adj = tt.tolist()
        return dict(ds=ds, m=m, rows=rows, t_total=t_total, t_fwd=t_fwd, t_adj=t_adj, steps=steps, scaling=scaling)

    def figures(r):
        m, ds = r["m"], r["ds"]
        return {"value": m.ndof * r["steps"] / r["t_fwd"], "unit": "DOF/s", "scaling": r["scaling"], "steps": r["steps"],
                "ms_per_step": r["t_total"] / r["steps"] * 1e3, "forward_ms": r["t_fwd"] / r["steps"] * 1e3,
                "adjoint_ms": r["t_adj"] / r["steps"] * 1e3, "ndof": m.ndof, "cells": m.nel, "ndof_per_gpu": ds.sub.ndof,
                "replicated_separator_dofs": ds.info["n_top"], "pcg_iterations_forward": r["rows"][-1][2],
                "pcg_iterations_adjoint": r["rows"][-1][4]}

    main_run = run_case(args.scaling, args.steps, args.warmup)
    ds, m, rows = main_run["ds"], main_run["m"], main_run["rows"]
    prof = apply_ms = None
    if not rehearsal:
        # the same roofline object as the N = 1 line, measured on rank 0's partition: the rank-k updates of ITS factorisation
        # (its subtree + the replicated top of the tree; one instrumented factorisation, HIP event pairs on the context's stream --
        # every rank takes part, the Schur all-gather sits in the middle) and its element operator
        ctx0 = ds.eng.ctx
        ctx0.set_option("profile", 1)
        with ds.eng.on_stream():
            ds.factorize()
        ctx0.set_option("profile", 0)
        prof = ctx0.factorize_profile(run=False)
        apply_ms = ctx0.bench_kernel("apply", 50)
    other = "weak" if args.scaling == "strong" else "strong"
    other_fig = None
    if not args.no_other_scaling:
        del main_run["ds"]
        if not rehearsal:
            ds.eng.ctx.close()
        other_run = run_case(other, max(3, args.steps // 4), min(args.warmup, 2))
        other_fig = figures(other_run)
        if not rehearsal:
            other_run["ds"].eng.ctx.close()
    if rank == 0:
        t_total, t_fwd, t_adj = main_run["t_total"], main_run["t_fwd"], main_run["t_adj"]
        out = {
            "metric": "DOF/s (assembly+solve), forward solve of the RM shell; adjoint-gradient wallclock in adjoint_ms",
            "value": m.ndof * args.steps / t_fwd, "unit": "DOF/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": t_total / args.steps * 1e3, "forward_ms": t_fwd / args.steps * 1e3,
            "adjoint_ms": t_adj / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic" if not rehearsal else "synthetic; REHEARSAL on the CPU (stand-in engine over gloo): launcher coverage, not a measurement",
            "config": {"workload": (f"wing1m, strong scaling: the {nc}x{ns} wing skin ({m.ndof} DOF) in {world} element partitions of "
                                    f"{ds.sub.nel} cells" if args.scaling == "strong" else
                                    f"wing1m x{world}, weak scaling: synthetic wing skin {nc}x{ns * world} quads (span x{world}), {m.ndof} DOF, "
                                    f"one element partition of {ds.sub.nel} cells per GPU"),
                       "ndof": m.ndof, "cells": m.nel, "ndof_per_gpu": ds.sub.ndof, "gauss_points_per_direction": ds.nquad,
                       "replicated_separator_dofs": ds.info["n_top"],
                       "solver": "PCG, matrix-free element-by-element operator, multifrontal Cholesky preconditioner "
                                 "(subtree per GPU, replicated top of the tree)",
                       "rtol": args.rtol, "pcg_iterations_forward": rows[-1][2], "pcg_iterations_adjoint": rows[-1][4],
                       "relres_forward": rows[-1][3], "relres_adjoint": rows[-1][5],
                       "parallelism": f"element partition over {world} GPUs, RCCL all-reduce of separator DOFs"},
        }
        if other_fig is not None:
            out[other] = other_fig
        if prof is not None:
            out["roofline"] = dict(trailing_roofline(prof, None)[0], where=f"rank 0 of {world}: its subtree + the replicated top of the tree")
            out["roofline_spmv"] = spmv_roofline(apply_ms, ds.sub.ndof, ds.sub.nel, None, where=f", rank 0 of {world}")
            out["factorisation_profile_ms"] = {k: v["ms"] for k, v in prof.items() if isinstance(v, dict)}
        print(json.dumps(out))
    dist.destroy_process_group()


def visible_gpus():
    """GPUs this process could use, counted WITHOUT touching HIP or torch (on ROCm ``torch.cuda.device_count()`` falls back to
    hipGetDeviceCount, which initialises the runtime in the caller): the KFD topology nodes that have SIMDs, cut down by the
    *_VISIBLE_DEVICES variables.  None when the topology cannot be read."""
    import glob
    n = 0
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    for path in nodes:
        try:
            props = dict(line.split(None, 1) for line in open(path).read().splitlines() if " " in line)
        except OSError:
            return None
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(args):
    """``python bench.py --gpus N`` without a launcher: this process parses the arguments, never initialises the GPU (devices
    are counted from sysfs; no torch import, no HIP call) and starts the N ranks itself (``torch.distributed.run`` as a child
    process; ``--standalone`` lets the launcher pick a free rendezvous port itself, so there is no bind-then-close race);
    rank 0's JSON line goes straight to this process's stdout, the exit code is the children's.  Under torchrun (WORLD_SIZE
    set) bench.py is a rank and never comes here.  No process that has initialised the GPU is ever replaced by another."""
    import subprocess
    env = dict(os.environ)
    # early refusal only: the topology lists the HOST's devices, a container may be allowed fewer of them -- the ranks, which
    # initialise the GPU anyway, count what they can really use and pick the collective backend (main(): --share-gpu -> gloo)
    ndev = visible_gpus()
    if ndev is not None and ndev < args.gpus and not args.share_gpu and args.rehearsal_engine is None:
        raise SystemExit(f"--gpus {args.gpus} but {ndev} device(s) visible (a rehearsal on fewer cards: --share-gpu)")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", os.path.abspath(__file__), *sys.argv[1:]]
    res = subprocess.run(cmd, env=env)
    if res.returncode:
        raise SystemExit(res.returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 80: a timed region of ~2 s at 1 M DOF (plate500k_dynamic: 3 marches)")
    ap.add_argument("--warmup", type=int, default=None, help="default 3 (plate500k_dynamic: 1)")
    ap.add_argument("--workload", default=os.environ.get("FEMO_BENCH_WORKLOAD", "wing1m"))
    ap.add_argument("--rtol", type=float, default=1e-10)
    ap.add_argument("--solver", default="frontal", choices=["frontal", "jacobi"])
    ap.add_argument("--leaf", type=int, default=None, help="cells per leaf of the nested dissection (default: ShellMesh.recommended_leaf_size, 12 quadrilaterals / 24 triangles)")
    ap.add_argument("--nquad", type=int, default=None, help="n x n Gauss points per quadrilateral (2..5).  Default: what the mesh asks for "
                    "(ShellMesh.recommended_nquad) -- the reference integrates (nearly) exactly, scripts/ufl_degree_estimate.py; n = 4 is exact "
                    "on flat cells, on the warped wing skin n = 5 is within 1e-9 of the limit and n = 4 7.5e-8 away in the gradient")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--keep-numbering", action="store_true", help="run on the generator's (shuffled) numbering")
    ap.add_argument("--no-keep-numbering-leg", action="store_true", help="skip the extra forward solve on the shuffled numbering")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: the same 1M-DOF skin split into N element partitions (BASELINE config 4), or N times the span")
    ap.add_argument("--no-other-scaling", action="store_true", help="N > 1: skip the shorter run of the other scaling (the 'weak' / 'strong' sub-object)")
    ap.add_argument("--rehearsal-engine", default=None, metavar="MODULE:CLASS",
                    help="tests only: N > 1 ranks run this engine in place of the HIP one, on the CPU over gloo (launcher coverage; the "
                         "line says REHEARSAL and is not a measurement)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="rehearsal: allow --gpus N on a box with fewer cards (the ranks share them; collectives over gloo)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    # FEMO_BENCH_FORCE_DIST=1: take the N > 1 code path with ONE rank (RCCL wants a device per rank, a one-GPU box has one): the whole
    # distributed bench -- process group over nccl, the driver's collectives on device tensors, the max-over-ranks timing -- runs
    # through RCCL; a rehearsal of what the 8-GPU node executes, not a measurement
    force_dist = world == 1 and os.environ.get("FEMO_BENCH_FORCE_DIST") == "1"
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:                 # one rank, one process: the port is taken again a moment later by this process itself
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if args.rehearsal_engine is not None:
        if world < 2:
            raise SystemExit("--rehearsal-engine is for the N > 1 launcher path")
        import torch.distributed as dist
        dist.init_process_group(backend="gloo")
        args.steps = 2 if args.steps is None else args.steps
        args.warmup = 0 if args.warmup is None else args.warmup
        return main_distributed(args, rank, local_rank, world, torch, dist)
    if world > 1 or force_dist:
        import torch.distributed as dist
        ndev = torch.cuda.device_count()
        if ndev < world and not args.share_gpu and "FEMO_BENCH_BACKEND" not in os.environ and not force_dist:
            raise SystemExit(f"{world} ranks but {ndev} device(s) visible (a rehearsal on fewer cards: --share-gpu)")
        torch.cuda.set_device(local_rank if ndev > local_rank else 0)
        # RCCL wants one device per rank: ranks that share a card (rehearsals, --share-gpu) talk over gloo
        backend = os.environ.get("FEMO_BENCH_BACKEND", "gloo" if ndev < world else "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank if ndev > local_rank else 0))
        else:
            dist.init_process_group(backend=backend)
    if args.gpus != world and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    from femo_alpha_amd.backend import ShellContext
    if args.workload == "plate500k_dynamic":
        if world > 1:
            raise SystemExit("the transient workload runs on one GPU (BASELINE config 5)")
        return main_dynamic(args, torch)
    args.steps = 80 if args.steps is None else args.steps
    args.warmup = 3 if args.warmup is None else args.warmup
    if world > 1 or force_dist:
        return main_distributed(args, rank, local_rank, world, torch, dist)
    setup = {}
    m, fields, marker, desc = make_workload(args.workload, renumber=not args.keep_numbering, timings=setup)
    args.leaf = m.recommended_leaf_size() if args.leaf is None else args.leaf
    t0 = time.perf_counter()
    ctx = ShellContext(m, device=local_rank, nquad=args.nquad)
    nquad = ctx.nquad                                   # the rule in use: --nquad, or what the mesh asks for
    for k, v in fields.items():
        ctx.set_field(k, v)
    ctx.set_penalty_facets(m.penalty_facets(marker))
    setup["context_s"] = time.perf_counter() - t0
    if args.solver == "frontal":
        # symbolic analysis (nested dissection, fronts, index maps) and its upload: mesh only, once per mesh, outside the
        # timed region like MUMPS' analysis phase would be if the reference kept its KSP -- it does not (setUpKSP_MUMPS per
        # assemble_derivatives, state_operation.py:296), so the numbers are printed beside the headline
        t0 = time.perf_counter()
        ctx.enable_frontal(args.leaf)
        setup["symbolic_s"] = ctx.symbolic_s
        setup["plan_upload_s"] = time.perf_counter() - t0 - ctx.symbolic_s
        ctx.set_solver(preconditioner=2, rtol=args.rtol, maxit=50, check_every=1)
    else:
        ctx.set_solver(rtol=args.rtol, maxit=400000, check_every=200)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    thickness = ctx.get_field("thickness")

    def step():
        # a new design: the thickness upload marks the operator set-up (factorisation / diagonal) stale
        ctx.set_field("thickness", thickness)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        it, rr = ctx.solve_state(zero_guess=True)
        t1 = time.perf_counter()
        step.timing = ctx.last_timing()
        g, it2, rr2 = ctx.total_gradient("compliance", "thickness")
        t2 = time.perf_counter()
        return (t1 - t0, t2 - t1, it, rr, it2, rr2)

    for _ in range(args.warmup):
        step()
    barrier()
    t_start = time.perf_counter()
    rows = [step() for _ in range(args.steps)]
    barrier()
    t_total = time.perf_counter() - t_start
    t_fwd = sum(r[0] for r in rows)
    t_adj = sum(r[1] for r in rows)
    if dist is not None:
        tt = torch.tensor([t_total, t_fwd, t_adj], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        t_total, t_fwd, t_adj = tt.tolist()

    # the matrix-free element operator (the SpMV of the north star), HIP events around back-to-back launches
    apply_ms = ctx.bench_kernel("apply", 100)
    traffic, traffic_trailing, traffic_by_class = pmc_traffic(args.workload)
    roof_spmv = spmv_roofline(apply_ms, m.ndof, m.nel, traffic, nq=nquad ** 2)
    roof = roof_spmv
    prof = None
    if args.solver == "frontal":
        # dominant kernel of the frontal path: the trailing update of the partial Cholesky (fp64 rank-k updates);
        # one instrumented factorisation with a HIP event pair around every launch on the context's stream
        prof = ctx.factorize_profile()
        roof, roof_main, roof_other = trailing_roofline(prof, traffic_trailing, traffic_by_class)
        kernels = {}
        for cls, fk in (("trailing", "trailing"), ("panel_rows", "panel_rows"), ("panel_diag", "panel_diag")):
            ms = prof[cls]["ms"]
            kernels[cls] = {"ms": ms, "launches": prof[cls]["launches"], "gflop": prof[fk + "_flops"] / 1e9,
                            "tflops": prof[fk + "_flops"] / (ms * 1e-3) / 1e12 if ms > 0 else None,
                            "compulsory_GB": prof[fk + "_bytes"] / 1e9}

    # the adjoint gradient of FOUR outputs of the state in one grouped solve (femo_total_gradients: compliance, elastic energy and two
    # stress aggregates -- the reference registers these on disp_solid, rm_shell_model.py:221-253, and solves one adjoint per output):
    # the four right-hand sides share the triangular sweeps (csrc/sweeps_multi.h); beside it the sweeps alone for 1 / 2 / 4 vectors
    multi = None
    if args.solver == "frontal" and world == 1:
        ctx.solve_state(zero_guess=True)
        ctx.set_stress_params(m=1e-6, rho=6.0)
        names = ["compliance", "elastic_energy", "pnorm_stress", "tip_disp"]
        def timed(fn, reps=7):
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize(); t0 = time.perf_counter(); out_ = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            return float(np.median(ts[1:]) * 1e3), out_
        t4, (_, its4, _) = timed(lambda: ctx.total_gradients(names, "thickness"))
        t1, _ = timed(lambda: ctx.total_gradient("compliance", "thickness"))
        t4s, _ = timed(lambda: [ctx.total_gradient(nm, "thickness") for nm in names], reps=4)
        sw = {f"vectors_{nr}": ctx.bench_kernel(f"sweeps{nr}", 20) for nr in (1, 2, 4)}
        multi = {"adjoint_ms_4_outputs": t4, "adjoint_ms_1_output_same_clock": t1, "adjoint_ms_4_outputs_one_at_a_time": t4s,
                 "ratio_4_outputs_to_1": t4 / t1, "outputs": names, "pcg_iterations": [int(v) for v in its4],
                 "preconditioner_application_ms": sw,
                 "what": "host wall-clock around the C-ABI call (device-to-host copy of the gradients inside), median of 6"}

    # true residual of the last forward solve, || F - K w || / || F || evaluated by the matrix-free operator (the
    # relres_* figures of the Krylov loop are recurrence residuals)
    ctx.solve_state(zero_guess=True)
    r_true = ctx.residual()
    F_load = ctx.load_vector()
    true_relres = float(np.linalg.norm(r_true) / np.linalg.norm(F_load))

    # the same forward solve on the generator's shuffled numbering (config 3 asks for destroyed locality; the headline
    # renumbers at input like dolfinx does): one more context, outside the timed region
    keep = None
    if args.solver == "frontal" and args.workload.startswith("wing") and not args.keep_numbering and not args.no_keep_numbering_leg and world == 1:
        m2, fields2, marker2, _ = make_workload(args.workload, renumber=False)
        c2 = ShellContext(m2, device=local_rank, nquad=nquad)
        for k, v in fields2.items():
            c2.set_field(k, v)
        c2.set_penalty_facets(m2.penalty_facets(marker2))
        c2.enable_frontal(args.leaf)
        c2.set_solver(preconditioner=2, rtol=args.rtol, maxit=50, check_every=1)
        h2 = c2.get_field("thickness")
        ts = []
        for _ in range(4):
            c2.set_field("thickness", h2)
            t0 = time.perf_counter()
            it_k, rr_k = c2.solve_state(zero_guess=True)
            ts.append(time.perf_counter() - t0)
        keep = {"forward_ms": float(np.median(ts[1:]) * 1e3), "pcg_iterations": it_k, "apply_ms": c2.bench_kernel("apply", 50),
                "symbolic_s": c2.symbolic_s}
        c2.close()

    # secondary: the same forward solve and adjoint with the 4 x 4 rule of rounds 1-3 where the mesh asks for more (on warped cells
    # it is 7.5e-8 away from the reference's near-exact integration in the gradient: not the headline)
    rule4 = None
    if args.solver == "frontal" and nquad != 4 and m.is_quad and world == 1 and m.ndof <= 20_000_000:   # a second context: two factors in HBM
        c4 = ShellContext(m, device=local_rank, nquad=4)
        for k, v in fields.items():
            c4.set_field(k, v)
        c4.set_penalty_facets(m.penalty_facets(marker))
        c4.enable_frontal(args.leaf)
        c4.set_solver(preconditioner=2, rtol=args.rtol, maxit=50, check_every=1)
        h4 = c4.get_field("thickness")
        tf, ta = [], []
        for _ in range(5):
            c4.set_field("thickness", h4)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            c4.solve_state(zero_guess=True)
            t1 = time.perf_counter()
            c4.total_gradient("compliance", "thickness")
            tf.append(t1 - t0); ta.append(time.perf_counter() - t1)
        rule4 = {"gauss_points_per_direction": 4, "forward_ms": float(np.median(tf[1:]) * 1e3), "adjoint_ms": float(np.median(ta[1:]) * 1e3),
                 "dof_per_s": m.ndof / float(np.median(tf[1:])), "apply_ms": c4.bench_kernel("apply", 50)}
        c4.close()

    if rank == 0:
        out = {
            "metric": "DOF/s (assembly+solve), forward solve of the RM shell; adjoint-gradient wallclock in adjoint_ms",
            "value": m.ndof * args.steps * world / t_fwd,
            "unit": "DOF/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": t_total / args.steps * 1e3,
            "forward_ms": t_fwd / args.steps * 1e3,
            "adjoint_ms": t_adj / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {desc}", "ndof": m.ndof, "cells": m.nel,
                       **({"gauss_points_per_direction": nquad} if m.is_quad else
                          {"triangle_rule_degree": nquad, "quadrature_points_per_cell": ctx.quadrature()[1]}),
                       "gauss_points_rule": ("--nquad" if args.nquad is not None else
                                             "ShellMesh.recommended_nquad: quadrilaterals 4 points per direction on affine cells (exact), 5 / 6 when "
                                             "cells are warped; triangles the symmetric rule of degree 6 (exact for cell-wise polynomial data), 9 "
                                             "when a nodal Poisson ratio varies"),
                       # schedule options taken from the environment (FEMO_OPTIONS): a line from a non-default schedule says so
                       "femo_options_env": dict(ctx.env_options),
                       "true_relres_forward": true_relres,
                       "solver": ("PCG, matrix-free element-by-element operator, multifrontal Cholesky preconditioner "
                                  f"(nested dissection, leaves of about {args.leaf} cells)" if args.solver == "frontal"
                                  else "Jacobi-PCG, matrix-free element-by-element operator"), "rtol": args.rtol,
                       "pcg_iterations_forward": rows[-1][2], "pcg_iterations_adjoint": rows[-1][4],
                       "relres_forward": rows[-1][3], "relres_adjoint": rows[-1][5],
                       "parallelism": "single"},
            "roofline": roof,
            "roofline_spmv": roof_spmv,
            "forward_split_ms": {"assemble_factorise": step.timing["setup_ms"], "pcg": step.timing["krylov_ms"]},
            # once per mesh, outside the timed region (first solve of a new mesh = these + one step)
            "setup_s": setup,
        }
        if multi is not None:
            out["adjoint_ms_4_outputs"] = multi["adjoint_ms_4_outputs"]
            out["adjoint_of_several_outputs"] = multi
        if keep is not None:
            out["keep_numbering"] = keep
        if rule4 is not None:
            out["secondary_rule_4x4"] = rule4
        if prof is not None:
            for o in (roof_main, roof_other):
                if o is not None:
                    out[f"roofline_trailing_{o['bound']}_class"] = o
            out["factorisation_profile_ms"] = {k: v["ms"] for k, v in prof.items() if isinstance(v, dict)}
            out["factorisation_kernels"] = kernels
            sw = np.min([ctx.sweep_profile() for _ in range(3)], axis=0)
            fac_bytes = float(np.sum(ctx.plan.nf.astype(np.float64) * ctx.plan.npiv) * 8)
            out["preconditioner_apply"] = {"ms": float(sw.sum()), "forward_sweep_ms": float(sw[:, 0].sum()), "backward_sweep_ms": float(sw[:, 1].sum()),
                                           "factor_GB_per_sweep": fac_bytes / 1e9,
                                           "achieved_GBs": 2 * fac_bytes / (sw.sum() * 1e-3) / 1e9,
                                           "frac_of_hbm_peak": 2 * fac_bytes / (sw.sum() * 1e-3) / 1e9 / HBM_PEAK_GBS}
            out["frontal"] = {k: (float(v) if not isinstance(v, int) else v) for k, v in ctx.frontal_info().items()}
        if not args.no_cpu_baseline and world == 1:
            ctx.close()                                   # the CPU leg wants the host memory bandwidth to itself
            out["cpu_baseline"] = cpu_baseline(m, fields, marker, args.leaf, workload=args.workload, nquad=nquad)
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

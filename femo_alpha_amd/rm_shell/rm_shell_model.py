"""``RMShellModel``: the model facade example scripts call.

Constructor and ``evaluate`` follow femo_alpha/rm_shell/rm_shell_model.py:31-81,364-490; the
``mesh`` argument is a ``femo_alpha_amd.mesh.ShellMesh`` (nodes + connectivity, the data
``reconstructFEAMesh`` takes) instead of a dolfinx mesh.  Outputs, as in the reference:
``disp_solid`` (state), ``compliance``, ``mass``, ``elastic_energy``, ``pnorm_stress``, the DG1 field
``stress``, ``disp_extracted`` and ``aggregated_stress``.
"""
from __future__ import annotations

import numpy as np

from .. import csdl
from ..csdl_alpha_opt.fea_model import FEAModel
from ..fea.fea_hip import FEA, DirichletBC, Function
from ..mesh import ShellMesh
from .rm_shell_pde import FacetSet, RMShellPDE


def solve_linear(A, b, ctx):
    """pressure = A^-1 force as a differentiable node: ``csdl.solve_linear`` when csdl_alpha is installed (what the
    reference calls, rm_shell_model.py:420); otherwise an explicit operation whose solve runs on the device
    (``femo_force_to_pressure``: Jacobi-PCG with the consistent [CG1]^3 mass matrix applied cell by cell -- the matrix ``A`` of
    ``construct_force_to_pressure_map`` is the same operator and is not touched).  ``ctx``: the shell context whose device runs
    that solve (required: there is no host fallback)."""
    if csdl.HAVE_CSDL_ALPHA:
        return csdl.solve_linear(A.toarray(), b)
    if ctx is None:
        raise ValueError("solve_linear: without csdl_alpha the force -> pressure solve runs on the device and needs the shell context")

    class _Solve(csdl.CustomExplicitOperation):
        def evaluate(self, rhs):
            self.declare_input("b", rhs)
            x = self.create_output("x", rhs.shape)
            self.declare_derivative_parameters("x", "*", dependent=True)
            self._finish_evaluate()
            return x

        def compute(self, input_vals, output_vals):
            output_vals["x"] = ctx.force_to_pressure(np.asarray(input_vals["b"], dtype=np.float64))

        def vjp(self, bar):                      # A is symmetric: bar_b = A^-T bar_x
            return ctx.force_to_pressure(bar)

    return _Solve().evaluate(b)


def createCustomMeasure(mesh: ShellMesh, dim, SubdomainFunc, measure: str, tag: int):
    """Tagged facet measure (femo_alpha/fea/utils_dolfinx.py:555-565): 'ds' = boundary facets whose
    vertices all satisfy the marker, 'dS' = interior facets, integrated from both sides."""
    if dim != 1:
        raise ValueError("shell facets have dimension 1")
    if measure == "ds":
        ed = mesh.locate_facets(SubdomainFunc, boundary_only=True)
        pairs = np.stack([mesh.edge_cells[ed, 0], mesh.edge_local[ed, 0]], axis=1)
    else:
        ed = mesh.locate_facets(SubdomainFunc, boundary_only=False)
        ed = ed[mesh.edge_cells[ed, 1] >= 0]
        if mesh.is_manifold:
            pairs = np.vstack([np.stack([mesh.edge_cells[ed, s], mesh.edge_local[ed, s]], axis=1) for s in (0, 1)])
        else:
            pairs = mesh.facet_incidence(ed)           # branching edges: every incident cell
    sets = {tag: FacetSet(pairs)}
    return lambda t: sets.get(t, FacetSet(np.zeros((0, 2), np.int32)))


class RMShellModel:
    def __init__(self, mesh: ShellMesh, shell_bc_func: callable = None, element_wise_material=False, rho=100,
                 PENALTY_BC=True, additional_outputs=None, mesh_tags=None, record=True, elementwise_pressure=False,
                 device=0, renumber=False, nquad=None):
        # caller order <-> solver order.  dolfinx reorders every mesh it is given and the reference carries the maps
        # (rm_shell_model.py:116, 396-438, 505-527); with renumber=True this build does the same with a Morton order of
        # the cells (ShellMesh.renumbered): inputs are gathered into solver order, nodal displacements come back in
        # caller order, the state ``disp_solid`` lives in solver order as it does in the reference.
        self.caller_mesh = mesh
        if renumber:
            mesh, self.vertex_of_new, self.cell_of_new = mesh.renumbered()
        else:
            self.vertex_of_new, self.cell_of_new = np.arange(mesh.nn), np.arange(mesh.nel)
        self.new_of_vertex = np.empty(mesh.nn, dtype=np.int64)
        self.new_of_vertex[self.vertex_of_new] = np.arange(mesh.nn)
        self.new_of_cell = np.empty(mesh.nel, dtype=np.int64)
        self.new_of_cell[self.cell_of_new] = np.arange(mesh.nel)
        if mesh_tags is not None:
            for inds in mesh_tags.values():
                inds = np.asarray(inds, dtype=np.int64)
                if inds.size and (inds.min() < 0 or inds.max() >= mesh.nel):
                    raise ValueError("mesh_tags: cell index out of range")
            mesh_tags = {tag: self.new_of_cell[np.asarray(inds, dtype=np.int64)] for tag, inds in mesh_tags.items()}
        self.mesh = mesh
        self.mesh_tags = mesh_tags
        self.additional_outputs = additional_outputs
        self.shell_bc_func = shell_bc_func
        self.element_wise_material = element_wise_material
        self.record = record
        self.m, self.rho = 1e-6, rho
        self.PENALTY_BC = PENALTY_BC
        self.nel, self.nn = mesh.nel, mesh.nn
        self.elementwise_pressure = elementwise_pressure
        self.device = device
        # n x n Gauss points per quadrilateral for the static forms (2..5).  The reference leaves the degree to UFL's
        # estimate, which on quadrilaterals comes out near 47 (scripts/ufl_degree_estimate.py): exact integration.  Default:
        # what the mesh asks for -- 4 on affine cells (exact there), 5 as soon as one cell is warped (within 1e-9 of the
        # limit at BASELINE config 3; ShellMesh.recommended_nquad).
        # On triangles nquad is the degree of the symmetric rule: 6 (exact for cell-wise polynomial data), and the context raises it to
        # UFL's 9 by itself when a nodal Poisson ratio that varies over the cells arrives -- unless the caller names the rule here.
        self._nquad_arg = None if nquad is None else int(nquad)
        self.association_table = None
        if shell_bc_func is None:
            raise ValueError("Please provide the shell bc location function.\n"
                             " Example:\n def ClampedBoundary(x):\n    return np.less(x[1], 0.0)")
        self.set_up_bcs(shell_bc_func, PENALTY_BC)
        self.set_up_fea()

    @property
    def nquad(self):
        """The rule in use (ShellContext.nquad)."""
        return self.shell_pde.ctx.nquad

    def set_up_subdomains(self, mesh_tags):
        """mesh_tags: {tag: [cell indices]} without duplicates (rm_shell_model.py:101-133).  Builds
        ``association_table`` {tag: sub-domain index} and hands the per-cell index array to the backend."""
        vals = -np.ones(self.mesh.nel, dtype=np.int32)
        for i, inds in enumerate(mesh_tags.values()):
            inds = np.asarray(inds, dtype=np.int64)
            if inds.size and (inds.min() < 0 or inds.max() >= self.mesh.nel):
                raise ValueError("mesh_tags: cell index out of range")
            if np.any(vals[inds] != -1):
                raise ValueError("mesh_tags: a cell may carry one tag only")
            vals[inds] = i
        self.association_table = {key: i for i, key in enumerate(mesh_tags.keys())}
        self.shell_pde.ctx.set_cell_tags(vals, len(mesh_tags))

    def set_up_bcs(self, bc_locs_func, PENALTY_BC):
        if PENALTY_BC:
            self.dss = createCustomMeasure(self.mesh, 1, bc_locs_func, measure="ds", tag=100)(100)
            self.dSS = createCustomMeasure(self.mesh, 1, bc_locs_func, measure="dS", tag=100)(100)
        else:
            self.dss = self.dSS = None

    def set_up_fea(self):
        mesh = self.mesh
        shell_pde = self.shell_pde = RMShellPDE(mesh, element_wise_material=self.element_wise_material,
                                                elementwise_pressure=self.elementwise_pressure, device=self.device,
                                                nquad=self._nquad_arg)
        fea = FEA(mesh)
        fea.PDE_SOLVER = "Newton"
        fea.REPORT = False
        fea.record = False                    # XDMF recording is out of scope; the flag is accepted and ignored
        fea.linear_problem = True
        h, f = Function(shell_pde.VT), Function(shell_pde.VF)
        E, nu, density = Function(shell_pde.VT), Function(shell_pde.VT), Function(shell_pde.VT)
        uhat = Function(shell_pde.VU)
        w = Function(shell_pde.W)
        if not self.PENALTY_BC:
            dofs = mesh.locate_dofs_geometrical(self.shell_bc_func)
            shell_pde.ctx.set_strong_dofs(dofs)
            fea.bc = [DirichletBC(dofs)]
        g = Function(shell_pde.W)
        residual_form = shell_pde.pdeRes(h=h, w=w, uhat=uhat, f=f, E=E, nu=nu, penalty=self.PENALTY_BC,
                                         dss=self.dss, dSS=self.dSS, g=g)
        compliance_form = shell_pde.compliance(w, uhat, h, f)
        mass_form = shell_pde.mass(uhat, h, density)
        elastic_energy_form = shell_pde.elastic_energy(w, uhat, h, E)
        pnorm_stress_form = shell_pde.pnorm_stress(w, uhat, h, E, nu, None, m=self.m, rho=self.rho, alpha=None,
                                                   regularization=False)
        stress_form = shell_pde.von_Mises_stress(w, uhat, h, E, nu, surface="Top")
        fea.add_input("thickness", h, init_val=0.001)
        fea.add_input("F_solid", f, init_val=1.0)
        fea.add_input("E", E, init_val=1.0)
        fea.add_input("nu", nu, init_val=1.0)
        fea.add_input("density", density, init_val=1.0)
        fea.add_input("uhat", uhat, init_val=0.0)
        fea.add_state(name="disp_solid", function=w, residual_form=residual_form,
                      arguments=["thickness", "F_solid", "E", "nu", "uhat"])
        fea.add_output(name="compliance", form=compliance_form, arguments=["disp_solid", "F_solid", "thickness", "uhat"])
        fea.add_output(name="mass", form=mass_form, arguments=["thickness", "density", "uhat"])
        fea.add_output(name="elastic_energy", form=elastic_energy_form, arguments=["thickness", "disp_solid", "E", "uhat"])
        fea.add_output(name="pnorm_stress", form=pnorm_stress_form, arguments=["thickness", "disp_solid", "E", "nu", "uhat"])
        if self.mesh_tags is not None:
            self.set_up_subdomains(self.mesh_tags)
            for tag, i in self.association_table.items():
                # a stress aggregate per sub-domain, named after the caller's tag (rm_shell_model.py:242-253)
                form_i = shell_pde.pnorm_stress(w, uhat, h, E, nu, i, m=self.m, rho=self.rho, alpha=None, regularization=False)
                fea.add_output(name="pnorm_stress_" + str(tag), form=form_i,
                               arguments=["thickness", "disp_solid", "E", "nu", "uhat"])
        fea.add_field_output(name="stress", form=stress_form, arguments=["thickness", "disp_solid", "E", "nu", "uhat"],
                             function_space=("DG", 1), record=False, vtk=True)
        self.fea = fea

    def evaluate(self, force_vector, thickness, E, nu, density, node_disp=None, debug_mode=False, is_pressure=True):
        shell_inputs = csdl.VariableGroup()
        mesh = self.mesh
        # caller order == solver order here; the gathers are kept so that a renumbered mesh object
        # can plug in its permutations exactly where the reference applies them (:398-438)
        mat_idx = self.cell_of_new if self.element_wise_material else self.vertex_of_new
        shell_inputs.thickness = thickness[mat_idx]
        shell_inputs.E = E[mat_idx]
        shell_inputs.nu = nu[mat_idx]
        shell_inputs.density = density[mat_idx]
        prs_idx = self.cell_of_new if self.elementwise_pressure else self.vertex_of_new
        reshaped_force = csdl.reshape(force_vector[prs_idx], (-1,))
        if is_pressure:
            shell_inputs.F_solid = reshaped_force
        else:
            # nodal forces -> nodal pressures through the consistent mass matrix (rm_shell_model.py:414-421)
            # the matrix itself is only needed by csdl_alpha's solve_linear; the stand-in solves on the device
            if self.elementwise_pressure:          # as construct_force_to_pressure_map says, with or without csdl_alpha
                raise NotImplementedError("force -> pressure conversion is defined for nodal pressures")
            A = self.shell_pde.construct_force_to_pressure_map() if csdl.HAVE_CSDL_ALPHA else None
            shell_inputs.F_solid = solve_linear(A, reshaped_force, ctx=self.shell_pde.ctx)
        shell_inputs.F_solid.add_name("F_solid")
        if node_disp is None:
            node_disp = csdl.Variable(value=0.0, shape=(mesh.nn, 3), name="node_disp")
        reshaped_node_disp = node_disp[self.vertex_of_new].reshape((-1,))
        reshaped_node_disp.add_name("uhat")
        shell_inputs.uhat = reshaped_node_disp
        for n in ("thickness", "E", "nu", "density"):
            getattr(shell_inputs, n).add_name(n)

        solid_model = FEAModel(fea=[self.fea], fea_name="rm_shell")
        shell_outputs = solid_model.evaluate(shell_inputs, debug_mode=debug_mode)

        disp_extracted = DisplacementExtractionModel(shell_pde=self.shell_pde).evaluate(shell_outputs.disp_solid)
        disp_extracted = disp_extracted[self.new_of_vertex]              # solver order -> caller order
        disp_extracted.add_name("disp_extracted")
        shell_outputs.disp_extracted = disp_extracted
        aggregated_stress = AggregatedStressModel(m=self.m, rho=self.rho).evaluate(shell_outputs.pnorm_stress)
        aggregated_stress.add_name("aggregated_stress")
        shell_outputs.aggregated_stress = aggregated_stress
        return shell_outputs


class AggregatedStressModel:
    """aggregated_stress = pnorm^(1/rho) / m (rm_shell_model.py:493-503)."""

    def __init__(self, m: float, rho: int):
        self.m, self.rho = m, rho

    def evaluate(self, pnorm_stress):
        return 1 / self.m * pnorm_stress ** (1 / self.rho)


class DisplacementExtractionModel:
    """(nn, 3) vertex displacements in caller node order (rm_shell_model.py:505-527)."""

    def __init__(self, shell_pde: RMShellPDE):
        self.shell_pde = shell_pde

    def evaluate(self, disp_vec):
        nn = self.shell_pde.mesh.nn
        return disp_vec[np.arange(3 * nn)].reshape((nn, 3))


class ForceReshapingModel:
    def __init__(self, shell_pde: RMShellPDE):
        self.shell_pde = shell_pde

    def evaluate(self, nodal_force_mat):
        return csdl.reshape(nodal_force_mat[np.arange(self.shell_pde.mesh.nn)], (-1,))

"""Dynamic operators: ``StateOperation`` (whole displacement history as one implicit state),
``TotalStrainEnergyOperation`` and ``VolumeOperation`` -- interfaces of
femo_alpha/dynamic_rm_shell/state_operation_dynamic.py:20-137,141-706, total_strain_energy_operation.py:20-138
and volume_operation.py:20-70.  Histories cross the boundary as vectors flattened column-major from
(fe_dofs, time_levels) (dynamic_rm_shell/utils.py:9-16)."""
import numpy as np

from .. import csdl


def stack_array_into_vector(a):
    return np.ravel(a, order="F")


def reshape_vector_into_array(v, n_cols):
    n_rows = v.shape[0] // n_cols
    if v.shape[0] != n_rows * n_cols:
        raise ValueError(f"`inp_vec` of shape {v.shape} cannot be reshaped into {n_rows} rows and {n_cols} columns")
    return np.reshape(v, (n_rows, n_cols), order="F")


def _declare(op, inputs):
    for name in op.args_dict:
        var = getattr(inputs, name)
        if var is None:
            raise ValueError(f"Variable {name} not found in the FEA model.")
        op.declare_input(name, var)


def _finish(op):
    if not csdl.HAVE_CSDL_ALPHA:
        op._finish_evaluate()


class StateOperation(csdl.experimental.CustomImplicitOperation):
    def __init__(self, plate_sim, gradient_mode="numpy", debug_mode=False, record=False, path="./"):
        super().__init__()
        csdl.check_parameter(plate_sim, "plate_sim")
        self.plate_sim, self.gradient_mode, self.debug_mode = plate_sim, gradient_mode, debug_mode
        self.state_name = "disp_history"
        self.args_dict = ["thickness", "force_history"]
        self.input_name = "thickness"
        self.bc_dofs = plate_sim.bc_dofs
        self.eval_iter = 0

    def evaluate(self, inputs):
        _declare(self, inputs)
        ps = self.plate_sim
        state = self.create_output(self.state_name, shape=(ps.fe_dofs * ps.time_levels,))
        state.add_name(self.state_name)
        self.declare_derivative_parameters(self.state_name, "*", dependent=True)
        _finish(self)
        return state

    def solve_residual_equations(self, input_vals, output_vals):
        ps = self.plate_sim
        ps.update_t(input_vals["thickness"])
        ps.update_f_history(np.asarray(input_vals["force_history"]).reshape(-1, 3 * ps.nn))
        self.eval_iter += 1
        output_vals[self.state_name] = stack_array_into_vector(ps.solve_dynamic_problem())

    def apply_inverse_jacobian(self, input_vals, output_vals, d_outputs, d_residuals, mode):
        ps = self.plate_sim
        if mode == "rev":
            G = reshape_vector_into_array(np.asarray(d_outputs[self.state_name]), ps.time_levels)
            d_residuals[self.state_name] = stack_array_into_vector(ps.adjoint_history(G))
        elif mode == "fwd":
            # the direct method / tangent linear model (state_operation_dynamic.py:534-605)
            dR = reshape_vector_into_array(np.asarray(d_residuals[self.state_name]), ps.time_levels)
            d_outputs[self.state_name] = stack_array_into_vector(ps.tangent_history(dR))
        else:
            raise ValueError("mode must be either 'fwd' or 'rev'.")

    def compute_jacvec_product(self, input_vals, output_vals, d_inputs, d_outputs, d_residuals, mode):
        ps = self.plate_sim
        if mode == "rev":
            if self.state_name in d_residuals:
                Lam = reshape_vector_into_array(np.asarray(d_residuals[self.state_name]), ps.time_levels)
                g_t, g_f = ps.residual_T_products(Lam)
                if "thickness" in d_inputs:
                    d_inputs["thickness"] += g_t
                if "force_history" in d_inputs:
                    d_inputs["force_history"] += g_f.reshape(np.shape(d_inputs["force_history"]))
        elif mode == "fwd":
            # d_inputs, d_outputs --> d_residuals (state_operation_dynamic.py:228-329)
            if self.state_name in d_residuals:
                dY = reshape_vector_into_array(np.asarray(d_outputs[self.state_name]), ps.time_levels) if self.state_name in d_outputs else None
                dth = np.asarray(d_inputs["thickness"]) if "thickness" in d_inputs else None
                dF = np.asarray(d_inputs["force_history"]).reshape(ps.time_levels, -1) if "force_history" in d_inputs else None
                d_residuals[self.state_name] = d_residuals[self.state_name] + stack_array_into_vector(ps.jacobian_products_fwd(dY, dth, dF))
        else:
            raise ValueError("mode must be either 'fwd' or 'rev'.")


class TotalStrainEnergyOperation(csdl.CustomExplicitOperation):
    """sum over time levels of the strain energy (total_strain_energy_operation.py:56-125)."""

    def __init__(self, plate_sim):
        super().__init__()
        csdl.check_parameter(plate_sim, "plate_sim")
        self.plate_sim = plate_sim
        self.args_dict = ["thickness", "disp_history"]
        self.output_name = "total_strain_energy"
        self.regularization = False

    def evaluate(self, inputs):
        _declare(self, inputs)
        out = self.create_output(self.output_name, (1,))
        out.add_name(self.output_name)
        self.declare_derivative_parameters(self.output_name, "*", dependent=True)
        _finish(self)
        return out

    def compute(self, input_vals, output_vals):
        ps = self.plate_sim
        W = reshape_vector_into_array(np.asarray(input_vals["disp_history"]), ps.time_levels)
        ps.update_t(input_vals["thickness"])
        output_vals[self.output_name] = np.array([sum(ps.assembleStrainEnergy(W[:, i]) for i in range(ps.time_levels))])

    def compute_derivatives(self, input_vals, output_vals, derivatives):
        ps = self.plate_sim
        W = reshape_vector_into_array(np.asarray(input_vals["disp_history"]), ps.time_levels)
        ps.update_t(input_vals["thickness"])
        dEdt = np.zeros(ps.num_var)
        dEdw = np.zeros_like(W)
        for i in range(ps.time_levels):
            gt, gw = ps.strain_energy_gradients(W[:, i])
            dEdt += gt
            dEdw[:, i] = gw
        derivatives[self.output_name, "thickness"] = dEdt
        derivatives[self.output_name, "disp_history"] = stack_array_into_vector(dEdw)


class VolumeOperation(csdl.CustomExplicitOperation):
    """int t dx and its thickness gradient (volume_operation.py:57-70)."""

    def __init__(self, plate_sim):
        super().__init__()
        csdl.check_parameter(plate_sim, "plate_sim")
        self.plate_sim = plate_sim
        self.args_dict = ["thickness"]
        self.output_name = "volume"

    def evaluate(self, inputs):
        _declare(self, inputs)
        out = self.create_output(self.output_name, (1,))
        out.add_name(self.output_name)
        self.declare_derivative_parameters(self.output_name, "*", dependent=True)
        _finish(self)
        return out

    def compute(self, input_vals, output_vals):
        self.plate_sim.update_t(input_vals["thickness"])
        output_vals["volume"] = np.array([self.plate_sim.volume()])

    def compute_derivatives(self, input_vals, output_vals, derivatives):
        self.plate_sim.update_t(input_vals["thickness"])
        derivatives["volume", "thickness"] = self.plate_sim.dvolume_dt()

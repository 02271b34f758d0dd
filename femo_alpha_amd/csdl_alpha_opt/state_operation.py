"""``StateOperation``: the implicit operator  inputs -> state  of one registered PDE state.

Same constructor, methods and dictionary conventions as the reference class
(femo_alpha/csdl_alpha_opt/state_operation.py:8-296), so a CSDL simulator -- or the in-tree
stand-in -- drives it unchanged:

  solve_residual_equations      forward solve            (reference :86-131)
  apply_inverse_jacobian        K^-1 / K^-T solve        (:188-220)
  compute_jacvec_product        (dR/d arg)^T lambda      (:134-186)

Differences that are deliberate: the Jacobian-side data are
matrix-free and always consistent with the current inputs and state (the reference assembles
them at ``opt_iter == 1`` only, quirk Q2); forward-mode products with dR/d(input) raise
``NotImplementedError`` where the reference raises ``KeyError`` (quirk Q1).
"""
from .. import csdl
from ..fea.fea_hip import (FEA, assembleMatrix, assembleSystem, computeMatVecProductBwd, computeMatVecProductFwd,
                           computePartials, createFunction, getFuncArray, setUpKSP_MUMPS, update)
from ._common import banner, collect_arguments, declare_all_inputs, finish, push_inputs


class StateOperation(csdl.experimental.CustomImplicitOperation):
    def __init__(self, fea, args_name_list, state_name, debug_mode=False):
        super().__init__()
        csdl.check_parameter(fea, "fea", types=FEA)
        csdl.check_parameter(args_name_list, "args_name_list", types=list)
        csdl.check_parameter(state_name, "state_name", types=str)
        self.fea = fea
        self.state_name = self._label = state_name
        self.debug_mode = debug_mode
        self.args_dict = collect_arguments(fea, args_name_list, allow_states=False)
        self.fea_state = fea.states_dict[state_name]
        self.fea_dR = self.fea_state["d_residual"]
        self.fea_du = self.fea_state["d_state"]
        self.set_up_fea_derivatives()

    # ------------------------------------------------------------------ graph construction
    def evaluate(self, inputs: csdl.VariableGroup):
        banner(self, "evaluate")
        declare_all_inputs(self, inputs)
        state = self.create_output(self.state_name, shape=(self.fea_state["shape"],))
        state.add_name(self.state_name)
        self.declare_derivative_parameters(self.state_name, "*", dependent=True)
        finish(self)
        return state

    # ------------------------------------------------------------------ forward
    def solve_residual_equations(self, input_vals, output_vals):
        banner(self, "solve_residual_equations")
        self.fea.opt_iter += 1
        push_inputs(self, input_vals)
        self.fea.solve(self.fea_state["residual_form"], self.fea_state["function"], self.fea.bc)
        output_vals[self.state_name] = getFuncArray(self.fea_state["function"])
        if self.fea.linear_problem is False or self.fea.opt_iter == 1:
            self.assemble_derivatives(input_vals, output_vals)

    # ------------------------------------------------------------------ derivatives
    def compute_jacvec_product(self, input_vals, output_vals, d_inputs, d_outputs, d_residuals, mode):
        banner(self, "compute_jacvec_product")
        name = self.state_name
        if mode == "fwd":
            if name in d_residuals:
                if name in d_outputs:
                    update(self.fea_du, d_outputs[name])
                    d_residuals[name] += computeMatVecProductFwd(self.dRdu, self.fea_du)
                for arg, entry in self.dR_df_dict.items():
                    if arg in d_inputs:
                        update(entry["fea_df"], d_inputs[arg])
                        d_residuals[name] += computeMatVecProductFwd(entry["dRdf"], entry["fea_df"])
        elif mode == "rev":
            if name in d_residuals:
                update(self.fea_dR, d_residuals[name])
                for arg, entry in self.dR_df_dict.items():
                    if arg in d_inputs:
                        d_inputs[arg] += computeMatVecProductBwd(entry["dRdf"], self.fea_dR)
        else:
            raise ValueError("mode must be either 'fwd' or 'rev'.")

    def apply_inverse_jacobian(self, input_vals, output_vals, d_outputs, d_residuals, mode):
        banner(self, "apply_inverse_jacobian")
        name = self.state_name
        if mode == "fwd":
            d_outputs[name] = self.fea.solveLinearFwd(self.fea_du, self.A, self.fea_dR, d_residuals[name], self.ksp)
        elif mode == "rev":
            # (d_outputs[name] of shape (k, ndof): the seeds of k outputs at once -- one grouped adjoint solve, FEA.solveLinearBwd)
            d_residuals[name] = self.fea.solveLinearBwd(self.fea_dR, self.A, self.fea_du, d_outputs[name], self.ksp)
            for bc in self.fea.bc:
                d_residuals[name][..., bc.dof_indices()[0]] = 0.0
        else:
            raise ValueError("mode must be either 'fwd' or 'rev'.")

    def set_up_fea_derivatives(self):
        banner(self, "set_up_fea_derivatives")
        st = self.fea_state
        if st["dR_du"] is None:
            st["dR_du"] = computePartials(st["residual_form"], st["function"])
        given = st["dR_df_list"]
        self.dR_df_dict = {}
        for k, arg in enumerate(st["arguments"]):
            form = computePartials(st["residual_form"], self.args_dict[arg]["function"]) if given is None else given[k]
            self.dR_df_dict[arg] = dict(dR_df=form, fea_df=createFunction(self.args_dict[arg]["function"]))

    def assemble_derivatives(self, input_vals, output_vals):
        banner(self, "assemble_derivatives")
        push_inputs(self, input_vals)
        update(self.fea_state["function"], output_vals[self.state_name])
        for entry in self.dR_df_dict.values():
            entry["dRdf"] = assembleMatrix(entry["dR_df"])
        self.dRdu = assembleMatrix(self.fea_state["dR_du"])
        self.A, _ = assembleSystem(self.fea_state["dR_du"], self.fea_state["residual_form"], bcs=self.fea.bc)
        self.ksp = setUpKSP_MUMPS(self.A)

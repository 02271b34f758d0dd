"""``OutputOperation`` (scalar outputs) and ``OutputFieldOperation`` (projected fields).

Interface of the reference classes (femo_alpha/csdl_alpha_opt/output_operation.py:6-128):
``compute`` pushes the inputs and assembles the scalar; ``compute_derivatives`` returns one
gradient vector per declared argument under the key ``(output_name, arg_name)``.
"""
import numpy as np

from .. import csdl
from ..fea.fea_hip import FEA, assemble, computePartials, getFuncArray
from ._common import collect_arguments, declare_all_inputs, finish, push_inputs


class OutputOperation(csdl.CustomExplicitOperation):
    def __init__(self, fea, args_name_list, output_name):
        super().__init__()
        csdl.check_parameter(fea, "fea", types=FEA)
        csdl.check_parameter(args_name_list, "args_name_list", types=list)
        csdl.check_parameter(output_name, "output_name", types=str)
        self.fea = fea
        self.output_name = self._label = output_name
        self.args_dict = collect_arguments(fea, args_name_list, allow_states=True)
        self.fea_output = fea.outputs_dict[output_name]
        self.output_dim = 0

    def evaluate(self, inputs: csdl.VariableGroup):
        declare_all_inputs(self, inputs)
        output = self.create_output(self.output_name, (1,))
        output.add_name(self.output_name)
        self.declare_derivative_parameters(self.output_name, "*", dependent=True)
        finish(self)
        return output

    def compute(self, input_vals, output_vals):
        push_inputs(self, input_vals)
        output_vals[self.output_name] = np.array([assemble(self.fea_output["form"])])

    def compute_derivatives(self, input_vals, output_vals, derivatives):
        push_inputs(self, input_vals)
        for arg in input_vals:
            partial = computePartials(self.fea_output["form"], self.args_dict[arg]["function"])
            derivatives[self.output_name, arg] = assemble(partial, dim=self.output_dim + 1)


class OutputFieldOperation(csdl.CustomExplicitOperation):
    """Field output; like the reference (:100-128) it defines ``compute`` only, no derivatives."""

    def __init__(self, fea, args_name_list, output_name):
        super().__init__()
        csdl.check_parameter(fea, "fea", types=FEA)
        csdl.check_parameter(args_name_list, "args_name_list", types=list)
        csdl.check_parameter(output_name, "output_name", types=str)
        self.fea = fea
        self.output_name = self._label = output_name
        self.args_dict = collect_arguments(fea, args_name_list, allow_states=True)
        self.fea_output = fea.outputs_field_dict[output_name]
        self.output_dim = 1

    def evaluate(self, inputs: csdl.VariableGroup):
        declare_all_inputs(self, inputs)
        output = self.create_output(self.output_name, (self.fea_output["shape"],))
        output.add_name(self.output_name)
        self.declare_derivative_parameters(self.output_name, "*", dependent=True)
        finish(self)
        return output

    def compute(self, input_vals, output_vals):
        push_inputs(self, input_vals)
        self.fea.projectFieldOutput(self.fea_output["form"], self.fea_output["function"])
        output_vals[self.output_name] = getFuncArray(self.fea_output["function"])

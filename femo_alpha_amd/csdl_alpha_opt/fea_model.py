"""``FEAModel``: instantiate one operation per registered state / output and chain them.

Interface of femo_alpha/csdl_alpha_opt/fea_model.py:6-64: ``FEAModel(fea=[...], fea_name)`` and
``evaluate(inputs, debug_mode)`` returning the same VariableGroup enriched with the states and
outputs (states first, then scalar outputs, then field outputs)."""
from .. import csdl
from .output_operation import OutputFieldOperation, OutputOperation
from .state_operation import StateOperation


class FEAModel:
    def __init__(self, fea, fea_name="fea"):
        self.parameters = {"fea": fea, "fea_name": fea_name}

    def evaluate(self, inputs: csdl.VariableGroup, debug_mode=False):
        self.fea_list = self.parameters["fea"]
        variables = inputs
        self.operations = []
        for fea in self.fea_list:
            stages = ((fea.states_dict, lambda n, a: StateOperation(fea=fea, state_name=n, args_name_list=a,
                                                                    debug_mode=debug_mode)),
                      (fea.outputs_dict, lambda n, a: OutputOperation(fea=fea, output_name=n, args_name_list=a)),
                      (fea.outputs_field_dict, lambda n, a: OutputFieldOperation(fea=fea, output_name=n,
                                                                                  args_name_list=a)))
            for registry, make in stages:
                for name, entry in registry.items():
                    op = make(name, entry["arguments"])
                    setattr(variables, name, op.evaluate(variables))
                    self.operations.append(op)
        return variables

"""Shared plumbing of the three operator classes."""
from .. import csdl


def banner(op, what):
    if getattr(op, "debug_mode", False):
        print("=" * 15 + str(op._label) + "=" * 15)
        print(f"CSDL: Running {what}()...")
        print("=" * 40)


def collect_arguments(fea, names, allow_states):
    """name -> registry entry, looked up in fea.inputs_dict (and fea.states_dict for outputs)."""
    found = {}
    for n in names:
        if n in fea.inputs_dict:
            found[n] = fea.inputs_dict[n]
        elif allow_states and n in fea.states_dict:
            found[n] = fea.states_dict[n]
        elif not allow_states:
            raise KeyError(n)
    return found


def declare_all_inputs(op, inputs):
    for n in op.args_dict:
        var = getattr(inputs, n)
        if var is None:
            raise ValueError(f"Variable {n} not found in the FEA model.")
        op.declare_input(n, var)


def push_inputs(op, input_vals):
    from ..fea.fea_hip import update
    for n in input_vals:
        update(op.args_dict[n]["function"], input_vals[n])


def finish(op):
    """Inline execution hook of the stand-in; the real csdl_alpha drives execution itself."""
    if not csdl.HAVE_CSDL_ALPHA:
        op._finish_evaluate()

"""femo_alpha_amd: MI355X-native Reissner-Mindlin shell forward + adjoint path behind
femo_alpha's FEAModel / StateOperation / OutputOperation / RMShellModel operator surface."""
from .mesh import ShellMesh, plate_mesh, quads_to_triangles, wing_skin_mesh  # noqa: F401

__all__ = ["ShellMesh", "plate_mesh", "wing_skin_mesh", "quads_to_triangles"]

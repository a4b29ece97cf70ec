"""drtk.utils of the drop-in: the loader and the pinhole projection (drtk/utils/__init__.py:8-22 restricted to the
path); the mesh-geometry helpers (face_dpdt, face_info, vert_normals, vert_binormals, index) are outside it."""
from drtk.utils.load_torch_ops import load_torch_ops  # noqa: F401
from drtk.utils.projection import DISTORTION_MODES, project_pinhole, project_points  # noqa: F401

_OUT_OF_SCOPE = {"face_dpdt", "face_info", "vert_binormals", "vert_normals", "index", "project_points_grad"}


def __getattr__(name):
    if name in _OUT_OF_SCOPE:
        raise AttributeError(f"drtk.utils.{name} is not provided by drtk_amd's drop-in (outside the rasterize -> render -> "
                             "interpolate -> edge_grad path; DESIGN.md, out of scope)")
    raise AttributeError(f"module 'drtk.utils' has no attribute '{name}'")

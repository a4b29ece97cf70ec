"""drtk_amd -- MI355X-native differentiable rasterization hot path
(rasterize -> render -> interpolate -> edge_grad), drop-in for the `drtk.*` functions of
facebookresearch/DRTK on PyTorch-ROCm.  Kernels: hand-written HIP for gfx950 in
`drtk_amd/csrc`, C ABI in `include/drtk_amd.h`."""
from drtk_amd.edge_grad_estimator import edge_grad_estimator  # noqa: F401
from drtk_amd.graph import capture_step  # noqa: F401
from drtk_amd.interpolate import (  # noqa: F401
    interpolate,
    interpolate_masked,
    interpolation_matrix,
    interpolation_normal_matrix,
)
from drtk_amd.mipmap_grid_sample import mipmap_grid_sample  # noqa: F401
from drtk_amd.rasterize import get_depth_order, rasterize, rasterize_with_depth, set_depth_order  # noqa: F401
from drtk_amd.render import render  # noqa: F401
from drtk_amd.screen_space_uv_derivative import screen_space_uv_derivative  # noqa: F401
from drtk_amd.transform import transform, transform_with_v_cam  # noqa: F401

__version__ = "0.1.0"

# The public surface.  Same names, arguments and defaults as `drtk.*` for everything on the hot path and its "next"
# rows; `interpolate_masked` (interpolate with the background written as 0), `capture_step` (a whole step as a
# HIP graph) and `set_depth_order` / `get_depth_order` (the rasterizer's depth order: the reference's source, or the
# reference as its setup.py builds it) are this package's additions.  Not
# provided: grid_scatter, msi, filter2d and the pure-PyTorch `*_ref` models (out of scope).
__all__ = [
    "rasterize",
    "rasterize_with_depth",
    "render",
    "interpolate",
    "interpolate_masked",
    "edge_grad_estimator",
    "interpolation_matrix",
    "interpolation_normal_matrix",
    "mipmap_grid_sample",
    "screen_space_uv_derivative",
    "transform",
    "transform_with_v_cam",
    "capture_step",
    "set_depth_order",
    "get_depth_order",
]

"""`import drtk` drop-in: the reference's import surface (drtk/__init__.py:8-33) for everything on the
rasterize -> render -> interpolate -> edge_grad path and its "next" rows, served by drtk_amd's HIP kernels.

Code written against facebookresearch/DRTK -- `from drtk import rasterize, render, interpolate,
edge_grad_estimator` (test/two_triangles.py:11), `drtk.transform(...)`, `drtk.mipmap_grid_sample(...)`,
`from drtk.screen_space_uv_derivative import screen_space_uv_derivative`, `drtk.utils.load_torch_ops(
"drtk.rasterize_ext")`, `import drtk.rasterize_ext` -- runs unchanged with this directory on the path.
Names of the reference that are outside the path (grid_scatter, msi, filter2d, the pure-PyTorch `*_ref`
models, the mesh-geometry helpers of drtk.utils) raise an AttributeError that says so instead of being absent
silently.  The package proper is `drtk_amd`; nothing is implemented here."""
import sys as _sys

import drtk_amd as _impl
from drtk_amd import (  # noqa: F401
    edge_grad_estimator,
    interpolate,
    interpolation_matrix,
    interpolation_normal_matrix,
    mipmap_grid_sample,
    rasterize,
    rasterize_with_depth,
    render,
    transform,
    transform_with_v_cam,
)

from . import utils  # noqa: F401

# `import drtk.rasterize`, `from drtk.screen_space_uv_derivative import ...`: the reference's submodule names
# resolve to the drtk_amd modules of the same name (as in the reference, the attribute `drtk.rasterize` is the
# FUNCTION -- the `from drtk_amd import ...` above -- while the module is reachable through the import system).
for _name in ("edge_grad_estimator", "interpolate", "mipmap_grid_sample", "rasterize", "render",
              "screen_space_uv_derivative", "transform"):
    _sys.modules[f"{__name__}.{_name}"] = _sys.modules[f"drtk_amd.{_name}"]
del _name

__version__ = _impl.__version__

# the reference's __init__ exports, restricted to the path; the rest -> __getattr__
__all__ = [
    "utils", "edge_grad_estimator", "interpolate", "interpolation_matrix", "interpolation_normal_matrix",
    "mipmap_grid_sample", "rasterize", "rasterize_with_depth", "render", "transform", "transform_with_v_cam",
]

_OUT_OF_SCOPE = {
    "edge_grad_estimator_ref", "interpolate_ref", "render_ref", "mipmap_grid_sample_ref", "grid_scatter_ref",
    "grid_scatter", "msi", "downsample", "filter", "FilterOptions", "FilterType", "low_pass_filter",
    "make_resampling_kernel", "resample_filter", "upsample",
}


def __getattr__(name):
    if name in _OUT_OF_SCOPE:
        raise AttributeError(
            f"drtk.{name} is not provided: this `drtk` is drtk_amd's drop-in for the rasterize -> render -> interpolate "
            "-> edge_grad path of facebookresearch/DRTK (plus transform, the sparse interpolation operators, "
            "mipmap_grid_sample and screen_space_uv_derivative); grid_scatter, msi, filter2d and the pure-PyTorch *_ref "
            "models are outside it (DESIGN.md, out of scope)")
    raise AttributeError(f"module 'drtk' has no attribute '{name}'")

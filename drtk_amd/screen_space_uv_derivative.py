"""`screen_space_uv_derivative` -- host-side mirror of drtk/screen_space_uv_derivative.py:15-80."""
from typing import Optional, Sequence

import torch as th
from drtk_amd.utils import load_torch_ops

load_torch_ops("drtk.interpolate_ext")


def screen_space_uv_derivative(
    v: th.Tensor,
    vt: th.Tensor,
    vi: th.Tensor,
    vti: th.Tensor,
    index_img: th.Tensor,
    bary_img: th.Tensor,
    mask: th.Tensor,
    campos: th.Tensor,
    camrot: th.Tensor,
    focal: th.Tensor,
    dist_mode: Optional[Sequence[str]] = None,
    dist_coeff: Optional[th.Tensor] = None,
) -> th.Tensor:
    """Per-pixel derivative of the uv coordinates wrt the pixel position: `vt_dxdy_img [N, H, W, 2, 2]`
    with Jacobians `[[du/dx, dv/dx], [du/dy, dv/dy]]`, the footprint input of `mipmap_grid_sample`.

    Same arguments as the reference (world-space `v [N,V,3]`, `vt [N,T,2]`, `vi`/`vti [F,3]`, the
    rasterizer's `index_img` / `bary_img`, a bool `mask [N,H,W]`, pinhole camera `campos [N,3]`,
    `camrot [N,3,3]`, `focal [N,2,2]`).  The reference composes it from face_dpdt, two interpolate calls,
    project_points_grad and a batched 2x2 inverse; here it is one kernel.  Differences: pixels with
    `index_img == -1` are written 0 even when `mask` is set there (the reference inverts the background
    sweep of `interpolate` at such pixels), and the result is not differentiable (its consumer defines no
    gradient for it).  Distortion models are not supported, exactly as in `project_points_grad`.
    """
    if dist_mode is not None:
        raise NotImplementedError("screen_space_uv_derivative(): only the undistorted pinhole projection is implemented")
    if th.is_grad_enabled() and any(t.requires_grad for t in (v, vt, campos, camrot, focal)):
        # The reference's composite would carry gradients from vt_dxdy_img back to these; the kernel is forward only.
        # Its one consumer on the path, mipmap_grid_sample, defines no gradient for vt_dxdy_img (mipmap_grid_sampler_
        # module.cpp backward returns none for it), so a rendering loss is unaffected -- anything else that
        # differentiates through the Jacobians gets zeros, hence said once (Python shows a warning once per call site).
        import warnings

        warnings.warn("drtk_amd.screen_space_uv_derivative is not differentiable: vt_dxdy_img carries no gradient to v, vt or the "
                      "camera (harmless in front of mipmap_grid_sample, which defines none for it)", stacklevel=2)
    with th.no_grad():
        return th.ops.drtk_amd_ext.screen_space_uv_derivative(
            v, vt, vi.int(), vti.int(), index_img, bary_img, mask, campos, camrot, focal
        )

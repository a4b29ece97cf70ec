"""`screen_space_uv_derivative` -- host-side mirror of drtk/screen_space_uv_derivative.py:15-80."""
from typing import Optional, Sequence

import torch as th
from drtk_amd.utils import load_torch_ops

load_torch_ops("drtk.interpolate_ext")


class _ForwardOnly(th.autograd.Function):
    """Attaches `out` to the graph of the tensors it was computed from without defining a gradient: a backward pass
    that delivers NO gradient for it (mipmap_grid_sample, its one consumer on the path, defines none for vt_dxdy_img)
    passes through untouched; one that does deliver a gradient raises -- as the reference does: its composite ends with
    an in-place mask on the output of linalg.inv_ex (drtk/screen_space_uv_derivative.py:79), so `backward()` through it
    fails with "modified by an inplace operation" (tests/golden/refpy_uv_derivative_autograd.npz holds the message)."""

    @staticmethod
    def forward(ctx, out, *sources):
        ctx.set_materialize_grads(False)
        ctx.n_inputs = 1 + len(sources)
        return out.view_as(out)

    @staticmethod
    def backward(ctx, g):
        if g is not None:
            raise RuntimeError(
                "screen_space_uv_derivative(): vt_dxdy_img is not differentiable -- the reference's composite cannot be "
                "differentiated either (it masks the output of linalg.inv_ex in place, drtk/screen_space_uv_derivative.py:79, and "
                "backward() raises); use it as the footprint input of mipmap_grid_sample, which defines no gradient for it")
        return (None,) * ctx.n_inputs


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
    sweep of `interpolate` at such pixels).  Autograd: forward only, as in effect upstream -- the reference's composite
    fails in `backward()` (see `_ForwardOnly`); a gradient that reaches the result raises here too, a loss that only
    feeds it to `mipmap_grid_sample` never sends one.  Distortion models are not supported, exactly as in
    `project_points_grad`.
    """
    if dist_mode is not None:
        raise NotImplementedError("screen_space_uv_derivative(): only the undistorted pinhole projection is implemented")
    with th.no_grad():
        out = th.ops.drtk_amd_ext.screen_space_uv_derivative(
            v, vt, vi.int(), vti.int(), index_img, bary_img, mask, campos, camrot, focal
        )
    sources = [t for t in (v, vt, bary_img, campos, camrot, focal) if t.requires_grad]
    if th.is_grad_enabled() and sources:
        # like the reference's result, the Jacobians are part of the autograd graph (requires_grad is True, a loss that
        # never routes a gradient into them is unaffected) -- and like there, a gradient that does arrive is an error
        return _ForwardOnly.apply(out, *sources)
    return out

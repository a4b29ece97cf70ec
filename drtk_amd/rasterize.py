"""`rasterize` / `rasterize_with_depth` -- host-side mirror of drtk/rasterize.py:17-103."""
from typing import Tuple

import torch as th
from drtk_amd.utils import load_torch_ops

load_torch_ops("drtk.rasterize_ext")


def _batched_vi(vi: th.Tensor, n: int) -> th.Tensor:
    # [F,3] topology is shared by all views through a stride-0 expand (drtk/rasterize.py:61-62);
    # the native side keeps it un-materialised.
    return vi[None].expand(n, -1, -1) if vi.ndim == 2 else vi


@th.compiler.disable
def rasterize(
    v: th.Tensor,
    vi: th.Tensor,
    height: int,
    width: int,
    wireframe: bool = False,
) -> th.Tensor:
    """Z-buffer rasterization of a triangle mesh.

    Args:
        v: `[N, V, 3]` pixel-space vertices: x, y on the image plane (pixel centres at integer
            coordinates, the image spans `[-0.5, width-0.5] x [-0.5, height-0.5]`), z = camera
            space depth.  float32 or float64.
        vi: `[F, 3]` or `[N, F, 3]` int32 triangle list.  The top nibble of `vi[..., 0]` is
            ignored in triangle mode; in wireframe mode its bits 0..2 switch the edges
            (v0,v1), (v1,v2), (v0,v2) on (which limits the vertex count to 268435455).
        height, width: image size in pixels.
        wireframe: rasterize the enabled edges (diamond-exit rule) instead of the triangles; the
            triangles still occlude, with index -1.

    Returns:
        `index_img [N, H, W]` int32: id of the nearest triangle covering each pixel centre, `-1`
        where there is none.  Not differentiable; use `edge_grad_estimator` for gradients.
    """
    _, index_img = th.ops.rasterize_ext.rasterize(v, _batched_vi(vi, v.shape[0]), height, width, wireframe)
    return index_img


@th.compiler.disable
def rasterize_with_depth(
    v: th.Tensor,
    vi: th.Tensor,
    height: int,
    width: int,
    wireframe: bool = False,
) -> Tuple[th.Tensor, th.Tensor]:
    """Like :func:`rasterize` but also returns the (non-differentiable, always float32) z-buffer
    depth; empty pixels hold 0.  Returns `(depth_img, index_img)` (drtk/rasterize.py:68-103)."""
    depth_img, index_img = th.ops.rasterize_ext.rasterize(
        v, _batched_vi(vi, v.shape[0]), height, width, wireframe
    )
    return depth_img, index_img

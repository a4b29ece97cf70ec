"""`render` -- host-side mirror of drtk/render.py:18-39."""
from typing import Tuple

import torch as th
from drtk_amd.utils import load_torch_ops

load_torch_ops("drtk.render_ext")


@th.compiler.disable
def render(v: th.Tensor, vi: th.Tensor, index_img: th.Tensor) -> Tuple[th.Tensor, th.Tensor]:
    """Differentiable depth and perspective-correct barycentrics for an index image.

    Args:
        v: `[N, V, 3]` pixel-space vertices (as for `rasterize`).
        vi: `[F, 3]` or `[N, F, 3]` int32 triangles.
        index_img: `[N, H, W]` int32 from `rasterize`.

    Returns:
        `depth_img [N, H, W]` and `bary_img [N, 3, H, W]` (planar), both zero where
        `index_img == -1`; gradients flow to `v`.
    """
    if vi.ndim == 2:
        vi = vi[None].expand(v.shape[0], -1, -1)
    depth_img, bary_img = th.ops.render_ext.render(v, vi, index_img)
    return depth_img, bary_img

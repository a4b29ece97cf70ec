"""`edge_grad_estimator` -- host-side mirror of drtk/edge_grad_estimator.py:20-180."""
from typing import Callable, Optional

import torch as th
from drtk_amd.interpolate import interpolate
from drtk_amd.utils import load_torch_ops

load_torch_ops("drtk.edge_grad_ext")


@th.compiler.disable
def edge_grad_estimator(
    v_pix: th.Tensor,
    vi: th.Tensor,
    bary_img: th.Tensor,
    img: th.Tensor,
    index_img: th.Tensor,
    v_pix_img_hook: Optional[Callable[[th.Tensor], None]] = None,
    max_dp_dr: float = 1e4,
) -> th.Tensor:
    """Adds gradients at visibility discontinuities (arXiv 2405.02508) to a rasterized image.

    Returns `img` unchanged in value; in backward the per-pixel edge gradients
    (`grad_v_pix_img [N, 3, H, W]`) are routed to `v_pix` through the backward of a C=3
    `interpolate(v_pix, ...)` whose forward value is never read (reference graph,
    drtk/edge_grad_estimator.py:168-179).  When no `v_pix_img_hook` is given the two backward stages
    run fused and `grad_v_pix_img` is never materialised (identical results).

    Args:
        v_pix: `[N, V, 3]` pixel-space vertices with camera-space z.
        vi: `[F, 3]` or `[N, F, 3]` int32.
        bary_img: `[N, 3, H, W]`; detached internally.
        img: `[N, C, H, W]` rendered image; must correspond pixel-for-pixel to `index_img`
            (no blur / warp / masking holes before this call).
        index_img: `[N, H, W]` int32.
        v_pix_img_hook: optional backward hook registered on the internal `v_pix_img`.
        max_dp_dr: clamp on |dp/dr| at triangle intersections; 0 disables it.
    """
    if vi.ndim == 2:
        vi = vi[None, ...].expand(v_pix.shape[0], -1, -1)

    if v_pix_img_hook is None:
        # Default route: the reference's own TODO (drtk/edge_grad_estimator.py:168-171) -- skip the
        # C=3 interpolate whose value is never read and scatter the edge gradients straight to
        # v_pix in backward.  Same forward value, same gradients.
        return th.ops.edge_grad_ext.edge_grad_estimator_fused(
            v_pix, vi, bary_img.detach(), img, index_img, max_dp_dr
        )

    # With a hook the intermediate v_pix_img must exist: the reference's graph, op for op.
    v_pix_img = interpolate(v_pix, vi, index_img, bary_img.detach())
    img = th.ops.edge_grad_ext.edge_grad_estimator(v_pix, v_pix_img, vi, img, index_img, max_dp_dr)
    v_pix_img.register_hook(v_pix_img_hook)
    return img

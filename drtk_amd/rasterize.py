"""`rasterize` / `rasterize_with_depth` -- host-side mirror of drtk/rasterize.py:17-103."""
import ctypes
from typing import Tuple

import torch as th
from drtk_amd.utils import load_torch_ops
from drtk_amd.utils.load_torch_ops import native_library_paths

load_torch_ops("drtk.rasterize_ext")

_DEPTH_ORDERS = {"strict": 0, "fastmath": 1}  # drtk_depth_order_t (include/drtk_amd.h)


def set_depth_order(order: str) -> None:
    """Which order the rasterizer sums the depth in, for every later call of this process -- `drtk_amd.rasterize`,
    `drtk.rasterize`, `torch.ops.rasterize_ext.rasterize` and the C ABI alike (they share one library and its one
    setting, include/drtk_amd.h: drtk_amd_set_depth_order).

    `"strict"` (default): the order the reference's source spells (rasterize_kernel.cu:148-153), IEEE -- what a strict
    build of the reference computes.  `"fastmath"`: the order of the reference AS BUILT by its own setup.py:22-24
    (`-O3 --fast-math`): the last bit of ~40 % of the depths moves and with it the owner of a handful of exact near-tie
    pixels per 2048^2 view; `index_img` then equals that build's bit for bit (tests/golden/fastmath_owner_changes_*.npz).
    `DRTK_AMD_DEPTH_ORDER=fastmath` in the environment selects it without a code change."""
    if order not in _DEPTH_ORDERS:
        raise ValueError(f"depth order must be one of {sorted(_DEPTH_ORDERS)}, got {order!r}")
    status = ctypes.CDLL(native_library_paths()[0]).drtk_amd_set_depth_order(_DEPTH_ORDERS[order])
    if status != 0:
        raise RuntimeError(f"drtk_amd_set_depth_order: status {status}")


def get_depth_order() -> str:
    """The current setting: `"strict"` or `"fastmath"` (see :func:`set_depth_order`)."""
    o = ctypes.CDLL(native_library_paths()[0]).drtk_amd_get_depth_order()
    return next(k for k, v in _DEPTH_ORDERS.items() if v == o)


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

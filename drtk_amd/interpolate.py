"""`interpolate` -- host-side mirror of drtk/interpolate.py:20-50."""
import torch as th
from drtk_amd.utils import load_torch_ops

load_torch_ops("drtk.interpolate_ext")


@th.compiler.disable
def interpolate(
    vert_attributes: th.Tensor,
    vi: th.Tensor,
    index_img: th.Tensor,
    bary_img: th.Tensor,
) -> th.Tensor:
    """Barycentric interpolation of per-vertex attributes over the rasterized fragments.

    Args:
        vert_attributes: `[N, V, C]`.
        vi: `[F, 3]` or `[N, F, 3]` int32.
        index_img: `[N, H, W]` int32.
        bary_img: `[N, 3, H, W]`, same dtype as `vert_attributes`.

    Returns:
        `[N, C, H, W]`.  Pixels with `index_img == -1` hold a +-1 coordinate sweep (even channels:
        x, odd: y), exactly like the reference -- mask them out before use.  Gradients flow to
        `vert_attributes` and `bary_img`.
    """
    if vi.ndim == 2:
        vi = vi[None].expand(vert_attributes.shape[0], -1, -1)
    return th.ops.interpolate_ext.interpolate(vert_attributes, vi, index_img, bary_img)

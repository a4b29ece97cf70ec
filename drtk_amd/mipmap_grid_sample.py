"""`mipmap_grid_sample` -- host-side mirror of drtk/mipmap_grid_sample.py:17-127."""
from typing import List, Optional

import torch as th
from drtk_amd.utils import load_torch_ops

load_torch_ops("drtk.mipmap_grid_sampler_ext")


@th.compiler.disable
def mipmap_grid_sample(
    input: List[th.Tensor],
    grid: th.Tensor,
    vt_dxdy_img: th.Tensor,
    max_aniso: int,
    mode: str = "bilinear",
    padding_mode: str = "zeros",
    align_corners: Optional[bool] = None,
    force_max_aniso: Optional[bool] = False,
    clip_grad: Optional[bool] = False,
) -> th.Tensor:
    """`torch.nn.functional.grid_sample` with mipmapping and anisotropic filtering, mimicking graphics
    hardware (OpenGL core profile §8.14): per output pixel the uv Jacobian picks the two nearest mip
    levels (trilinear blend) and up to `max_aniso` taps spread along the major axis of the footprint.

    Args:
        input: mip pyramid, finest level first: `[N, C, H_in, W_in]`, `[N, C, H_in/2, W_in/2]`, ...
            (1 to 11 levels; missing coarse levels are allowed, see `clip_grad`).
        grid: uv field `[N, H_out, W_out, 2]` in `[-1, 1]`.
        vt_dxdy_img: Jacobian of uv (in 0..1 units) wrt the pixel position, `[N, H_out, W_out, 2, 2]`,
            `[[du/dx, dv/dx], [du/dy, dv/dy]]`.
        max_aniso: maximum number of anisotropic taps.
        mode: `'bilinear'` | `'bicubic'`.
        padding_mode: `'zeros'` | `'border'` | `'reflection'`.
        align_corners: as in `grid_sample` (default False).  NB: like the reference's kernel, the
            forward pass always samples with `align_corners=False`; only the backward pass honours it.
        force_max_aniso: always take `max_aniso` taps (debugging / comparison with the PyTorch model).
        clip_grad: when the footprint needs a level beyond the pyramid, shrink the tap spacing to the
            coarsest available level instead of sampling it sparsely.

    Returns:
        `[N, C, H_out, W_out]`.  Gradients flow to every level of `input` and to `grid`.
    """
    if mode != "bilinear" and mode != "bicubic":
        raise ValueError(
            "mipmap_grid_sample(): only 'bilinear' and 'bicubic' modes are supported " "but got: '{}'".format(mode)
        )
    if padding_mode != "zeros" and padding_mode != "border" and padding_mode != "reflection":
        raise ValueError(
            "mipmap_grid_sample(): expected padding_mode "
            "to be 'zeros', 'border', or 'reflection', "
            "but got: '{}'".format(padding_mode)
        )
    mode_enum = 0 if mode == "bilinear" else 2
    padding_mode_enum = {"zeros": 0, "border": 1, "reflection": 2}[padding_mode]
    if align_corners is None:
        align_corners = False
    return th.ops.mipmap_grid_sampler_ext.mipmap_grid_sampler_2d(
        input,
        grid,
        vt_dxdy_img,
        max_aniso,
        padding_mode_enum,
        mode_enum,
        align_corners,
        force_max_aniso,
        clip_grad,
    )

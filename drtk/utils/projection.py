"""drtk.utils.projection of the drop-in: the pinhole branch of `project_points` (drtk/utils/projection.py:33-53,
486-570) -- what `drtk.transform` is made of.  Distortion modes raise NotImplementedError (out of scope)."""
from typing import List, Optional, Tuple, Union

import torch as th

from drtk_amd.transform import project_pinhole, transform_with_v_cam  # noqa: F401

DISTORTION_MODES = {None, "pinhole"}  # the reference also has radial-tangential, fisheye, fisheye62(_lut): not built


def project_points(
    v: th.Tensor,
    campos: th.Tensor,
    camrot: th.Tensor,
    focal: th.Tensor,
    princpt: th.Tensor,
    distortion_mode: Optional[Union[List[str], str]] = None,
    distortion_coeff: Optional[th.Tensor] = None,
    fov: Optional[th.Tensor] = None,
    lut_vector_field: Optional[th.Tensor] = None,
    lut_spacing: Optional[th.Tensor] = None,
) -> Tuple[th.Tensor, th.Tensor]:
    """`(v_pix, v_cam)`, both `[N,V,3]`; `v_cam = camrot @ (v - campos)`, `v_pix = (x_pix, y_pix, z_cam)`."""
    return transform_with_v_cam(v, campos, camrot, focal, princpt, None, None, distortion_mode, distortion_coeff, fov,
                                lut_vector_field, lut_spacing)

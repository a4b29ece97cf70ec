"""Pinhole `transform` -- the step before the hot path (drtk/transform.py:13-119,
drtk/utils/projection.py:33-53,486-540).  Same signature as the reference; only the undistorted pinhole
model is provided (`distortion_mode` None / "pinhole" / a list of those) -- the distortion models are
outside the hot-path scope and raise NotImplementedError instead of computing something else.  On a HIP device, when the camera parameters do
not require gradients, `transform` runs as ONE fused kernel each way (`drtk_amd_ext::transform_pinhole`,
csrc/transform.hip); otherwise the PyTorch formulation below is used."""
from typing import List, Optional, Tuple, Union

import torch as th


def _require_pinhole(distortion_mode, distortion_coeff, fov=None, lut_vector_field=None, lut_spacing=None) -> None:
    """The reference's mode handling (utils/projection.py:537-561) restricted to what is built here."""
    if distortion_mode is not None:
        assert distortion_coeff is not None, "Missing distortion coefficients."
    modes = set(distortion_mode) if isinstance(distortion_mode, (list, tuple)) else {distortion_mode}
    if not modes <= {None, "pinhole"}:
        raise NotImplementedError(
            f"drtk_amd.transform implements the pinhole camera only; distortion_mode={distortion_mode!r} "
            "(radial-tangential / fisheye / fisheye62 of drtk.utils.projection) is outside the rasterize -> render -> "
            "interpolate -> edge_grad path this package rebuilds (DESIGN.md, out of scope)")


def project_pinhole(v_cam: th.Tensor, focal: th.Tensor, princpt: th.Tensor) -> th.Tensor:
    z = v_cam[:, :, 2:3]
    z = th.where(z < 0, z.clamp(max=-1e-8), z.clamp(min=1e-8))
    v_proj = v_cam[:, :, 0:2] / z
    # focal @ v_proj per vertex, written as ONE [N,V,2]x[N,2,2] batched product: the reference's
    # per-vertex `focal[:, None] @ v_proj[..., None]` launches a degenerate N*V-batch GEMM that
    # takes milliseconds on ROCm.
    return th.bmm(v_proj, focal.transpose(1, 2)) + princpt[:, None]


def transform_with_v_cam(
    v: th.Tensor,
    campos: Optional[th.Tensor] = None,
    camrot: Optional[th.Tensor] = None,
    focal: Optional[th.Tensor] = None,
    princpt: Optional[th.Tensor] = None,
    K: Optional[th.Tensor] = None,
    Rt: Optional[th.Tensor] = None,
    distortion_mode: Optional[Union[List[str], str]] = None,
    distortion_coeff: Optional[th.Tensor] = None,
    fov: Optional[th.Tensor] = None,
    lut_vector_field: Optional[th.Tensor] = None,
    lut_spacing: Optional[th.Tensor] = None,
) -> Tuple[th.Tensor, th.Tensor]:
    if not ((camrot is not None and campos is not None) ^ (Rt is not None)):
        raise ValueError("You must provide exactly one of Rt or (campos, camrot).")
    if not ((focal is not None and princpt is not None) ^ (K is not None)):
        raise ValueError("You must provide exactly one of K or (focal, princpt).")
    if campos is None:
        camrot = Rt[:, :3, :3]
        campos = -(camrot.transpose(-2, -1) @ Rt[:, :3, 3:4])[..., 0]
    if focal is None:
        focal = K[:, :2, :2]
        princpt = K[:, :2, 2]
    _require_pinhole(distortion_mode, distortion_coeff, fov, lut_vector_field, lut_spacing)
    # camrot @ (v - campos) per vertex as one [N,V,3]x[N,3,3] batched product (see project_pinhole)
    v_cam = th.bmm(v - campos[:, None], camrot.transpose(1, 2))
    v_pix = project_pinhole(v_cam, focal, princpt)
    return th.cat((v_pix, v_cam[:, :, 2:3]), dim=-1), v_cam


def transform(
    v: th.Tensor,
    campos: Optional[th.Tensor] = None,
    camrot: Optional[th.Tensor] = None,
    focal: Optional[th.Tensor] = None,
    princpt: Optional[th.Tensor] = None,
    K: Optional[th.Tensor] = None,
    Rt: Optional[th.Tensor] = None,
    distortion_mode: Optional[Union[List[str], str]] = None,
    distortion_coeff: Optional[th.Tensor] = None,
    fov: Optional[th.Tensor] = None,
) -> th.Tensor:
    """World space `[N,V,3]` (or one shared `[1,V,3]`) -> `(x_pix, y_pix, z_cam)`;
    `v_cam = camrot @ (v - campos)`.  Signature of drtk/transform.py:13-24."""
    if v.is_cuda and v.dtype in (th.float32, th.float64):
        if not ((camrot is not None and campos is not None) ^ (Rt is not None)):
            raise ValueError("You must provide exactly one of Rt or (campos, camrot).")
        if not ((focal is not None and princpt is not None) ^ (K is not None)):
            raise ValueError("You must provide exactly one of K or (focal, princpt).")
        if campos is None:
            camrot = Rt[:, :3, :3]
            campos = -(camrot.transpose(-2, -1) @ Rt[:, :3, 3:4])[..., 0]
        if focal is None:
            focal = K[:, :2, :2]
            princpt = K[:, :2, 2]
        _require_pinhole(distortion_mode, distortion_coeff, fov)
        cams = (campos, camrot, focal, princpt)
        if not (th.is_grad_enabled() and any(c.requires_grad for c in cams)):
            from drtk_amd.utils import load_torch_ops

            load_torch_ops("drtk.rasterize_ext")
            if v.shape[0] != 1 and v.stride(0) == 0:
                v = v[:1]  # expanded world-space vertices: keep them shared, the kernel broadcasts
            return th.ops.drtk_amd_ext.transform_pinhole(v, *cams)
        return transform_with_v_cam(v, *cams)[0]
    return transform_with_v_cam(v, campos, camrot, focal, princpt, K, Rt, distortion_mode, distortion_coeff, fov)[0]

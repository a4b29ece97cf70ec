"""Deterministic synthetic scenes for benchmarks, smoke tests and parity tests.

Shapes follow BASELINE.md / SURVEY.md §8d: UV spheres of `n_lat x n_lon` quads
(F = 2*n_lat*n_lon, V = (n_lat+1)*n_lon; pole triangles are zero-area on purpose -- they exercise
culling), N cameras on a ring at distance 3 with focal = 1.2*W, principal point at the image
centre (sphere covers ~57 % of the frame, z in [2, 4]).  Everything is generated with torch on the
requested device; no RNG except where a generator seed is passed explicitly.
"""
import math
from typing import Optional, Tuple

import torch as th

from drtk_amd.transform import transform

# (n_lat, n_lon) giving the triangle counts named in BASELINE.json
MESH_SIZES = {
    "10k": (70, 72),     # F = 10 080
    "100k": (224, 224),  # F = 100 352
    "250k": (354, 354),  # F = 250 632
    "1M": (708, 708),    # F = 1 002 528
}


def uv_sphere(
    n_lat: int,
    n_lon: int,
    radius: float = 1.0,
    lobes: float = 0.0,
    device="cpu",
    dtype=th.float32,
) -> Tuple[th.Tensor, th.Tensor]:
    """Returns `(v [V,3] world space, vi [F,3] int32)`.  `lobes > 0` displaces the radius by
    `1 + lobes * (sin(3 theta) cos(2 phi) + 0.5 cos(5 theta) sin(3 phi))` ("head"/"body" variants)."""
    i = th.arange(n_lat + 1, dtype=th.float64)
    j = th.arange(n_lon, dtype=th.float64)
    theta = (math.pi * i / n_lat)[:, None].expand(-1, n_lon)
    phi = (2.0 * math.pi * j / n_lon)[None, :].expand(n_lat + 1, -1)
    r = th.full_like(theta, radius)
    if lobes != 0.0:
        r = r * (1.0 + lobes * (th.sin(3 * theta) * th.cos(2 * phi) + 0.5 * th.cos(5 * theta) * th.sin(3 * phi)))
    x = r * th.sin(theta) * th.cos(phi)
    y = r * th.cos(theta)
    z = r * th.sin(theta) * th.sin(phi)
    v = th.stack([x, y, z], dim=-1).reshape(-1, 3)

    ii = th.arange(n_lat)[:, None].expand(-1, n_lon)
    jj = th.arange(n_lon)[None, :].expand(n_lat, -1)
    jn = (jj + 1) % n_lon
    v00 = ii * n_lon + jj
    v01 = ii * n_lon + jn
    v10 = (ii + 1) * n_lon + jj
    v11 = (ii + 1) * n_lon + jn
    t0 = th.stack([v00, v10, v11], dim=-1)
    t1 = th.stack([v00, v11, v01], dim=-1)
    vi = th.stack([t0, t1], dim=2).reshape(-1, 3).to(th.int32)
    return v.to(dtype).to(device), vi.to(device)


def ring_cameras(n: int, width: int, height: int, distance: float = 3.0, device="cpu", dtype=th.float32):
    """N cameras on a ring around the origin: `camrot = R_y(2 pi k / N + 0.1)`, looking at the
    origin from `distance`.  Returns `(campos [N,3], camrot [N,3,3], focal [N,2,2], princpt [N,2])`."""
    k = th.arange(n, dtype=th.float64)
    a = 2.0 * math.pi * k / n + 0.1
    c, s = th.cos(a), th.sin(a)
    zero, one = th.zeros_like(a), th.ones_like(a)
    camrot = th.stack(
        [th.stack([c, zero, s], -1), th.stack([zero, one, zero], -1), th.stack([-s, zero, c], -1)], dim=1
    )
    # v_cam = R (v - campos); the origin must land at (0, 0, distance)
    campos = -(camrot.transpose(1, 2) @ th.tensor([0.0, 0.0, distance], dtype=th.float64)[None, :, None])[..., 0]
    focal = th.eye(2, dtype=th.float64)[None].repeat(n, 1, 1) * (1.2 * width)
    princpt = th.tensor([width / 2.0, height / 2.0], dtype=th.float64)[None].repeat(n, 1)
    return tuple(t.to(dtype).to(device) for t in (campos, camrot, focal, princpt))


def sphere_views(
    n_views: int,
    n_lat: int,
    n_lon: int,
    height: int,
    width: int,
    lobes: float = 0.0,
    second_sphere: bool = False,
    device="cpu",
    dtype=th.float32,
):
    """Projected sphere scene: returns `(v_pix [N,V,3], vi [F,3] int32)`.

    `second_sphere=True` appends an interpenetrating copy offset by (+40.3/512*W px, +0.11 z) to
    create occlusion and intersection edges (SURVEY.md §8d "parity scene add-on")."""
    v, vi = uv_sphere(n_lat, n_lon, lobes=lobes, device=device, dtype=dtype)
    campos, camrot, focal, princpt = ring_cameras(n_views, width, height, device=device, dtype=dtype)
    v_pix = transform(v[None].expand(n_views, -1, -1), campos, camrot, focal, princpt)
    if second_sphere:
        off = th.tensor([40.3 * width / 512.0, 0.0, 0.11], dtype=dtype, device=device)
        v_pix = th.cat([v_pix, v_pix + off], dim=1)
        vi = th.cat([vi, vi + v.shape[0]], dim=0)
    return v_pix.contiguous(), vi.contiguous()


def two_triangles(height: int = 64, width: int = 64, device="cpu", dtype=th.float32):
    """The scene of the reference's test/two_triangles.py:17-38, scaled from 512x512 to
    `width x height`: returns `(v [1,6,3], vi [2,3] int32, vt [1,6,2], tex [1,3,16,16])`."""
    v = th.tensor(
        [[10, 200, 100], [300, 50, 100], [400, 500, 100], [50, 400, 200], [400, 50, 50], [300, 500, 200]],
        dtype=th.float64,
    )
    v[:, 0] *= width / 512.0
    v[:, 1] *= height / 512.0
    vt = th.zeros(1, 6, 2, dtype=th.float64)
    vt[:, 3:6, 0] = 1
    vi = th.arange(6, dtype=th.int32).view(2, 3)
    tex = th.ones(1, 3, 16, 16, dtype=th.float64)
    tex[:, :, :, 8:] = 0.5
    return v[None].to(dtype).to(device), vi.to(device), vt.to(dtype).to(device), tex.to(dtype).to(device)


def random_attributes(n_views: int, n_vertices: int, channels: int, seed: int = 0, device="cpu", dtype=th.float32,
                      shared: bool = True) -> th.Tensor:
    """`attr ~ U[0,1)` of shape `[1,V,C]` expanded to N (shared across views) or `[N,V,C]`."""
    g = th.Generator(device="cpu")
    g.manual_seed(seed)
    if shared:
        a = th.rand(1, n_vertices, channels, generator=g, dtype=th.float32).to(dtype).to(device)
        return a.expand(n_views, -1, -1)
    return th.rand(n_views, n_vertices, channels, generator=g, dtype=th.float32).to(dtype).to(device)


def fwd_bwd_step(v_pix: th.Tensor, vi: th.Tensor, attr: th.Tensor, height: int, width: int,
                 ops=None, max_dp_dr: float = 1e4):
    """One full hot-path step as defined in SURVEY.md §8d:
    rasterize -> render -> interpolate(attr) -> mask -> edge_grad_estimator -> loss -> backward.
    `v_pix` and `attr` must be leaf tensors requiring grad.  Returns (loss, index_img)."""
    if ops is None:
        import drtk_amd as ops
    index_img = ops.rasterize(v_pix, vi, height, width)
    depth_img, bary_img = ops.render(v_pix, vi, index_img)
    img = ops.interpolate(attr, vi, index_img, bary_img)
    img = img * (index_img != -1)[:, None]
    img = ops.edge_grad_estimator(v_pix=v_pix, vi=vi, bary_img=bary_img, img=img, index_img=index_img,
                                  max_dp_dr=max_dp_dr)
    loss = (img * img).mean() + depth_img.mean()
    loss.backward()
    return loss, index_img


def uv_sphere_atlas(n_lat: int, n_lon: int, device="cpu", dtype=th.float32) -> Tuple[th.Tensor, th.Tensor]:
    """Latitude / longitude texture atlas of `uv_sphere(n_lat, n_lon)`: `(vt [(n_lat+1)*(n_lon+1), 2]` in `[0,1]`,
    `vti [F,3] int32)`.  The atlas has its own topology -- one more column of uv vertices than the mesh has, so that
    the last column of quads runs to u = 1 instead of wrapping back to u = 0 (no seam triangles spanning the whole
    texture); faces are in the order of `uv_sphere`'s `vi`."""
    i = th.arange(n_lat + 1, dtype=th.float64)
    j = th.arange(n_lon + 1, dtype=th.float64)
    # keep the atlas a little inside the texture so that the outermost taps stay off the border handling
    u = (0.02 + 0.96 * j / n_lon)[None, :].expand(n_lat + 1, -1)
    w = (0.02 + 0.96 * i / n_lat)[:, None].expand(-1, n_lon + 1)
    vt = th.stack([u, w], dim=-1).reshape(-1, 2)
    ii = th.arange(n_lat)[:, None].expand(-1, n_lon)
    jj = th.arange(n_lon)[None, :].expand(n_lat, -1)
    s = n_lon + 1
    v00, v01, v10, v11 = ii * s + jj, ii * s + jj + 1, (ii + 1) * s + jj, (ii + 1) * s + jj + 1
    t0 = th.stack([v00, v10, v11], dim=-1)
    t1 = th.stack([v00, v11, v01], dim=-1)
    vti = th.stack([t0, t1], dim=2).reshape(-1, 3).to(th.int32)
    return vt.to(dtype).to(device), vti.to(device)


def texture_pyramid(n: int, channels: int, size: int, seed: int = 1, device="cpu", dtype=th.float32, max_levels: int = 11):
    """`[n, channels, size, size]` of smooth colour fields plus noise, and its box-filtered mip chain (finest first,
    at most `max_levels` levels -- the sampler's limit, mipmap_grid_sampler_kernel.cu:912)."""
    g = th.Generator(device="cpu").manual_seed(seed)
    yy, xx = th.meshgrid(th.linspace(0, 1, size, dtype=th.float32), th.linspace(0, 1, size, dtype=th.float32), indexing="ij")
    base = th.stack([0.5 + 0.5 * th.sin((3 + c) * math.pi * xx + c) * th.cos((2 + c) * math.pi * yy) for c in range(channels)])
    lv = [(0.7 * base[None] + 0.3 * th.rand(n, channels, size, size, generator=g)).to(device)]
    while lv[-1].shape[-1] > 1 and len(lv) < max_levels:
        lv.append(th.nn.functional.avg_pool2d(lv[-1], 2))
    return [t.to(dtype) for t in lv]


def textured_shading(ops, v_world, v_pix, vi, vt, vti, tex, campos, camrot, focal, height, width, max_aniso=8,
                     uv_jacobian=None, mipmap=None):
    """Forward of BASELINE.json configs[4]'s textured pipeline, spelled with the drtk functions of `ops`:
    rasterize -> render -> interpolate(uv) -> screen_space_uv_derivative -> mipmap_grid_sample -> mask ->
    edge_grad_estimator (the reference's usage: drtk/mipmap_grid_sample.py:17-128, drtk/screen_space_uv_derivative.py:
    15-80).  `uv_jacobian` / `mipmap` let a test substitute those two stages.  Returns a dict of the stage outputs."""
    index_img = ops.rasterize(v_pix, vi, height, width)
    depth_img, bary_img = ops.render(v_pix, vi, index_img)
    mask = index_img != -1
    uv_img = ops.interpolate(vt, vti, index_img, bary_img)  # [N,2,H,W]; fp16-stored uv is upcast by the op under autocast
    jac_fn = uv_jacobian if uv_jacobian is not None else ops.screen_space_uv_derivative
    vt_dxdy_img = jac_fn(v_world, vt.float() if vt.dtype == th.float16 else vt, vi, vti, index_img, bary_img, mask, campos, camrot, focal)
    # the background mask spelled with torch.where: the same values as `* mask` for finite inputs, without the bool ->
    # float promotion that sends the product through ATen's slow strided kernel (0.26 ms per call at 2 x 4096^2)
    grid = th.where(mask[:, None], uv_img * 2 - 1, 0.0).permute(0, 2, 3, 1)
    sample = mipmap if mipmap is not None else ops.mipmap_grid_sample
    shaded = sample(tex, grid, vt_dxdy_img, max_aniso, padding_mode="border")
    img = th.where(mask[:, None], shaded, 0.0)
    img = ops.edge_grad_estimator(v_pix=v_pix, vi=vi, bary_img=bary_img, img=img, index_img=index_img)
    return dict(index_img=index_img, depth_img=depth_img, bary_img=bary_img, uv_img=uv_img, vt_dxdy_img=vt_dxdy_img,
                shaded=shaded, img=img)

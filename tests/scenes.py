"""TEST INFRASTRUCTURE -- named deterministic parity scenes (inputs only).

Each builder returns a dict with `v` [N,V,3], `vi` [F,3] or [N,F,3] int32, `H`, `W`, `C` and (seeded)
`attr` [N,V,C] plus O(1) upstream gradients `gd` [N,H,W], `gb` [N,3,H,W], `go` [N,C,H,W] drawn from
U[-1,1] (SURVEY.md §7 hard part 3: make the 1e-5 tolerance bite).  Random numbers are always drawn in
float32 and converted, so f32 and f64 variants of a scene share their values.
"""
import torch as th

from drtk_amd import synthetic as S


def _seeded(n, v_count, c, h, w, seed, dtype):
    g = th.Generator().manual_seed(seed)
    attr = th.rand(n, v_count, c, generator=g, dtype=th.float32).to(dtype)
    gd = (th.rand(n, h, w, generator=g, dtype=th.float32) * 2 - 1).to(dtype)
    gb = (th.rand(n, 3, h, w, generator=g, dtype=th.float32) * 2 - 1).to(dtype)
    go = (th.rand(n, c, h, w, generator=g, dtype=th.float32) * 2 - 1).to(dtype)
    return attr, gd, gb, go


def _pack(v, vi, h, w, c, seed, dtype):
    v = v.to(dtype).contiguous()
    attr, gd, gb, go = _seeded(v.shape[0], v.shape[1], c, h, w, seed, dtype)
    return dict(v=v, vi=vi.contiguous(), H=h, W=w, C=c, attr=attr, gd=gd, gb=gb, go=go)


def two_triangles(dtype=th.float32, perturbed=True):
    """test/two_triangles.py:17-46 scaled to 64x64; perturbation randn*20/8 with a CPU generator."""
    v, vi, _, _ = S.two_triangles(64, 64, dtype=th.float64)
    if perturbed:
        g = th.Generator().manual_seed(10)
        noise = th.randn(v.shape, generator=g, dtype=th.float32).double() * (20.0 / 8.0)
        noise[..., 2] *= 8.0  # z is not rescaled with the image
        v = v + noise
    return _pack(v, vi, 64, 64, 3, 11, dtype)


def tutorial3(dtype=th.float32):
    """Intersecting 4-triangle scene of docs/source/tutorials/DRTK_Tutorial_3 (cell 4), initial
    vertices, scaled from 512x512 to 64x64."""
    v = th.tensor(
        [[12.08, 31.02, 100], [455.8, 71.94, 100], [168.9, 540.1, 100], [260.0, 110.0, 80], [478.0, 110.0, 80],
         [260.0, 235.0, 80], [478, 235.0, 80], [75.85, 386.0, 95], [215.1, 226.1, 95], [378.8, 481.1, 280]],
        dtype=th.float64,
    )
    v[:, :2] *= 64.0 / 512.0
    vi = th.tensor([[0, 1, 2], [3, 4, 5], [6, 4, 5], [7, 8, 9]], dtype=th.int32)
    return _pack(v[None], vi, 64, 64, 4, 12, dtype)


def spheres(dtype=th.float32, n=2, n_lat=20, n_lon=24, h=48, w=64, c=5):
    """Two interpenetrating UV spheres seen from a camera ring: silhouettes, occlusion edges,
    intersection edges, zero-area pole triangles, 2x overdraw (no back-face culling)."""
    v, vi = S.sphere_views(n, n_lat, n_lon, h, w, second_sphere=True, dtype=th.float64)
    return _pack(v, vi, h, w, c, 13, dtype)


def spheres_c16(dtype=th.float32):
    """Same geometry class with the graded 16-channel attribute width and W not a multiple of 4's
    neighbour sizes exercised elsewhere; here W % 4 == 0 (vector path)."""
    return spheres(dtype, n=2, n_lat=16, n_lon=20, h=40, w=56, c=16)


def ragged(dtype=th.float32):
    """W % 4 != 0 (scalar fallback paths), H odd, per-view topology [N,F,3] padded with degenerate
    (all-equal-index) triangles, and (for rasterize) the top nibble of vi[...,0] set on some faces."""
    v, vi = S.sphere_views(2, 10, 12, 37, 45, second_sphere=False, dtype=th.float64)
    f = vi.shape[0]
    vib = vi[None].repeat(2, 1, 1)
    pad = th.zeros(2, 7, 3, dtype=th.int32)
    pad[1] = 5
    vib = th.cat([vib, pad], dim=1)
    d = _pack(v, vib, 37, 45, 3, 14, dtype)
    # top nibble of vi[...,0]: masked by rasterize ONLY (rasterize_kernel.cu:74); render /
    # interpolate / edge_grad read vi raw, so the flagged copy is used for rasterize alone.
    vir = vib.clone()
    vir[0, : f // 2, 0] |= 0x30000000
    d["vi_raster"] = vir
    return d


def edge_cases(dtype=th.float32):
    """Appendix C/E micro-scenes on one 16x16 canvas (one view per case):
      0 coincident triangles, opposite windings, equal depth      -> lower id wins
      1 quad split along the diagonal, pixel-aligned vertices     -> top-left rule, no holes/doubles
      2 one vertex at z = 0 (near plane)                          -> whole triangle culled
      3 fully off-screen + zero-area + (a,a,b) index pattern      -> nothing drawn
      4 huge triangle covering the canvas with far-away vertices  -> clamped bbox
      5 triangle partially outside the canvas (negative coords)
    """
    views = []
    tris = []
    # case 0
    views.append([[2, 2, 2], [12, 3, 2], [4, 13, 2], [2, 2, 2], [4, 13, 2], [12, 3, 2]])
    tris.append([[0, 1, 2], [3, 4, 5]])
    # case 1
    views.append([[0, 0, 1], [6, 0, 1], [6, 6, 1], [0, 6, 1], [0, 0, 1], [0, 0, 1]])
    tris.append([[0, 1, 2], [0, 2, 3]])
    # case 2
    views.append([[1, 1, 1], [9, 1, 0], [1, 9, 1], [2, 2, 1e-9], [9, 3, 1], [3, 9, 1]])
    tris.append([[0, 1, 2], [3, 4, 5]])
    # case 3
    views.append([[20, 20, 1], [30, 20, 1], [20, 30, 1], [1, 1, 1], [5, 5, 1], [9, 9, 1]])
    tris.append([[0, 1, 2], [3, 4, 5]])
    # case 4
    views.append([[-1e4, -1e4, 3], [3e4, -1e4, 3], [-1e4, 3e4, 5], [5, 5, 2], [9, 5, 2], [5, 9, 2]])
    tris.append([[0, 1, 2], [3, 4, 5]])
    # case 5
    views.append([[-5.5, -3.25, 2], [7.75, 2.5, 3], [-2.0, 11.5, 2.5], [12, 12, 1], [18.5, 12.5, 1], [12.25, 19, 2]])
    tris.append([[0, 1, 2], [3, 4, 5]])
    v = th.tensor(views, dtype=th.float64)
    vi = th.tensor(tris, dtype=th.int32)
    vi[3, 1] = th.tensor([3, 3, 4], dtype=th.int32)  # (a,a,b): goes to the area test, not the index test
    return _pack(v, vi, 16, 16, 2, 15, dtype)


SCENES = {
    "two_triangles": two_triangles,
    "tutorial3": tutorial3,
    "spheres": spheres,
    "spheres_c16": spheres_c16,
    "ragged": ragged,
    "edge_cases": edge_cases,
}

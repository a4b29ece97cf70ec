"""TEST INFRASTRUCTURE ONLY -- ctypes front-end of oracle/libdrtk_oracle.so.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.  The
product package (drtk_amd/) never imports this module.

All functions take/return CPU torch tensors (float32 or float64, int32 indices) and mirror the
reference's CPU entry points (file:line in drtk_oracle.h).  `vi` may be [F,3] (shared topology) or
[N,F,3].  `nthreads=1` is the deterministic parity mode; `nthreads=0` uses every core (baseline).
"""
import ctypes
import os

import torch
import torch as th

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdrtk_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.isfile(_SO):
            import importlib.util

            spec = importlib.util.spec_from_file_location("_oracle_build", os.path.join(_HERE, "build.py"))
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            mod.build(verbose=False)
        _lib = ctypes.CDLL(_SO)
        _lib.drtk_oracle_max_threads.restype = ctypes.c_int
    return _lib


def max_threads() -> int:
    return int(lib().drtk_oracle_max_threads())


class accumulated_magnitudes:
    """`with accumulated_magnitudes(): A = render_backward(...)`: inside the block the backward functions (render,
    interpolate, edge_grad) sum |term| instead of term, so each output element is the MAGNITUDE accumulated into it --
    the per-element scale of a float32 evaluation's rounding error (tests/f64_distance.py).  Process-wide switch of the
    oracle library; restored on exit."""

    def __enter__(self):
        lib().drtk_oracle_set_abs_accumulate(1)
        return self

    def __exit__(self, *exc):
        lib().drtk_oracle_set_abs_accumulate(0)
        return False


def _sfx(t: torch.Tensor) -> str:
    if t.dtype == torch.float32:
        return "f32"
    if t.dtype == torch.float64:
        return "f64"
    raise TypeError(f"oracle: unsupported dtype {t.dtype}")


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _i64(x):
    return ctypes.c_int64(int(x))


def _prep_vi(vi: torch.Tensor, N: int):
    assert vi.dtype == torch.int32
    if vi.ndim == 2:
        vi_c = vi.contiguous()
        return vi_c, 0, vi_c.shape[0]
    assert vi.ndim == 3
    if vi.shape[0] == N and vi.stride(0) == 0:
        vi_c = vi[0].contiguous()
        return vi_c, 0, vi_c.shape[0]
    vi_c = vi.contiguous()
    return vi_c, vi_c.shape[1] * 3, vi_c.shape[1]


def _call(name, sfx, *args):
    fn = getattr(lib(), f"drtk_oracle_{name}_{sfx}")
    fn.restype = ctypes.c_int
    rc = fn(*args)
    if rc != 0:
        raise RuntimeError(f"drtk_oracle_{name}_{sfx} failed with code {rc}")


def rasterize(v, vi, height, width, nthreads=1):
    v = v.contiguous()
    N, V, _ = v.shape
    vi_c, vi_sN, F = _prep_vi(vi, N)
    depth = torch.empty(N, height, width, dtype=torch.float32)
    index = torch.empty(N, height, width, dtype=torch.int32)
    _call("rasterize", _sfx(v), _p(v), _p(vi_c), _i64(N), _i64(V), _i64(F), _i64(vi_sN),
          _i64(height), _i64(width), _p(depth), _p(index), ctypes.c_int(nthreads))
    return depth, index


def rasterize_lines(v, vi, height, width):
    """Wireframe mode (rasterize_kernel.cu:170-400 restated; parity unpinned)."""
    v = v.contiguous()
    N, V, _ = v.shape
    vi_c, vi_sN, F = _prep_vi(vi, N)
    depth = torch.empty(N, height, width, dtype=torch.float32)
    index = torch.empty(N, height, width, dtype=torch.int32)
    _call("rasterize_lines", _sfx(v), _p(v), _p(vi_c), _i64(N), _i64(V), _i64(F), _i64(vi_sN), _i64(height),
          _i64(width), _p(depth), _p(index))
    return depth, index


def render(v, vi, index_img, nthreads=1):
    v = v.contiguous()
    index_img = index_img.contiguous()
    N, V, _ = v.shape
    H, W = index_img.shape[1:]
    vi_c, vi_sN, F = _prep_vi(vi, N)
    depth = torch.empty(N, H, W, dtype=v.dtype)
    bary = torch.empty(N, 3, H, W, dtype=v.dtype)
    _call("render", _sfx(v), _p(v), _p(vi_c), _p(index_img), _i64(N), _i64(V), _i64(F),
          _i64(vi_sN), _i64(H), _i64(W), _p(depth), _p(bary), ctypes.c_int(nthreads))
    return depth, bary


def render_backward(v, vi, index_img, grad_depth_img, grad_bary_img, nthreads=1):
    v = v.contiguous()
    index_img = index_img.contiguous()
    gd = grad_depth_img.contiguous()
    gb = grad_bary_img.contiguous()
    N, V, _ = v.shape
    H, W = index_img.shape[1:]
    vi_c, vi_sN, F = _prep_vi(vi, N)
    grad_v = torch.zeros(N, V, 3, dtype=v.dtype)
    _call("render_backward", _sfx(v), _p(v), _p(vi_c), _p(index_img), _p(gd), _p(gb), _i64(N),
          _i64(V), _i64(F), _i64(vi_sN), _i64(H), _i64(W), _p(grad_v), ctypes.c_int(nthreads))
    return grad_v


def interpolate(attrs, vi, index_img, bary_img, nthreads=1):
    attrs = attrs.contiguous()
    index_img = index_img.contiguous()
    bary_img = bary_img.contiguous()
    N, V, C = attrs.shape
    H, W = index_img.shape[1:]
    vi_c, vi_sN, F = _prep_vi(vi, N)
    out = torch.empty(N, C, H, W, dtype=attrs.dtype)
    _call("interpolate", _sfx(attrs), _p(attrs), _p(vi_c), _p(index_img), _p(bary_img), _i64(N),
          _i64(V), _i64(C), _i64(F), _i64(vi_sN), _i64(H), _i64(W), _p(out), ctypes.c_int(nthreads))
    return out


def interpolate_backward(grad_out, attrs, vi, index_img, bary_img, vert_requires_grad=True,
                         bary_requires_grad=True, nthreads=1):
    grad_out = grad_out.contiguous()
    attrs = attrs.contiguous()
    index_img = index_img.contiguous()
    bary_img = bary_img.contiguous()
    N, V, C = attrs.shape
    H, W = index_img.shape[1:]
    vi_c, vi_sN, F = _prep_vi(vi, N)
    attr_grad = torch.zeros(N, V, C, dtype=attrs.dtype) if vert_requires_grad else None
    bary_grad = torch.empty(N, 3, H, W, dtype=attrs.dtype) if bary_requires_grad else None
    _call("interpolate_backward", _sfx(attrs), _p(grad_out), _p(attrs), _p(vi_c), _p(index_img),
          _p(bary_img), _i64(N), _i64(V), _i64(C), _i64(F), _i64(vi_sN), _i64(H), _i64(W),
          _p(attr_grad), _p(bary_grad), ctypes.c_int(nthreads))
    return attr_grad, bary_grad


def edge_grad_backward(v_pix, img, index_img, vi, grad_output, max_dp_dr=1e4, nthreads=1):
    v_pix = v_pix.contiguous()
    img = img.contiguous()
    index_img = index_img.contiguous()
    grad_output = grad_output.contiguous()
    N, V, _ = v_pix.shape
    C, H, W = img.shape[1:]
    vi_c, vi_sN, F = _prep_vi(vi, N)
    g = torch.zeros(N, 3, H, W, dtype=v_pix.dtype)
    _call("edge_grad_backward", _sfx(v_pix), _p(v_pix), _p(img), _p(index_img), _p(vi_c),
          _p(grad_output), _i64(N), _i64(V), _i64(C), _i64(F), _i64(vi_sN), _i64(H), _i64(W),
          ctypes.c_double(max_dp_dr), _p(g), ctypes.c_int(nthreads))
    return g


# ---- sparse interpolation operators (interpolate_kernel_cpu.cpp:411-693, interpolate_module.cpp:167-241)


def _batched_vi(vi, n):
    return vi[None].expand(n, -1, -1) if vi.ndim == 2 else (vi.expand(n, -1, -1) if vi.shape[0] == 1 else vi)


def interpolation_matrix(vi, index_img, bary_img):
    """Returns (crow_indices, col_indices, values, row_pixels) like interpolation_matrix_cpu."""
    index_img = index_img.contiguous()
    bary_img = bary_img.contiguous()
    N, H, W = index_img.shape
    vi_c, vi_sN, F = _prep_vi(vi, N)
    row_pixels = th.nonzero(index_img.reshape(-1).ne(-1)).reshape(-1).contiguous()
    R = row_pixels.numel()
    crow = th.arange(0, R * 3 + 1, 3, dtype=th.int64)
    col = th.empty(R * 3, dtype=th.int64)
    values = th.empty(R * 3, dtype=bary_img.dtype)
    _call("interpolation_matrix", _sfx(bary_img), _p(vi_c), _p(index_img), _p(bary_img), _p(row_pixels), _i64(R),
          _i64(N), _i64(F), _i64(vi_sN), _i64(H), _i64(W), _p(col), _p(values))
    return crow, col, values, row_pixels


def interpolation_matrix_backward(grad_values, vi, index_img, bary_img, row_pixels):
    index_img = index_img.contiguous()
    N, H, W = index_img.shape
    vi_c, vi_sN, F = _prep_vi(vi, N)
    g = grad_values.contiguous()
    rp = row_pixels.contiguous()
    bary_grad = th.zeros(N, 3, H, W, dtype=g.dtype)
    _call("interpolation_matrix_backward", _sfx(g), _p(g), _p(vi_c), _p(index_img), _p(rp), _i64(rp.numel()), _i64(N),
          _i64(F), _i64(vi_sN), _i64(H), _i64(W), _p(bary_grad))
    return bary_grad


def normal_matrix_structure(vi, num_vertices):
    """CSR pattern of A^T A and the per-face pair lookup (interpolate_module.cpp:167-241):
    returns (crow_indices int64 [V+1], col_indices int64 [nnz], pair_indices int32 [N,F,9])."""
    import numpy as np

    v = vi.cpu().numpy().astype(np.int64)
    assert v.ndim == 3
    assert (v >= 0).all() and (v < num_vertices).all(), "vi contains a vertex index outside [0, num_vertices)"
    keys = (v[:, :, :, None] * num_vertices + v[:, :, None, :]).reshape(-1)  # [N,F,3(i),3(j)] -> rows[i]*V + rows[j]
    uniq = np.unique(keys)
    rows = uniq // max(num_vertices, 1)
    cols = uniq - rows * num_vertices
    counts = np.bincount(rows, minlength=num_vertices)[:num_vertices]
    crow = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    pair = np.searchsorted(uniq, keys).astype(np.int32).reshape(v.shape[0], v.shape[1], 9)
    return th.from_numpy(crow), th.from_numpy(cols.astype(np.int64)), th.from_numpy(pair)


def normal_matrix_values(pair_indices, index_img, bary_img, nnz):
    index_img = index_img.contiguous()
    bary_img = bary_img.contiguous()
    pair = pair_indices.contiguous()
    N, H, W = index_img.shape
    F = pair.shape[1]
    values = th.zeros(nnz, dtype=bary_img.dtype)
    _call("normal_matrix_values", _sfx(bary_img), _p(pair), _p(index_img), _p(bary_img), _i64(N), _i64(F), _i64(F * 9),
          _i64(H), _i64(W), _p(values))
    return values


def normal_matrix_values_backward(grad_values, pair_indices, index_img, bary_img):
    index_img = index_img.contiguous()
    bary_img = bary_img.contiguous()
    pair = pair_indices.contiguous()
    g = grad_values.contiguous()
    N, H, W = index_img.shape
    F = pair.shape[1]
    bary_grad = th.zeros(N, 3, H, W, dtype=bary_img.dtype)
    _call("normal_matrix_values_backward", _sfx(bary_img), _p(g), _p(pair), _p(index_img), _p(bary_img), _i64(N),
          _i64(F), _i64(F * 9), _i64(H), _i64(W), _p(bary_grad))
    return bary_grad


# ---- anisotropic mipmap grid sampler (mipmap_grid_sampler_kernel.cu) --------------------------------


def _levels(levels):
    lv = [t.contiguous() for t in levels]
    n = len(lv)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in lv])
    lh = (ctypes.c_int64 * n)(*[t.shape[2] for t in lv])
    lw = (ctypes.c_int64 * n)(*[t.shape[3] for t in lv])
    return lv, ptrs, lh, lw


def mipmap_grid_sampler_2d(levels, grid, vt_dxdy_img, max_aniso, padding_mode=0, interpolation_mode=0,
                           align_corners=False, force_max_aniso=False, clip_grad=False):
    """Forward of mipmap_grid_sampler_ext::mipmap_grid_sampler_2d (enum arguments as the op takes them)."""
    lv, ptrs, lh, lw = _levels(levels)
    grid = grid.contiguous()
    vt = vt_dxdy_img.contiguous()
    N, C = lv[0].shape[:2]
    H, W = grid.shape[1:3]
    out = th.empty(N, C, H, W, dtype=lv[0].dtype)
    _call("mipmap_grid_sampler_2d", _sfx(lv[0]), ptrs, lh, lw, ctypes.c_int(len(lv)), _p(grid), _p(vt), _i64(N), _i64(C),
          _i64(H), _i64(W), ctypes.c_int(max_aniso), ctypes.c_int(padding_mode), ctypes.c_int(interpolation_mode),
          ctypes.c_int(bool(align_corners)), ctypes.c_int(bool(force_max_aniso)), ctypes.c_int(bool(clip_grad)), _p(out))
    return out


def mipmap_grid_sampler_2d_backward(grad_out, levels, grid, vt_dxdy_img, max_aniso, padding_mode=0,
                                    interpolation_mode=0, align_corners=False, force_max_aniso=False, clip_grad=False):
    """-> ([grad per level], grad_grid)."""
    lv, ptrs, lh, lw = _levels(levels)
    grid = grid.contiguous()
    vt = vt_dxdy_img.contiguous()
    go = grad_out.contiguous()
    N, C = lv[0].shape[:2]
    H, W = grid.shape[1:3]
    glv = [th.zeros_like(t) for t in lv]
    gptrs = (ctypes.c_void_p * len(lv))(*[t.data_ptr() for t in glv])
    ggrid = th.empty_like(grid)
    _call("mipmap_grid_sampler_2d_backward", _sfx(lv[0]), _p(go), ptrs, lh, lw, ctypes.c_int(len(lv)), _p(grid), _p(vt),
          _i64(N), _i64(C), _i64(H), _i64(W), ctypes.c_int(max_aniso), ctypes.c_int(padding_mode),
          ctypes.c_int(interpolation_mode), ctypes.c_int(bool(align_corners)), ctypes.c_int(bool(force_max_aniso)),
          ctypes.c_int(bool(clip_grad)), gptrs, _p(ggrid))
    return glv, ggrid


# ---- screen_space_uv_derivative (drtk/screen_space_uv_derivative.py:15-80) ---------------------------
def screen_space_uv_derivative(v, vt, vi, vti, index_img, bary_img, mask, campos, camrot, focal):
    """CPU restatement of the reference's PyTorch composite, step by step, on top of the oracle's own
    `interpolate`: face_dpdt (drtk/utils/geometry.py:71-82) -> interpolate of the per-face Jacobian and of
    the face vertex positions with a de-indexed face list -> project_points_grad (pinhole,
    drtk/utils/projection.py:683-699) -> 2x2 inverse -> mask.  Pinned against a fixture produced by the
    reference's own function (oracle/gen_golden_mipmap.py)."""
    vi_l, vti_l = vi.long(), vti.long()
    v012 = v[:, vi_l]  # [N,F,3,3]
    vt012 = vt[:, vti_l]  # [N,F,3,2]
    dpdb_t = v012[:, :, 1:3] - v012[:, :, 0:1]
    dtdb_t = vt012[:, :, 1:3] - vt012[:, :, 0:1]
    dpdt_t = th.inverse(dtdb_t) @ dpdb_t  # [N,F,2,3]
    N, F = dpdt_t.shape[:2]
    dpdt3 = dpdt_t[:, :, None].expand(-1, -1, 3, -1, -1)
    vi_dis = th.arange(0, 3 * F, dtype=th.int32).view(-1, 3)
    dpdt_img = interpolate(dpdt3.reshape(N, F * 3, 6).contiguous(), vi_dis, index_img, bary_img).permute(0, 2, 3, 1)
    dpdt_img = dpdt_img.reshape(*dpdt_img.shape[:3], 2, 3)
    vf_img = interpolate(v012.reshape(N, F * 3, 3).contiguous(), vi_dis, index_img, bary_img).permute(0, 2, 3, 1)
    vf_img = vf_img[:, :, :, None].expand(-1, -1, -1, 2, -1)
    v_grad = dpdt_img.reshape(N, -1, 3)
    vv = vf_img.reshape(N, -1, 3)
    v_cam_grad = (camrot[:, None] @ v_grad[..., None])[..., 0]
    v_cam = (camrot[:, None] @ (vv - campos[:, None])[..., None])[..., 0]
    z = v_cam[:, :, 2:3]
    z_grad = v_cam_grad[:, :, 2:3]
    z = th.where(z < 0, z.clamp(max=-1e-8), z.clamp(min=1e-8))
    v_proj_grad = (v_cam_grad[:, :, 0:2] * z - v_cam[:, :, 0:2] * z_grad) / z**2.0
    v_pix_grad = (focal[:, None] @ v_proj_grad[..., None])[..., 0]
    J = v_pix_grad.view(*dpdt_img.shape[:3], 2, 2)
    out, _ = th.linalg.inv_ex(J)
    out[~mask, :, :] = 0
    return out

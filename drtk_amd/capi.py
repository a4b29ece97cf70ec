"""ctypes binding of the C ABI (include/drtk_amd.h) on torch device tensors.

This is the same entry-point set the torch-op shim uses, callable without the dispatcher: parity
tests and the benchmark's per-kernel timing go through here.  Tensors must live on a HIP device;
every call enqueues on the given stream (default: torch's current stream) and returns new
output tensors.  There is no CPU path.
"""
import ctypes
import os
from typing import Optional, Tuple

import torch as th

from drtk_amd.utils import native_library_paths

_lib = None

DRTK_F32, DRTK_F64 = 0, 1


_lib_path = None

_POISON = os.environ.get("DRTK_CAPI_POISON", "") not in ("", "0")


# DRTK_CAPI_GUARD=g (tests): every output / workspace of this binding is carved out of a larger flat allocation with g
# sentinel elements on either side (16 g bytes for byte workspaces, which must stay 16-byte aligned) -- so outputs are
# only element-aligned (g = 1, 2, 3), and `check_guards()` tells whether a kernel wrote outside what it was given.
_GUARD = int(os.environ.get("DRTK_CAPI_GUARD", "0") or 0)
_guards = []


def _sentinel(dtype):
    return 0x5A if dtype == th.uint8 else (-7.25 if dtype.is_floating_point else -(2 ** 29) - 3)


def _out(*shape, dtype, device):
    """Output / workspace allocation of this binding: uninitialised memory -- or, with DRTK_CAPI_POISON=1 in the
    environment (the fuzzers and the GPU suite set it), memory pre-filled with NaN / a large negative integer / 0xA5
    bytes, so that an element a kernel forgot to write cannot pass for a value (freshly allocated device memory reads as
    zeros, which is a plausible image; see profiles/NOTES.md 3.1, round 3)."""
    if _GUARD:
        n = 1
        for d in shape:
            n *= int(d)
        g = _GUARD * (16 if dtype == th.uint8 else 1)
        flat = th.full((n + 2 * g,), _sentinel(dtype), dtype=dtype, device=device)
        _guards.append((flat, g, n))
        t = flat[g:g + n].view(*shape)
    else:
        t = th.empty(*shape, dtype=dtype, device=device)
    if _POISON and t.numel():
        if t.dtype.is_floating_point:
            t.fill_(float("nan"))
        elif t.dtype == th.uint8:
            t.fill_(0xA5)
        else:
            t.fill_(-(2 ** 30) - 7)
    return t


def check_guards() -> int:
    """DRTK_CAPI_GUARD: assert that the sentinel elements around every output and workspace handed out since the last
    call are intact (synchronises); returns how many allocations were checked.  Without the variable: 0."""
    th.cuda.synchronize()
    k = len(_guards)
    for flat, g, n in _guards:
        s = _sentinel(flat.dtype)
        ok = bool((flat[:g] == s).all()) and bool((flat[g + n:] == s).all())
        if not ok:
            _guards.clear()
            raise AssertionError(f"a kernel wrote outside a {flat.dtype} output / workspace of {n} elements")
    _guards.clear()
    return k



def use_profiling_library(path: str) -> None:
    """profiles/kernel_bench.py --flags only: bind this module to the ablation build of the same sources
    (profiles/libdrtk_amd_ablate.so, `python drtk_amd/build.py --ablation`) instead of the product library.
    Must be called before the first C-ABI call of the process."""
    global _lib_path
    assert _lib is None, "the C-ABI library is already loaded"
    _lib_path = path


DEPTH_ORDERS = {"strict": 0, "fastmath": 1}  # drtk_depth_order_t


def use_depth_order(order: str) -> None:
    """The rasterizer's depth-order setting (include/drtk_amd.h: drtk_amd_set_depth_order): `"fastmath"` evaluates the
    depth sum in the order of the reference AS BUILT by its setup.py (`-O3 --fast-math`), `"strict"` (the default) in the
    order its source spells.  One setting per process and library -- the torch operators call the same library and
    follow it; DRTK_AMD_DEPTH_ORDER=fastmath in the environment sets the initial value.  Affects later launches."""
    assert order in DEPTH_ORDERS, order
    _check(lib().drtk_amd_set_depth_order(DEPTH_ORDERS[order]), "set_depth_order")


def depth_order() -> str:
    o = lib().drtk_amd_get_depth_order()
    return next(k for k, v in DEPTH_ORDERS.items() if v == o)


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        path = _lib_path or native_library_paths()[0]
        if not os.path.isfile(path):
            build_py = os.path.join(os.path.dirname(os.path.abspath(__file__)), "build.py")
            raise ImportError(f"{path} is missing: run `python {build_py}`")
        L = ctypes.CDLL(path)
        L.drtk_amd_status_string.restype = ctypes.c_char_p
        L.drtk_amd_status_string.argtypes = [ctypes.c_int]
        L.drtk_amd_version.restype = ctypes.c_char_p
        for name in EXPORTS:
            if name not in ("drtk_amd_status_string", "drtk_amd_version"):
                getattr(L, name).restype = ctypes.c_int
        _lib = L
    return _lib


EXPORTS = [
    "drtk_amd_status_string",
    "drtk_amd_version",
    "drtk_amd_set_depth_order",
    "drtk_amd_get_depth_order",
    "drtk_amd_rasterize_workspace_bytes",
    "drtk_amd_rasterize_lines_workspace_bytes",
    "drtk_amd_rasterize",
    "drtk_amd_render",
    "drtk_amd_render_backward",
    "drtk_amd_interpolate",
    "drtk_amd_interpolate_masked",
    "drtk_amd_interpolate_backward",
    "drtk_amd_interpolate_backward_workspace_bytes",
    "drtk_amd_interpolate_backward_ws",
    "drtk_amd_edge_grad_backward_workspace_bytes",
    "drtk_amd_edge_grad_backward",
    "drtk_amd_edge_grad_backward_fused_workspace_bytes",
    "drtk_amd_edge_grad_backward_fused",
    "drtk_amd_interpolation_matrix",
    "drtk_amd_interpolation_matrix_backward",
    "drtk_amd_interpolation_normal_matrix_values",
    "drtk_amd_interpolation_normal_matrix_values_backward",
    "drtk_amd_mipmap_grid_sampler_2d",
    "drtk_amd_mipmap_grid_sampler_2d_backward",
    "drtk_amd_screen_space_uv_derivative",
    "drtk_amd_transform_pinhole",
    "drtk_amd_transform_pinhole_backward",
    "drtk_amd_selftest_exact_div",
    "drtk_amd_kernel_timing_begin",
    "drtk_amd_kernel_timing_report",
]


class DrtkAmdError(RuntimeError):
    pass


def _check(status: int, what: str):
    if status != 0:
        raise DrtkAmdError(f"{what}: {lib().drtk_amd_status_string(status).decode()} (status {status})")


def _dt(t: th.Tensor) -> int:
    if t.dtype == th.float32:
        return DRTK_F32
    if t.dtype == th.float64:
        return DRTK_F64
    raise TypeError(f"drtk_amd: unsupported dtype {t.dtype}")


def _p(t: Optional[th.Tensor]):
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


def _i(x) -> ctypes.c_int64:
    return ctypes.c_int64(int(x))


def _on_tensor_device(fn):
    """Runs `fn` with the device of its first tensor argument made current.  The library sizes its grids from, and
    launches on, hipGetDevice() -- the torch-op shim installs a device guard for that, this is the same guard for
    ctypes callers: tensors on cuda:1 while cuda:0 is current would otherwise be launched against the wrong device.
    Tensors of one call that live on different devices are an error."""
    import functools

    def tensors(args):
        for a in args:
            if isinstance(a, th.Tensor):
                yield a
            elif isinstance(a, (list, tuple)):
                yield from (x for x in a if isinstance(x, th.Tensor))

    @functools.wraps(fn)
    def guarded(*args, **kwargs):
        ts = list(tensors(list(args) + list(kwargs.values())))
        if not ts:
            return fn(*args, **kwargs)
        if not ts[0].is_cuda:
            raise DrtkAmdError("drtk_amd implements the MI355X (HIP) path only; got a CPU tensor")
        devs = {t.device for t in ts}
        if len(devs) != 1:
            raise DrtkAmdError(f"drtk_amd: tensors of one call on different devices: {sorted(map(str, devs))}")
        with th.cuda.device(ts[0].device):
            return fn(*args, **kwargs)

    return guarded


def _stream(t: th.Tensor, stream) -> ctypes.c_void_p:
    if not t.is_cuda:
        raise DrtkAmdError("drtk_amd implements the MI355X (HIP) path only; got a CPU tensor")
    s = th.cuda.current_stream(t.device) if stream is None else stream
    return ctypes.c_void_p(s.cuda_stream)


def _vi(vi: th.Tensor, n: int):
    assert vi.dtype == th.int32
    if vi.ndim == 2:
        c = vi.contiguous()
        return c, 0, c.shape[0]
    if vi.shape[0] > 1 and vi.stride(0) == 0:
        c = vi[0].contiguous()
        return c, 0, c.shape[0]
    c = vi.contiguous()
    return c, c.shape[1] * 3, c.shape[1]


def rasterize_workspace_bytes(N, F, H, W) -> int:
    out = ctypes.c_size_t(0)
    _check(lib().drtk_amd_rasterize_workspace_bytes(_i(N), _i(F), _i(H), _i(W), ctypes.byref(out)), "rasterize")
    return out.value


def rasterize_lines_workspace_bytes(N, H, W) -> int:
    out = ctypes.c_size_t(0)
    _check(lib().drtk_amd_rasterize_lines_workspace_bytes(_i(N), _i(H), _i(W), ctypes.byref(out)), "rasterize")
    return out.value


@_on_tensor_device
def rasterize(v, vi, height, width, stream=None, workspace=None, wireframe=False, out=None) -> Tuple[th.Tensor, th.Tensor]:
    """`out` = (depth_img, index_img) to write into (contiguous [N,H,W] float32 / int32): tests pre-fill them with a
    sentinel to see that every pixel is written."""
    v = v.contiguous()
    N, V, _ = v.shape
    vi_c, vi_sN, F = _vi(vi, N)
    if out is not None:
        depth, index = out
        assert depth.shape == (N, height, width) and index.shape == (N, height, width) and depth.is_contiguous() and index.is_contiguous()
        assert depth.dtype == th.float32 and index.dtype == th.int32
    else:
        depth = _out(N, height, width, dtype=th.float32, device=v.device)
        index = _out(N, height, width, dtype=th.int32, device=v.device)
    nbytes = rasterize_lines_workspace_bytes(N, height, width) if wireframe else rasterize_workspace_bytes(N, F, height, width)
    ws = workspace if workspace is not None else _out(nbytes, dtype=th.uint8, device=v.device)
    _check(
        lib().drtk_amd_rasterize(
            ctypes.c_int(_dt(v)), _p(v), _p(vi_c), _i(N), _i(V), _i(F), _i(vi_sN), _i(height), _i(width),
            ctypes.c_int(1 if wireframe else 0), _p(depth), _p(index), _p(ws), ctypes.c_size_t(ws.numel()),
            _stream(v, stream)),
        "rasterize")
    return depth, index


@_on_tensor_device
def render(v, vi, index_img, stream=None):
    v = v.contiguous()
    index_img = index_img.contiguous()
    N, V, _ = v.shape
    H, W = index_img.shape[1:]
    vi_c, vi_sN, F = _vi(vi, N)
    depth = _out(N, H, W, dtype=v.dtype, device=v.device)
    bary = _out(N, 3, H, W, dtype=v.dtype, device=v.device)
    _check(
        lib().drtk_amd_render(
            ctypes.c_int(_dt(v)), _p(v), _p(vi_c), _p(index_img), _i(N), _i(V), _i(F), _i(vi_sN), _i(H), _i(W),
            _p(depth), _p(bary), _stream(v, stream)),
        "render")
    return depth, bary


@_on_tensor_device
def render_backward(v, vi, index_img, grad_depth_img, grad_bary_img, stream=None):
    v = v.contiguous()
    index_img = index_img.contiguous()
    gd, gb = grad_depth_img.contiguous(), grad_bary_img.contiguous()
    N, V, _ = v.shape
    H, W = index_img.shape[1:]
    vi_c, vi_sN, F = _vi(vi, N)
    grad_v = _out(N, V, 3, dtype=v.dtype, device=v.device)
    _check(
        lib().drtk_amd_render_backward(
            ctypes.c_int(_dt(v)), _p(v), _p(vi_c), _p(index_img), _p(gd), _p(gb), _i(N), _i(V), _i(F), _i(vi_sN),
            _i(H), _i(W), _p(grad_v), _stream(v, stream)),
        "render_backward")
    return grad_v


@_on_tensor_device
def interpolate(attrs, vi, index_img, bary_img, stream=None, masked=False):
    attrs = attrs.contiguous()
    index_img = index_img.contiguous()
    bary_img = bary_img.contiguous()
    N, V, C = attrs.shape
    H, W = index_img.shape[1:]
    vi_c, vi_sN, F = _vi(vi, N)
    out = _out(N, C, H, W, dtype=attrs.dtype, device=attrs.device)
    fn = lib().drtk_amd_interpolate_masked if masked else lib().drtk_amd_interpolate
    _check(
        fn(ctypes.c_int(_dt(attrs)), _p(attrs), _p(vi_c), _p(index_img), _p(bary_img), _i(N), _i(V), _i(C), _i(F),
           _i(vi_sN), _i(H), _i(W), _p(out), _stream(attrs, stream)),
        "interpolate_masked" if masked else "interpolate")
    return out


def interpolate_masked(attrs, vi, index_img, bary_img, stream=None):
    """drtk_amd extension: `interpolate` with the background pixels written as 0 (include/drtk_amd.h)."""
    return interpolate(attrs, vi, index_img, bary_img, stream, masked=True)


@_on_tensor_device
def interpolate_backward(grad_out, attrs, vi, index_img, bary_img, vert_requires_grad=True,
                         bary_requires_grad=True, stream=None, workspace=None):
    """workspace=False: the entry point's unpadded route (no scratch buffer), for A/B and for the parity tests of both routes."""
    grad_out = grad_out.contiguous()
    attrs = attrs.contiguous()
    index_img = index_img.contiguous()
    bary_img = bary_img.contiguous()
    N, V, C = attrs.shape
    H, W = index_img.shape[1:]
    vi_c, vi_sN, F = _vi(vi, N)
    ag = _out(N, V, C, dtype=attrs.dtype, device=attrs.device) if vert_requires_grad else None
    bg = _out(N, 3, H, W, dtype=attrs.dtype, device=attrs.device) if bary_requires_grad else None
    ws, nbytes = None, 0
    if os.environ.get("DRTK_CAPI_NO_INTERP_WS"):  # A/B of the two routes from the command line (profiles/shape_bench.py)
        workspace = False
    if vert_requires_grad and workspace is not False:  # the optional scratch of the padded-row route (include/drtk_amd.h)
        out = ctypes.c_size_t(0)
        _check(lib().drtk_amd_interpolate_backward_workspace_bytes(ctypes.c_int(_dt(attrs)), _i(N), _i(V), _i(C), ctypes.byref(out)), "interpolate_backward")
        nbytes = int(out.value)
        if nbytes:
            ws = _out((nbytes + 3) // 4, dtype=th.float32, device=attrs.device)
            assert ws.data_ptr() % 64 == 0 or os.environ.get("DRTK_CAPI_GUARD"), "torch allocations are 256-byte aligned"
            if ws.data_ptr() % 64 != 0:  # (guarded allocations are only element-aligned: no workspace then)
                ws, nbytes = None, 0
    _check(
        lib().drtk_amd_interpolate_backward_ws(
            ctypes.c_int(_dt(attrs)), _p(grad_out), _p(attrs), _p(vi_c), _p(index_img), _p(bary_img), _i(N), _i(V),
            _i(C), _i(F), _i(vi_sN), _i(H), _i(W), _p(ag), _p(bg), _p(ws), ctypes.c_size_t(nbytes), _stream(attrs, stream)),
        "interpolate_backward")
    return ag, bg


@_on_tensor_device
def interpolation_matrix(vi, index_img, bary_img, stream=None):
    """-> (crow_indices, col_indices, values, row_pixels); row_pixels/crow are built with torch ops
    (the caller's side of the C ABI), columns and values by drtk_amd_interpolation_matrix."""
    index_img = index_img.contiguous()
    bary_img = bary_img.contiguous()
    N, H, W = index_img.shape
    vi_c, vi_sN, F = _vi(vi, N)
    row_pixels = th.nonzero(index_img.reshape(-1).ne(-1)).reshape(-1)
    R = row_pixels.numel()
    crow = th.arange(0, 3 * R + 1, 3, dtype=th.int64, device=index_img.device)
    col = _out(3 * R, dtype=th.int64, device=index_img.device)
    values = _out(3 * R, dtype=bary_img.dtype, device=bary_img.device)
    _check(
        lib().drtk_amd_interpolation_matrix(
            ctypes.c_int(_dt(bary_img)), _p(vi_c), _p(index_img), _p(bary_img), _p(row_pixels), _i(R), _i(N), _i(F),
            _i(vi_sN), _i(H), _i(W), _p(col), _p(values), _stream(bary_img, stream)),
        "interpolation_matrix")
    return crow, col, values, row_pixels


@_on_tensor_device
def interpolation_matrix_backward(grad_values, vi, index_img, row_pixels, stream=None):
    grad_values = grad_values.contiguous()
    index_img = index_img.contiguous()
    row_pixels = row_pixels.contiguous()
    N, H, W = index_img.shape
    vi_c, vi_sN, F = _vi(vi, N)
    bg = _out(N, 3, H, W, dtype=grad_values.dtype, device=grad_values.device)
    _check(
        lib().drtk_amd_interpolation_matrix_backward(
            ctypes.c_int(_dt(grad_values)), _p(grad_values), _p(vi_c), _p(index_img), _p(row_pixels),
            _i(row_pixels.numel()), _i(N), _i(F), _i(vi_sN), _i(H), _i(W), _p(bg), _stream(grad_values, stream)),
        "interpolation_matrix_backward")
    return bg


def _pairs(pair_indices, N):
    assert pair_indices.dtype == th.int32 and pair_indices.shape[-1] == 9
    if pair_indices.ndim == 2:
        return pair_indices.contiguous(), 0, pair_indices.shape[0]
    if pair_indices.shape[0] == N and N > 1 and pair_indices.stride(0) == 0:
        return pair_indices[0].contiguous(), 0, pair_indices.shape[1]
    p = pair_indices.contiguous()
    return p, p.shape[1] * 9, p.shape[1]


@_on_tensor_device
def interpolation_normal_matrix_values(pair_indices, index_img, bary_img, nnz, stream=None):
    index_img = index_img.contiguous()
    bary_img = bary_img.contiguous()
    N, H, W = index_img.shape
    pr, pair_sN, F = _pairs(pair_indices, N)
    values = _out(nnz, dtype=bary_img.dtype, device=bary_img.device)
    _check(
        lib().drtk_amd_interpolation_normal_matrix_values(
            ctypes.c_int(_dt(bary_img)), _p(pr), _p(index_img), _p(bary_img), _i(N), _i(F), _i(pair_sN), _i(H), _i(W),
            _i(nnz), _p(values), _stream(bary_img, stream)),
        "interpolation_normal_matrix_values")
    return values


@_on_tensor_device
def interpolation_normal_matrix_values_backward(grad_values, pair_indices, index_img, bary_img, stream=None):
    grad_values = grad_values.contiguous()
    index_img = index_img.contiguous()
    bary_img = bary_img.contiguous()
    N, H, W = index_img.shape
    pr, pair_sN, F = _pairs(pair_indices, N)
    bg = _out(N, 3, H, W, dtype=bary_img.dtype, device=bary_img.device)
    _check(
        lib().drtk_amd_interpolation_normal_matrix_values_backward(
            ctypes.c_int(_dt(bary_img)), _p(grad_values), _p(pr), _p(index_img), _p(bary_img), _i(N), _i(F),
            _i(pair_sN), _i(H), _i(W), _p(bg), _stream(bary_img, stream)),
        "interpolation_normal_matrix_values_backward")
    return bg


def _level_table(levels):
    """Device pointers, sizes and view strides of a mip pyramid.  A level whose views are contiguous [C,h,w] blocks is
    passed as it is, whatever its batch stride -- in particular a [1,C,h,w] texture expanded to N views (stride 0) is
    not materialised N times; anything else is made contiguous first."""
    lv = []
    for t in levels:
        ok = t.dim() == 4 and (t.shape[0] == 0 or t[0].is_contiguous()) and (t.stride(0) == 0 or t.stride(0) >= t[0].numel() or t.shape[0] <= 1)
        lv.append(t if ok else t.contiguous())
    n = len(lv)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in lv])
    lh = (ctypes.c_int64 * n)(*[t.shape[2] for t in lv])
    lw = (ctypes.c_int64 * n)(*[t.shape[3] for t in lv])
    lsn = (ctypes.c_int64 * n)(*[(t.stride(0) if t.shape[0] > 1 else t.shape[1] * t.shape[2] * t.shape[3]) for t in lv])
    return lv, ptrs, lh, lw, lsn


def _grid_layout(grid):
    """(tensor to pass, {sN, sP, sC} in elements): a uv field [N,H,W,2] whose pixels are evenly spaced in memory is read
    in place -- contiguous, or the channel-first image of `interpolate` seen through permute(0, 2, 3, 1); anything else
    (overlapping views, rows with padding, other permutations) is made contiguous first."""
    N, H, W, two = grid.shape
    sN, sH, sW, sC = grid.stride()
    P = H * W
    rows_ok = H <= 1 or sH == W * sW
    pixel_major = sC == 1 and sW == 2 and rows_ok and (N <= 1 or sN >= 2 * P)             # [N,H,W,2]
    channel_major = sW == 1 and sC >= P and rows_ok and (N <= 1 or sN >= sC + P)          # [N,2,H,W] seen as [N,H,W,2]
    if not (two == 2 and P > 0 and (pixel_major or channel_major)):
        grid = grid.contiguous()
        sN, sW, sC = 2 * P, 2, 1
    return grid, (ctypes.c_int64 * 3)(sN if N > 1 else 2 * P, sW, sC)


@_on_tensor_device
def mipmap_grid_sampler_2d(levels, grid, vt_dxdy_img, max_aniso, padding_mode=0, interpolation_mode=0,
                           align_corners=False, force_max_aniso=False, clip_grad=False, stream=None):
    lv, ptrs, lh, lw, lsn = _level_table(levels)
    grid, glayout = _grid_layout(grid)
    vt = vt_dxdy_img.contiguous()
    N, C = lv[0].shape[:2]
    H, W = grid.shape[1:3]
    out = _out(N, C, H, W, dtype=lv[0].dtype, device=lv[0].device)
    _check(
        lib().drtk_amd_mipmap_grid_sampler_2d(
            ctypes.c_int(_dt(lv[0])), ptrs, lh, lw, lsn, ctypes.c_int(len(lv)), _p(grid), glayout, _p(vt), _i(N), _i(C), _i(H), _i(W),
            ctypes.c_int(max_aniso), ctypes.c_int(padding_mode), ctypes.c_int(interpolation_mode),
            ctypes.c_int(bool(align_corners)), ctypes.c_int(bool(force_max_aniso)), ctypes.c_int(bool(clip_grad)),
            _p(out), _stream(lv[0], stream)),
        "mipmap_grid_sampler_2d")
    return out


@_on_tensor_device
def mipmap_grid_sampler_2d_backward(grad_out, levels, grid, vt_dxdy_img, max_aniso, padding_mode=0,
                                    interpolation_mode=0, align_corners=False, force_max_aniso=False,
                                    clip_grad=False, stream=None):
    lv, ptrs, lh, lw, lsn = _level_table(levels)
    grid, glayout = _grid_layout(grid)
    vt = vt_dxdy_img.contiguous()
    grad_out = grad_out.contiguous()
    N, C = lv[0].shape[:2]
    H, W = grid.shape[1:3]
    # contiguous even for an expanded pyramid, and back to back in one buffer: the call then zeroes them with one launch
    flat = _out(sum(t.numel() for t in lv), dtype=lv[0].dtype, device=lv[0].device)
    glv, off = [], 0
    for t in lv:
        glv.append(flat[off:off + t.numel()].view(t.shape))
        off += t.numel()
    gptrs = (ctypes.c_void_p * len(lv))(*[t.data_ptr() for t in glv])
    ggrid = th.empty_strided(grid.shape, grid.stride(), dtype=grid.dtype, device=grid.device)  # laid out like the grid it belongs to
    _check(
        lib().drtk_amd_mipmap_grid_sampler_2d_backward(
            ctypes.c_int(_dt(lv[0])), _p(grad_out), ptrs, lh, lw, lsn, ctypes.c_int(len(lv)), _p(grid), glayout, _p(vt), _i(N), _i(C),
            _i(H), _i(W), ctypes.c_int(max_aniso), ctypes.c_int(padding_mode), ctypes.c_int(interpolation_mode),
            ctypes.c_int(bool(align_corners)), ctypes.c_int(bool(force_max_aniso)), ctypes.c_int(bool(clip_grad)),
            gptrs, _p(ggrid), glayout, _stream(lv[0], stream)),
        "mipmap_grid_sampler_2d_backward")
    return glv, ggrid


@_on_tensor_device
def screen_space_uv_derivative(v, vt, vi, vti, index_img, bary_img, mask, campos, camrot, focal, stream=None):
    """v [N,V,3] or shared [V,3]; vt [N,T,2] or shared [T,2]; mask bool/uint8 [N,H,W] or None."""
    index_img = index_img.contiguous()
    bary_img = bary_img.contiguous()
    N, H, W = index_img.shape
    v_c, vt_c = v.contiguous(), vt.contiguous()
    v_sN = 0 if v_c.ndim == 2 else v_c.shape[1] * 3
    vt_sN = 0 if vt_c.ndim == 2 else vt_c.shape[1] * 2
    V, T = v_c.shape[-2], vt_c.shape[-2]
    vi_c, vti_c = vi.contiguous(), vti.contiguous()
    m = None if mask is None else mask.to(th.uint8).contiguous()
    out = _out(N, H, W, 2, 2, dtype=bary_img.dtype, device=bary_img.device)
    _check(
        lib().drtk_amd_screen_space_uv_derivative(
            ctypes.c_int(_dt(bary_img)), _p(v_c), _i(v_sN), _p(vt_c), _i(vt_sN), _p(vi_c), _p(vti_c), _p(index_img),
            _p(bary_img), _p(m), _p(campos.contiguous()), _p(camrot.contiguous()), _p(focal.contiguous()), _i(N), _i(V),
            _i(T), _i(vi_c.shape[0]), _i(H), _i(W), _p(out), _stream(bary_img, stream)),
        "screen_space_uv_derivative")
    return out


def edge_grad_backward_workspace_bytes(dtype, N, H, W) -> int:
    out = ctypes.c_size_t(0)
    code = DRTK_F32 if dtype == th.float32 else DRTK_F64
    _check(lib().drtk_amd_edge_grad_backward_workspace_bytes(ctypes.c_int(code), _i(N), _i(H), _i(W), ctypes.byref(out)),
           "edge_grad_backward")
    return out.value


@_on_tensor_device
def edge_grad_backward(v_pix, img, index_img, vi, grad_output, max_dp_dr=1e4, stream=None, workspace=None):
    v_pix = v_pix.contiguous()
    img = img.contiguous()
    index_img = index_img.contiguous()
    grad_output = grad_output.contiguous()
    N, V, _ = v_pix.shape
    C, H, W = img.shape[1:]
    vi_c, vi_sN, F = _vi(vi, N)
    out = _out(N, 3, H, W, dtype=v_pix.dtype, device=v_pix.device)
    nbytes = edge_grad_backward_workspace_bytes(v_pix.dtype, N, H, W)
    ws = workspace if workspace is not None else _out(nbytes, dtype=th.uint8, device=v_pix.device)
    _check(
        lib().drtk_amd_edge_grad_backward(
            ctypes.c_int(_dt(v_pix)), _p(v_pix), _p(img), _p(index_img), _p(vi_c), _p(grad_output), _i(N), _i(V),
            _i(C), _i(F), _i(vi_sN), _i(H), _i(W), ctypes.c_double(max_dp_dr), _p(out), _p(ws),
            ctypes.c_size_t(ws.numel()), _stream(v_pix, stream)),
        "edge_grad_backward")
    return out


@_on_tensor_device
def edge_grad_backward_fused(v_pix, img, index_img, vi, bary_img, grad_output, max_dp_dr=1e4, stream=None):
    """grad_v_pix [N,V,3] = interpolate_backward(edge_grad_backward(...), v_pix, ...) in one call."""
    v_pix = v_pix.contiguous()
    img = img.contiguous()
    index_img = index_img.contiguous()
    bary_img = bary_img.contiguous()
    grad_output = grad_output.contiguous()
    N, V, _ = v_pix.shape
    C, H, W = img.shape[1:]
    vi_c, vi_sN, F = _vi(vi, N)
    out = _out(N, V, 3, dtype=v_pix.dtype, device=v_pix.device)
    nb = ctypes.c_size_t(0)
    code = DRTK_F32 if v_pix.dtype == th.float32 else DRTK_F64
    _check(lib().drtk_amd_edge_grad_backward_fused_workspace_bytes(ctypes.c_int(code), _i(N), _i(H), _i(W), ctypes.byref(nb)),
           "edge_grad_backward_fused")
    ws = _out(nb.value, dtype=th.uint8, device=v_pix.device)
    _check(
        lib().drtk_amd_edge_grad_backward_fused(
            ctypes.c_int(_dt(v_pix)), _p(v_pix), _p(img), _p(index_img), _p(vi_c), _p(bary_img), _p(grad_output),
            _i(N), _i(V), _i(C), _i(F), _i(vi_sN), _i(H), _i(W), ctypes.c_double(max_dp_dr), _p(out), _p(ws),
            ctypes.c_size_t(ws.numel()), _stream(v_pix, stream)),
        "edge_grad_backward_fused")
    return out


def selftest_exact_div(dtype=th.float32, seed=1, count=1 << 28, device="cuda:0") -> int:
    """Number of (n, d) pairs for which the rasterizer's exact division differs from IEEE `/` (must be 0)."""
    out = th.zeros(1, dtype=th.int64, device=device)
    code = DRTK_F32 if dtype == th.float32 else DRTK_F64
    _check(lib().drtk_amd_selftest_exact_div(ctypes.c_int(code), ctypes.c_uint64(seed), _i(count), _p(out),
                                             _stream(out, None)), "selftest_exact_div")
    return int(out.item())


def kernel_timing_begin() -> None:
    """Open a per-kernel timing collection: until `kernel_timing_report()` every kernel the library launches (through
    this module or through the torch operators -- same library) is bracketed by HIP events on its launch stream."""
    _check(lib().drtk_amd_kernel_timing_begin(), "kernel_timing_begin")


def kernel_timing_report() -> dict:
    """Close the collection and return {launch-site kernel name: (launches, total_ms)} in order of first launch."""
    need = ctypes.c_size_t(0)
    buf = ctypes.create_string_buffer(1 << 16)
    _check(lib().drtk_amd_kernel_timing_report(buf, ctypes.c_size_t(len(buf)), ctypes.byref(need)), "kernel_timing_report")
    out = {}
    for line in buf.value.decode().splitlines():
        name, count, ms = line.rsplit("\t", 2)
        out[name.strip("()")] = (int(count), float(ms))
    return out

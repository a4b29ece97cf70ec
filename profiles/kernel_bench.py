#!/usr/bin/env python3
"""Runs each hot-path kernel a few times through the C ABI on the bench workload (no autograd, no
PyTorch glue) -- the command to put behind `rocprofv3 --kernel-trace --stats` or `--pmc ...` when a
single kernel is under study.   python3 profiles/kernel_bench.py [--only rasterize] [--reps 3]"""
import argparse
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drtk_amd import capi  # noqa: E402
from drtk_amd import synthetic as S  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--only", default="")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--views", type=int, default=8)
ap.add_argument("--mesh", default="100k")
ap.add_argument("--res", type=int, default=2048)
ap.add_argument("--channels", type=int, default=16)
ap.add_argument("--tex", type=int, default=4096)
ap.add_argument("--uvscale", type=float, default=1.0)
ap.add_argument("--flags", default="0", help="comma list of ablation masks to time every kernel under; anything but 0 "
                "needs the ablation build (python drtk_amd/build.py --ablation -> profiles/libdrtk_amd_ablate.so): the "
                "product library has no such switches")
ap.add_argument("--check", action="store_true", help="also compare interpolate_backward of the bound library with a "
                "torch formulation of the same sums on the GPU (for --lib variants, which no test suite binds to)")
ap.add_argument("--dump", default="", help="save the outputs of the edge routes, render backward and rasterize (view 0) of the bound "
                "library: two libraries are compared with `mipmap_bench.py --compare A B`")
ap.add_argument("--lib", default="", help="A/B a kernel variant: path of another build of the library (e.g. one compiled with a -D switch)")
a = ap.parse_args()
if a.lib:
    capi.use_profiling_library(os.path.abspath(a.lib))
ABLATE = a.flags != "0" or bool(os.environ.get("DRTK_ABLATE"))
if ABLATE and not a.lib:
    _path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdrtk_amd_ablate.so")
    assert os.path.isfile(_path), f"{_path} missing: run `python drtk_amd/build.py --ablation`"
    capi.use_profiling_library(_path)


def set_flags(flags):
    if ABLATE:
        capi.lib().drtk_amd_debug_set_flags(flags)


dev = "cuda:0"
nl, no = S.MESH_SIZES[a.mesh]
v, vi = S.sphere_views(a.views, nl, no, a.res, a.res, lobes=0.05, device=dev)
attr = S.random_attributes(a.views, v.shape[1], a.channels, shared=False, device=dev)
H = W = a.res
depth0, index = capi.rasterize(v, vi, H, W)
depth, bary = capi.render(v, vi, index)
img = capi.interpolate(attr, vi, index, bary) * (index != -1)[:, None]
g = th.Generator(device=dev).manual_seed(0)
go = th.rand(img.shape, device=dev, generator=g) * 2 - 1
gd = th.rand(depth.shape, device=dev, generator=g)
gb = th.rand(bary.shape, device=dev, generator=g)
eg = capi.edge_grad_backward(v, img, index, vi, go)
kernels = {
    "rasterize": lambda: capi.rasterize(v, vi, H, W),
    "render": lambda: capi.render(v, vi, index),
    "interpolate": lambda: capi.interpolate(attr, vi, index, bary),
    "edge_grad_backward": lambda: capi.edge_grad_backward(v, img, index, vi, go),
    "edge_grad_backward_fused": lambda: capi.edge_grad_backward_fused(v, img, index, vi, bary, go),
    "interpolate_backward_vpix": lambda: capi.interpolate_backward(eg, v, vi, index, bary, True, False),
    "interpolate_backward": lambda: capi.interpolate_backward(go, attr, vi, index, bary, True, True),
    "render_backward": lambda: capi.render_backward(v, vi, index, gd, gb),
}
# algorithmic bytes per launch for the kernels that have a figure (SURVEY 8d per-pixel tensors; the sparse operators: DESIGN 3)
NPX = a.views * H * W
ALG_BYTES = {
    "rasterize": 8 * NPX, "render": 20 * NPX, "interpolate": (16 + 4 * a.channels) * NPX,
    "interpolate_backward": (4 * a.channels + 28) * NPX, "render_backward": 20 * NPX,
}
if any(k in a.only for k in ("interpolation_matrix", "normal_matrix", "wireframe", "sparse")):
    # SURVEY 8f rows 1 and 4 on the same views: the sparse interpolation operators and wireframe rasterization
    import drtk_amd  # noqa: F401  (registers the operators: the A^T A pattern is built by the shim's device-side builder)

    row_pixels = th.nonzero(index.reshape(-1).ne(-1)).reshape(-1)
    R = row_pixels.numel()
    crow, col, vals, _ = capi.interpolation_matrix(vi, index, bary)
    gvals = th.rand(vals.shape, device=dev, generator=g)
    V = v.shape[1]
    ncrow, ncol, pair = th.ops.drtk_amd_ext.normal_matrix_structure(vi[None].expand(a.views, -1, -1) if vi.dim() == 2 else vi, V)
    nnz = ncol.numel()
    gnv = th.rand(nnz, device=dev, generator=g)
    col_o = th.empty(3 * R, dtype=th.int64, device=dev)
    val_o = th.empty(3 * R, dtype=bary.dtype, device=dev)
    vi_c, vi_sN, F_ = capi._vi(vi, a.views)

    def matrix_kernel_only():  # the C-ABI entry alone (rows given): what the kernel costs without at::nonzero and the allocations
        capi._check(capi.lib().drtk_amd_interpolation_matrix(
            capi.ctypes.c_int(0), capi._p(vi_c), capi._p(index), capi._p(bary), capi._p(row_pixels), capi._i(R), capi._i(a.views), capi._i(F_),
            capi._i(vi_sN), capi._i(H), capi._i(W), capi._p(col_o), capi._p(val_o), capi._stream(bary, None)), "interpolation_matrix")

    kernels["interpolation_matrix_kernel_only"] = matrix_kernel_only
    kernels["interpolation_matrix"] = lambda: capi.interpolation_matrix(vi, index, bary)  # + nonzero (host sync), crow, allocations
    kernels["interpolation_matrix_backward"] = lambda: capi.interpolation_matrix_backward(gvals, vi, index, row_pixels)
    kernels["normal_matrix_values"] = lambda: capi.interpolation_normal_matrix_values(pair, index, bary, nnz)
    kernels["normal_matrix_values_backward"] = lambda: capi.interpolation_normal_matrix_values_backward(gnv, pair, index, bary)
    # wireframe: every edge enabled (bits 28..30 of vi[...,0]) -- the densest line image the mode can be asked for
    vi_w = vi.clone()
    vi_w[..., 0] |= 0x70000000
    kernels["rasterize_wireframe"] = lambda: capi.rasterize(v, vi_w, H, W, wireframe=True)
    # bytes: the row's pixel id (8) + index (4) + three corner ids (12) + three barycentrics (12) in, three columns (24) +
    # three values (12) out = 72 per row for the CSR build; backward reads 8 + 4 + 12 + 12 and writes 12 per row + the 12 B/px fill
    ALG_BYTES.update({
        "interpolation_matrix_kernel_only": 72 * R, "interpolation_matrix_backward": 48 * R + 12 * NPX,
        "normal_matrix_values": 16 * NPX + 36 * (R // 7), "normal_matrix_values_backward": 28 * NPX,
        "rasterize_wireframe": 8 * NPX,
    })
    print(f"sparse operators: R = {R} rows ({R / NPX:.2f} of the pixels), A^T A nnz = {nnz}, V = {V}")
if "mipmap" in a.only:
    # textured shading of the same views: the sphere's own lat/long atlas as uv
    # (smooth, anisotropic towards the limb and the poles), RGB texture with its full pyramid,
    # Jacobian from finite differences of the uv image.  --tex picks the texture size: 4096 gives
    # ~1-2 texels per pixel (levels 0-1), 16384-equivalent minification is emulated with --uvscale.
    S = a.tex
    gt = th.Generator(device=dev).manual_seed(1)
    tex = [th.rand(a.views, 3, S, S, device=dev, generator=gt)]
    while tex[-1].shape[-1] > 1 and len(tex) < 11:  # the op takes at most 11 levels
        tex.append(th.nn.functional.avg_pool2d(tex[-1], 2))
    vid = th.arange(v.shape[1], device=dev)
    uv_attr = th.stack([(vid % no).float() / no, (vid // no).float() / nl], -1)[None].expand(a.views, -1, -1).contiguous()
    uv = capi.interpolate(uv_attr, vi, index, bary).permute(0, 2, 3, 1).contiguous() * a.uvscale  # [N,H,W,2]
    uv = th.where((index != -1)[..., None], uv, th.zeros_like(uv))
    uvn = (uv % 1.0) * 2 - 1
    jac = th.zeros(a.views, H, W, 2, 2, device=dev)
    jac[:, :, :-1, 0, :] = uv[:, :, 1:] - uv[:, :, :-1]
    jac[:, :-1, :, 1, :] = uv[:, 1:] - uv[:, :-1]
    jac = th.where(jac.abs() > 0.25 * a.uvscale, th.zeros_like(jac), jac).contiguous()  # atlas seam / silhouette
    gmo = th.rand(a.views, 3, H, W, device=dev, generator=gt) * (index != -1)[:, None]  # masked like a real loss
    kernels["mipmap_fwd"] = lambda: capi.mipmap_grid_sampler_2d(tex, uvn, jac, 8, 1, 0)
    kernels["mipmap_bwd"] = lambda: capi.mipmap_grid_sampler_2d_backward(gmo, tex, uvn, jac, 8, 1, 0)
    kernels["mipmap_fwd_bicubic"] = lambda: capi.mipmap_grid_sampler_2d(tex, uvn, jac, 8, 1, 2)
    kernels["mipmap_bwd_bicubic"] = lambda: capi.mipmap_grid_sampler_2d_backward(gmo, tex, uvn, jac, 8, 1, 2)
    kernels["torch_grid_sample_fwd"] = lambda: th.nn.functional.grid_sample(tex[0], uvn, mode="bilinear", padding_mode="border", align_corners=False)
th.cuda.synchronize()
for flags in [int(x) for x in a.flags.split(",")]:
    set_flags(flags)
    for name, fn in kernels.items():
        if a.only and name not in a.only.split(","):
            continue
        ev0, ev1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        fn()
        ev0.record()
        for _ in range(a.reps):
            fn()
        ev1.record()
        th.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / a.reps
        rate = f"   {ALG_BYTES[name] / ms / 1e9:7.2f} TB/s on {ALG_BYTES[name] / 1e9:.3f} GB algorithmic = {ALG_BYTES[name] / ms / 1e9 / 8.0:.3f} of 8 TB/s" if name in ALG_BYTES and flags == 0 else ""
        print(f"{name}{'' if flags == 0 else f' [flags={flags}]'}: {ms:.3f} ms{rate}")
set_flags(0)

if os.environ.get("DRTK_ABLATE"):
    L = capi.lib()
    fn = kernels[os.environ["DRTK_ABLATE"]]
    for flags in [int(x) for x in os.environ.get('DRTK_ABLATE_FLAGS', '0,1,32,64,96,16').split(',')]:
        set_flags(flags)
        fn()
        th.cuda.synchronize()
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        th.cuda.synchronize()
        print(f"ablate {os.environ['DRTK_ABLATE']} flags={flags:2d}: {e0.elapsed_time(e1) / 5:.3f} ms")
    set_flags(0)

if a.check:
    ag, bg = capi.interpolate_backward(go, attr, vi, index, bary, True, True)
    N, V, C = attr.shape
    fg = index != -1
    tri = vi.long()[index.clamp(min=0).long()] if vi.dim() == 2 else None
    assert tri is not None
    worst = 0.0
    for n in range(N):
        ref = th.zeros(V, C, device=dev, dtype=th.float64)
        m = fg[n].reshape(-1)
        gn = go[n].reshape(C, -1).t()[m].double()
        refb = th.zeros(3, H * W, device=dev, dtype=th.float64)
        for k in range(3):
            ids = tri[n].reshape(-1, 3)[m, k]
            ref.index_add_(0, ids, gn * bary[n, k].reshape(-1)[m].double()[:, None])
            refb[k, m] = (gn * attr[n].double()[ids]).sum(1)
        e_a = (ag[n].double() - ref).abs().max().item() / ref.abs().max().item()
        e_b = (bg[n].reshape(3, -1).double() - refb).abs().max().item() / refb.abs().max().item()
        worst = max(worst, e_a, e_b)
    print(f"check interpolate_backward: worst error / max magnitude = {worst:.2e}")
    assert worst < 2e-5, worst

if a.dump:
    outs = [capi.interpolate(attr, vi, index, bary)[:1], capi.edge_grad_backward(v, img, index, vi, go)[:1], capi.edge_grad_backward_fused(v, img, index, vi, bary, go)[:1],
            capi.render_backward(v, vi, index, gd, gb)[:1], capi.rasterize(v, vi, H, W)[0][:1],
            capi.rasterize(v, vi, H, W)[1][:1].float() + 2]
    th.save([t.cpu() for t in outs], a.dump)

#!/usr/bin/env python3
"""Shapes OTHER than the benchmark's: what the kernels do where they were not tuned.

  --what interp_c   interpolate backward, C in {4,8,12,16,24,32,64} x {both, attr-only, bary-only}, 8 x 2048^2 (or --res):
                    ms, GB/s and fraction of the 8 TB/s HBM peak on SURVEY 8d's bytes (4C+28 B/px with the bary
                    gradient, 4C+16 without) -> profiles/rNN/interp_bwd_by_C.json
  --what raster     rasterize over regimes of triangle size (camera distance 3 / 1.5 / 1.1, a 10k mesh at 2048^2 and
                    4096^2, screen-filling quads, a 1M mesh at 512^2): per-kernel times, covered pixels, the share of
                    triangles that went to the per-view "big" list -> profiles/rNN/raster_regimes.json
  --what f64        every C-ABI kernel of the path at the bench shape in f64 beside f32, and at W = 2046 (scalar paths)
                    -> profiles/rNN/f64_and_odd_width.json

`--lib path.so` times another build of the library (same-box A/B).  Output: one JSON document on stdout (and --out)."""
import argparse
import json
import math
import os
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drtk_amd import capi  # noqa: E402
from drtk_amd import synthetic as S  # noqa: E402
from drtk_amd.transform import transform  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--what", default="interp_c,raster,f64")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--views", type=int, default=8)
ap.add_argument("--res", type=int, default=2048)
ap.add_argument("--mesh", default="100k")
ap.add_argument("--channels", default="4,8,12,16,24,32,64")
ap.add_argument("--lib", default="")
ap.add_argument("--out", default="")
ap.add_argument("--split-dir", default="", help="also write interp_bwd_by_C.json / raster_regimes.json / f64_and_odd_width.json there")
ap.add_argument("--grads", default="both,attr_only,bary_only")
ap.add_argument("--dtypes", default="f32", help="interp_c: f32, f64 or f32,f64")
ap.add_argument("--misalign", action="store_true", help="interp_c: the attribute tensor one element into a flat buffer (rows only element-aligned)")
ap.add_argument("--flags", type=int, default=0, help="ablation mask (needs profiles/libdrtk_amd_ablate.so: python drtk_amd/build.py --ablation)")
a = ap.parse_args()
if a.lib:
    capi.use_profiling_library(os.path.abspath(a.lib))
elif a.flags:
    capi.use_profiling_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdrtk_amd_ablate.so"))
dev = "cuda:0"


def set_flags(f):
    """ablation mask for the kernels UNDER TEST only: the masks are global to the library (flag 8 also empties the
    rasterizer's lists), so scenes are always prepared with 0"""
    if a.flags:
        capi.lib().drtk_amd_debug_set_flags(f)


PEAK = 8000.0  # GB/s, MI355X_MICROARCH.md


def timed(fn, reps):
    """median over three rounds of `reps` calls (a round now and then catches an allocator or clock event: 9 ms once)"""
    fn()
    th.cuda.synchronize()
    rounds = []
    for _ in range(3):
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        th.cuda.synchronize()
        rounds.append(e0.elapsed_time(e1) / reps)
    return sorted(rounds)[1]


def kernel_times(fn, reps):
    """{kernel name: ms per call} from the library's own HIP events around every launch."""
    fn()
    th.cuda.synchronize()
    capi.kernel_timing_begin()
    for _ in range(reps):
        fn()
    th.cuda.synchronize()
    rep = capi.kernel_timing_report()
    out = {}
    for k, (_count, total_ms) in rep.items():
        name = k.split("<")[0].strip("()")
        out[name] = out.get(name, 0.0) + total_ms / reps
    return out


def scene(views, mesh, res, distance=3.0, dtype=th.float32, lobes=0.05):
    nl, no = S.MESH_SIZES[mesh]
    v, vi = S.uv_sphere(nl, no, lobes=lobes, device=dev, dtype=dtype)
    campos, camrot, focal, princpt = S.ring_cameras(views, res, res, distance=distance, device=dev, dtype=dtype)
    v_pix = transform(v[None].expand(views, -1, -1), campos, camrot, focal, princpt)
    return v_pix.contiguous(), vi.contiguous()


result = {"device": th.cuda.get_device_name(0), "lib": a.lib or "product"}

if "interp_c" in a.what:
    H = W = a.res
    v, vi = scene(a.views, a.mesh, a.res)
    _, index = capi.rasterize(v, vi, H, W)
    _, bary = capi.render(v, vi, index)
    cov = (index != -1).float().mean().item()
    rows = []
    for dname in a.dtypes.split(","):
      dt = th.float64 if dname == "f64" else th.float32
      es = 8 if dname == "f64" else 4
      v_d, bary_d = (v.double(), None) if dname == "f64" else (v, bary)
      if dname == "f64":
          _, bary_d = capi.render(v_d, vi, index)
      for C in [int(c) for c in a.channels.split(",")]:
        attr = S.random_attributes(a.views, v.shape[1], C, shared=False, device=dev).to(dt)
        if a.misalign:
            flat = th.empty(attr.numel() + 1, device=dev, dtype=dt)
            flat[1:].copy_(attr.reshape(-1))
            attr = flat[1:].view_as(attr)
        g = th.Generator(device=dev).manual_seed(C)
        go = (th.rand(a.views, C, H, W, device=dev, generator=g) * 2 - 1).to(dt)
        px = a.views * H * W
        for name, hv, hb in (("both", True, True), ("attr_only", True, False), ("bary_only", False, True)):
            if name not in a.grads.split(","):
                continue
            set_flags(a.flags)
            ms = timed(lambda: capi.interpolate_backward(go, attr, vi, index, bary_d, hv, hb), a.reps)
            set_flags(0)
            bpp = es * C + 3 * es + 4 + (3 * es if hb else 0)
            gbps = bpp * px / ms / 1e6
            rows.append({"dtype": dname, "C": C, "grads": name, "ms": round(ms, 4), "bytes_per_px": bpp, "GBps": round(gbps, 1),
                         "frac": round(gbps / PEAK, 3), "us_per_channel": round(1e3 * ms / C, 2)})
            print(rows[-1], file=sys.stderr)
        del attr, go
    result["interp_bwd_by_C"] = {"views": a.views, "res": a.res, "mesh": a.mesh, "coverage": round(cov, 3), "peak_GBps": PEAK,
                                 "bytes": "SURVEY 8d: grad_out C + bary 3 elements + index 4 B read (+ bary_grad 3 elements written) per pixel, every pixel counted: f32 4C + 16 (+12), f64 8C + 28 (+24)", "rows": rows}

if "raster" in a.what:
    def quads(views, res, n_side, dtype=th.float32):
        """n_side x n_side screen-filling quads (2 triangles each) at slowly varying depth, shared by all views."""
        xs = th.linspace(-0.5, res - 0.5, n_side + 1, dtype=th.float64)
        yy, xx = th.meshgrid(xs, xs, indexing="ij")
        z = 2.0 + 0.5 * (xx / res) + 0.25 * (yy / res)
        v = th.stack([xx, yy, z], -1).reshape(-1, 3)
        ii, jj = th.meshgrid(th.arange(n_side), th.arange(n_side), indexing="ij")
        s = n_side + 1
        v00, v01, v10, v11 = ii * s + jj, ii * s + jj + 1, (ii + 1) * s + jj, (ii + 1) * s + jj + 1
        vi = th.stack([th.stack([v00, v10, v11], -1), th.stack([v00, v11, v01], -1)], 2).reshape(-1, 3).to(th.int32)
        return v[None].expand(views, -1, -1).to(dtype).contiguous().to(dev), vi.to(dev)

    regimes = [
        ("bench: 100k mesh, distance 3, 8 x 2048^2", lambda: scene(8, "100k", 2048, 3.0), 2048),
        ("100k mesh, distance 1.5, 8 x 2048^2", lambda: scene(8, "100k", 2048, 1.5), 2048),
        ("100k mesh, distance 1.1, 8 x 2048^2", lambda: scene(8, "100k", 2048, 1.1), 2048),
        ("10k mesh, distance 3, 8 x 2048^2", lambda: scene(8, "10k", 2048, 3.0), 2048),
        ("10k mesh, distance 3, 2 x 4096^2", lambda: scene(2, "10k", 4096, 3.0), 4096),
        ("10k mesh, distance 1.1, 2 x 4096^2", lambda: scene(2, "10k", 4096, 1.1), 4096),
        ("2 x 2 screen-filling quads, 8 x 2048^2", lambda: quads(8, 2048, 2), 2048),
        ("16 x 16 screen-filling quads, 8 x 2048^2", lambda: quads(8, 2048, 16), 2048),
        ("64 x 64 screen-filling quads, 2 x 4096^2", lambda: quads(2, 4096, 64), 4096),
        ("1M mesh, distance 3, 8 x 512^2", lambda: scene(8, "1M", 512, 3.0), 512),
    ]
    rows = []
    bench_ns_per_px = None
    for name, make, res in regimes:
        v, vi = make()
        N = v.shape[0]
        ms = timed(lambda: capi.rasterize(v, vi, res, res), a.reps)
        kt = kernel_times(lambda: capi.rasterize(v, vi, res, res), a.reps)
        _, index = capi.rasterize(v, vi, res, res)
        covered = int((index != -1).sum().item())
        # triangles whose clamped bbox spans more than 4 tiles of 64 x 64 (the binning's "big" list), counted from v
        f = vi.long()
        p = v[:, f]  # [N,F,3,3]
        mn = p[..., :2].amin(2).clamp(min=0)
        mx = p[..., :2].amax(2).clamp(max=res - 1)
        tiles = ((mx[..., 0] // 64 - mn[..., 0] // 64 + 1).clamp(min=0) * (mx[..., 1] // 64 - mn[..., 1] // 64 + 1).clamp(min=0))
        big = int((tiles > 4).sum().item())
        ns_per_px = 1e6 * ms / max(covered, 1)
        if bench_ns_per_px is None:
            bench_ns_per_px = ns_per_px
        rows.append({"regime": name, "views": N, "F": int(vi.shape[0]), "res": res, "ms": round(ms, 4),
                     "kernels_ms": {k: round(x, 4) for k, x in kt.items()}, "covered_px": covered,
                     "big_triangles_all_views": big, "ns_per_triangle_view": round(1e6 * ms / max(N * int(vi.shape[0]), 1), 4),
                     "ns_per_covered_px": round(ns_per_px, 4),
                     "vs_bench_per_covered_px": round(ns_per_px / bench_ns_per_px, 2)})
        print(rows[-1], file=sys.stderr)
        del v, vi, index, p
    result["raster_regimes"] = {"rows": rows}

if "f64" in a.what:
    rows = []
    for label, res_w, dtype in (("f32 2048x2048", 2048, th.float32), ("f64 2048x2048", 2048, th.float64), ("f32 2048x2046", 2046, th.float32)):
        Hh, Ww = 2048, res_w
        nl, no = S.MESH_SIZES["100k"]
        vw, vi = S.uv_sphere(nl, no, lobes=0.05, device=dev, dtype=dtype)
        campos, camrot, focal, princpt = S.ring_cameras(8, 2048, 2048, device=dev, dtype=dtype)
        v = transform(vw[None].expand(8, -1, -1), campos, camrot, focal, princpt).contiguous()
        attr = S.random_attributes(8, v.shape[1], 16, shared=False, device=dev, dtype=dtype)
        _, index = capi.rasterize(v, vi, Hh, Ww)
        depth, bary = capi.render(v, vi, index)
        img = capi.interpolate(attr, vi, index, bary) * (index != -1)[:, None]
        g = th.Generator(device=dev).manual_seed(0)
        go = (th.rand(img.shape, device=dev, generator=g) * 2 - 1).to(dtype)
        gd = th.rand(depth.shape, device=dev, generator=g).to(dtype)
        gb = th.rand(bary.shape, device=dev, generator=g).to(dtype)
        ks = {
            "rasterize": lambda: capi.rasterize(v, vi, Hh, Ww),
            "render": lambda: capi.render(v, vi, index),
            "interpolate": lambda: capi.interpolate(attr, vi, index, bary),
            "edge_grad_backward_fused": lambda: capi.edge_grad_backward_fused(v, img, index, vi, bary, go),
            "edge_grad_backward": lambda: capi.edge_grad_backward(v, img, index, vi, go),
            "interpolate_backward": lambda: capi.interpolate_backward(go, attr, vi, index, bary, True, True),
            "render_backward": lambda: capi.render_backward(v, vi, index, gd, gb),
        }
        row = {"shape": label}
        for k, fn in ks.items():
            row[k] = round(timed(fn, a.reps), 4)
        rows.append(row)
        print(row, file=sys.stderr)
        del v, attr, index, depth, bary, img, go, gd, gb
    base = rows[0]
    ratios = [{"shape": r["shape"], **{k: round(r[k] / base[k], 2) for k in r if k != "shape"}} for r in rows[1:]]
    result["f64_and_odd_width"] = {"ms": rows, "ratio_to_f32_2048": ratios}

txt = json.dumps(result, indent=1)
print(txt)
if a.split_dir:  # one file per section, as committed under profiles/rNN/
    os.makedirs(a.split_dir, exist_ok=True)
    for key, name in (("interp_bwd_by_C", "interp_bwd_by_C.json"), ("raster_regimes", "raster_regimes.json"), ("f64_and_odd_width", "f64_and_odd_width.json")):
        if key in result:
            with open(os.path.join(a.split_dir, name), "w") as f:
                f.write(json.dumps({"device": result["device"], "lib": result["lib"], key: result[key]}, indent=1) + "\n")
if a.out:
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        f.write(txt + "\n")

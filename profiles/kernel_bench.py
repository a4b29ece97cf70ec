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
a = ap.parse_args()
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
th.cuda.synchronize()
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
    print(f"{name}: {ev0.elapsed_time(ev1) / a.reps:.3f} ms")

if os.environ.get("DRTK_ABLATE"):
    L = capi.lib()
    fn = kernels[os.environ["DRTK_ABLATE"]]
    for flags in [int(x) for x in os.environ.get('DRTK_ABLATE_FLAGS', '0,1,32,64,96,16').split(',')]:
        L.drtk_amd_debug_set_flags(flags)
        fn()
        th.cuda.synchronize()
        e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        th.cuda.synchronize()
        print(f"ablate {os.environ['DRTK_ABLATE']} flags={flags:2d}: {e0.elapsed_time(e1) / 5:.3f} ms")
    L.drtk_amd_debug_set_flags(0)

#!/usr/bin/env python3
"""Where do tile_raster's workgroups spend their clocks?  Needs the ablation build (python drtk_amd/build.py --ablation):
thread 0 of every workgroup adds the shader clocks of each phase of each work item to a device counter
(csrc/rasterize.hip DRTK_PHASE).  Prints clocks per phase as a share of all workgroup-clocks of the launch.
    python3 profiles/raster_phases.py [--mesh 100k] [--res 2048] [--views 8]"""
import argparse
import ctypes
import os
import sys

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from drtk_amd import capi  # noqa: E402
from drtk_amd import synthetic as S  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mesh", default="100k")
ap.add_argument("--res", type=int, default=2048)
ap.add_argument("--views", type=int, default=8)
a = ap.parse_args()
capi.use_profiling_library(os.path.join(ROOT, "profiles", "libdrtk_amd_ablate.so"))
L = capi.lib()
L.drtk_amd_debug_set_flags(256)  # switch the phase clocks on
nl, no = S.MESH_SIZES[a.mesh]
v, vi = S.sphere_views(a.views, nl, no, a.res, a.res, lobes=0.05, device="cuda:0")
buf = (ctypes.c_ulonglong * 16)()
capi.rasterize(v, vi, a.res, a.res)
L.drtk_amd_debug_read_phases(buf)
reps = 5
e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    capi.rasterize(v, vi, a.res, a.res)
e1.record()
th.cuda.synchronize()
L.drtk_amd_debug_read_phases(buf)
names = ["queue pop", "clear tile", "first group (wave 0's share)", "big triangles", "wait: other waves' first group", "block-farthest reduction",
         "second group (wave 0's share)", "wait: other waves' second group", "unpack + store (issue)", "wait: other waves' stores"]
tot = sum(buf[i] for i in range(10))
print(f"rasterize {e0.elapsed_time(e1) / reps:.3f} ms per call (instrumented build); workgroup-clocks per call {tot / reps / 1e6:.1f} M")
for i, nme in enumerate(names):
    print(f"  {nme:36s} {100.0 * buf[i] / tot:5.1f} %   {buf[i] / reps / 1e6:8.2f} Mclk")
# row balance of the raster steps (flag 512): how many 16-pixel passes the waves run against what their four rows need
L.drtk_amd_debug_set_flags(512)
capi.rasterize(v, vi, a.res, a.res)
L.drtk_amd_debug_read_phases(buf)
if buf[12]:
    print(f"raster steps {buf[12]}: passes run {buf[10]} (mean {buf[10] / buf[12]:.2f} per step), passes needed by the four rows {buf[11]} "
          f"(mean {buf[11] / buf[12] / 4:.2f} per row and step) -> lane-row utilisation {buf[11] / (4.0 * buf[10]):.3f}")

#!/usr/bin/env python3
"""Reads the rasterizer's work-queue words back after one call (layout of csrc/rasterize.hip make_layout): number of work
items and what every shard's head ended at.  python3 profiles/queue_probe.py [--lib X.so]"""
import argparse, os, sys
import torch as th
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drtk_amd import capi, synthetic as S
ap = argparse.ArgumentParser(); ap.add_argument("--lib", default=""); a = ap.parse_args()
if a.lib: capi.use_profiling_library(os.path.abspath(a.lib))
N, H, W = 8, 2048, 2048
nl, no = S.MESH_SIZES["100k"]
v, vi = S.sphere_views(N, nl, no, H, W, lobes=0.05, device="cuda:0")
nb = capi.rasterize_workspace_bytes(N, vi.shape[0], H, W)
ws = th.zeros(nb, dtype=th.uint8, device="cuda:0")
import ctypes
depth = th.full((N, H, W), -7.0, dtype=th.float32, device="cuda:0")
index = th.full((N, H, W), -7, dtype=th.int32, device="cuda:0")
vi_c, vi_sN, F = capi._vi(vi, N)
rc = capi.lib().drtk_amd_rasterize(ctypes.c_int(capi._dt(v)), capi._p(v), capi._p(vi_c), capi._i(N), capi._i(v.shape[1]), capi._i(F), capi._i(vi_sN),
                                   capi._i(H), capi._i(W), ctypes.c_int(0), capi._p(depth), capi._p(index), capi._p(ws), ctypes.c_size_t(ws.numel()),
                                   capi._stream(v, None))
th.cuda.synchronize()
untouched = (index == -7).view(N, H // 64, 64, W // 64, 64).any(dim=4).any(dim=2)
print("rc", rc, "tiles with unwritten pixels:", int(untouched.sum()), "of", untouched.numel())
tiles = N * (H // 64) * (W // 64)
al = lambda x: (x + 255) // 256 * 256
off_queue = al(8 * tiles) * 2 + al(4 * N) + al(16 * N)
q = ws[off_queue:off_queue + 4 * 32 * 9].view(th.int32).cpu()
heads = [int(q[32 * (1 + s)]) for s in range(8)]
print("lib", a.lib or "main", "n_items", int(q[1]), "heads", heads, "sum", sum(heads), "q[0]", int(q[0]))

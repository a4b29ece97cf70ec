"""Sampler kernels on the textured workload's own inputs (bench.py --workload textured: 1M-tri sphere, 4096^2, its uv
atlas, the Jacobian of screen_space_uv_derivative, fp32 copies of the fp16 leaves) -- for same-box A/B of kernel
variants (--lib) and the ablation build's masks (--flags).  Prints fwd / bwd ms per call and, with --stats, what the
taps look like (count per pixel, levels)."""
import argparse
import os
import sys

import torch as th

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from drtk_amd import capi, synthetic as S  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--mesh", default="1M")
ap.add_argument("--res", type=int, default=4096)
ap.add_argument("--views", type=int, default=2)
ap.add_argument("--tex", type=int, default=4096)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--channels", type=int, default=3, help="texture channels (3 = the textured configuration; 8 / 16: neural textures -- the wave-private backward kernel)")
ap.add_argument("--bicubic", action="store_true")
ap.add_argument("--padding", default="border", choices=["zeros", "border", "reflection"], help="padding mode (the textured configuration samples with border)")
ap.add_argument("--uv", action="store_true", help="also time screen_space_uv_derivative on the same scene (through the C ABI: --lib applies)")
ap.add_argument("--f64", action="store_true", help="the sampler's inputs in double (the reference dispatches float and double alike)")
ap.add_argument("--flags", default="0")
ap.add_argument("--lib", default="")
ap.add_argument("--stats", action="store_true")
ap.add_argument("--rounds-stats", action="store_true", help="ablation build: how many tiles of the tiled backward enter each further round, how many taps stay pending")
ap.add_argument("--leftover-dump", default="", help="ablation build: save {view, tile, thread, level, x, y} of the (tap, level) pairs that end in global memory after the last round (npz)")
ap.add_argument("--tile-phases", action="store_true", help="with --tile-times: the phases of a tile (inputs there / windows placed / taps done / end) instead of its start and end")
ap.add_argument("--tile-times", default="", help="ablation build: save {view, tile, further rounds, start, end (10 ns ticks)} of every tile of the lean backward that had upstream gradient (npz)")
ap.add_argument("--dump", default="", help="save the backward's outputs (compare two libraries with --compare A B)")
ap.add_argument("--compare", nargs=2, default=None)
a = ap.parse_args()
if a.compare:
    A, B = th.load(a.compare[0]), th.load(a.compare[1])
    worst = max((x.double() - y.double()).abs().max().item() / y.abs().max().item() for x, y in zip(A, B))
    print(f"compare {a.compare[0]} {a.compare[1]}: worst difference / max magnitude = {worst:.2e}")
    sys.exit(0 if worst < 2e-5 else 1)
ABLATE = a.flags != "0" or a.rounds_stats or bool(a.leftover_dump) or bool(a.tile_times)
if a.lib:
    capi.use_profiling_library(os.path.abspath(a.lib))
elif ABLATE:
    capi.use_profiling_library(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdrtk_amd_ablate.so"))
import drtk_amd  # noqa: E402

dev = "cuda:0"
H = W = a.res
nl, no = S.MESH_SIZES[a.mesh]
v_world, vi = S.uv_sphere(nl, no, lobes=0.05, device=dev)
campos, camrot, focal, princpt = S.ring_cameras(a.views, W, H, device=dev)
vt, vti = S.uv_sphere_atlas(nl, no, device=dev)
tex = [t.expand(a.views, -1, -1, -1).contiguous() for t in S.texture_pyramid(1, a.channels, a.tex, device=dev)]
v_pix = drtk_amd.transform(v_world[None], campos, camrot, focal, princpt)
with th.no_grad():
    index = drtk_amd.rasterize(v_pix, vi, H, W)
    _, bary = drtk_amd.render(v_pix, vi, index)
    mask = index != -1
    vtn = vt[None].expand(a.views, -1, -1).contiguous()
    uv = drtk_amd.interpolate(vtn, vti, index, bary)
    import warnings
    warnings.simplefilter("ignore")
    jac = drtk_amd.screen_space_uv_derivative(v_world[None].expand(a.views, -1, -1), vtn, vi, vti, index, bary, mask, campos, camrot, focal)
    grid = ((uv.permute(0, 2, 3, 1) * 2 - 1) * mask[..., None]).contiguous()
g = th.Generator(device=dev).manual_seed(0)
go = (th.rand(a.views, a.channels, H, W, device=dev, generator=g) * 2 - 1) * mask[:, None]
print(f"coverage {mask.float().mean().item():.3f}")

if a.stats:
    j = jac.reshape(-1, 2, 2)[mask.reshape(-1)].double()
    size = th.tensor([a.tex, a.tex], device=dev, dtype=th.float64)
    px = (j[:, 0] * size).norm(dim=1)
    py = (j[:, 1] * size).norm(dim=1)
    pmax, pmin = th.maximum(px, py), th.minimum(px, py)
    n = th.clamp(th.ceil(pmax / pmin.clamp(min=1e-12)), 1, 8)
    lod = th.log2((pmax / n).clamp(min=1e-12))
    print(f"taps/pixel: mean {n.mean().item():.2f}  hist {th.bincount(n.long(), minlength=9).tolist()}")
    print(f"lod: mean {lod.mean().item():.2f}  <0: {(lod < 0).float().mean().item():.3f}  hist(floor,-2..6) "
          f"{th.histc(lod.floor().clamp(-2, 6), 9, -2, 7).long().tolist()}")

    # how often neighbouring pixels hit the same texel (same-address LDS adds are serialised): the centre tap's north-west
    # texel on level 0, per 16-pixel row segment and per 16 x 4 wave footprint, foreground segments only
    tx = th.floor((grid[..., 0] * 0.5 + 0.5) * a.tex - 0.5).long()
    ty = th.floor((grid[..., 1] * 0.5 + 0.5) * a.tex - 0.5).long()
    key = (ty * a.tex + tx)
    k16 = key.reshape(a.views, H, W // 16, 16)
    m16 = mask.reshape(a.views, H, W // 16, 16).all(-1)
    runs = 1 + (k16[..., 1:] != k16[..., :-1]).sum(-1)
    print(f"distinct texels per 16-pixel row segment (horizontal runs): mean {runs[m16].float().mean().item():.2f} of 16")
    kw = key.reshape(a.views, H // 4, 4, W // 16, 16).permute(0, 1, 3, 2, 4).reshape(a.views, H // 4, W // 16, 64)
    mw = mask.reshape(a.views, H // 4, 4, W // 16, 16).permute(0, 1, 3, 2, 4).reshape(a.views, H // 4, W // 16, 64).all(-1)
    ks = kw[mw].sort(-1).values
    distinct = 1 + (ks[:, 1:] != ks[:, :-1]).sum(-1)
    print(f"distinct texels per 16 x 4 wave footprint: mean {distinct.float().mean().item():.2f} of 64; "
          f"after merging horizontal runs the rows still share {(runs.reshape(a.views, H // 4, 4, W // 16).sum(2)[mw].float().mean() / distinct.float().mean()).item():.2f}x")


def set_flags(f):
    if ABLATE:
        capi.lib().drtk_amd_debug_set_flags(f)


def timeit(fn):
    fn()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps):
        fn()
    e1.record()
    th.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps


if a.uv:
    vw = v_world[None].expand(a.views, -1, -1).contiguous()
    args_uv = (vw, vtn, vi, vti, index, bary, mask, campos, camrot, focal)
    print(f"screen_space_uv_derivative: {timeit(lambda: capi.screen_space_uv_derivative(*args_uv)):.4f} ms")
if a.f64:
    tex = [t.double() for t in tex]
    grid, jac, go = grid.double(), jac.double(), go.double()
for flags in [int(x) for x in a.flags.split(",")]:
    set_flags(flags)
    MODE = 2 if a.bicubic else 0
    PADM = {"zeros": 0, "border": 1, "reflection": 2}[a.padding]
    f = timeit(lambda: capi.mipmap_grid_sampler_2d(tex, grid, jac, 8, PADM, MODE))
    b = timeit(lambda: capi.mipmap_grid_sampler_2d_backward(go, tex, grid, jac, 8, PADM, MODE))
    print(f"flags={flags} C={a.channels}{' bicubic' if a.bicubic else ''}{' f64' if a.f64 else ''} {a.padding}: mipmap fwd {f:.3f} ms   bwd (incl. zero-fill of the pyramid) {b:.3f} ms")
set_flags(0)

if a.rounds_stats:
    import ctypes

    L = capi.lib()
    buf = (ctypes.c_ulonglong * 16)()
    L.drtk_amd_debug_read_mip_stats(buf)  # clear
    L.drtk_amd_debug_set_flags(64)
    capi.mipmap_grid_sampler_2d_backward(go, tex, grid, jac, 8, 1, 0)
    th.cuda.synchronize()
    L.drtk_amd_debug_read_mip_stats(buf)
    L.drtk_amd_debug_set_flags(0)
    st = list(buf)
    print(f"tiles with gradient {st[0]}; entering round 1..7: {st[1:8]}; (tap, level) pairs {st[8]}, pending after the first pass {st[9]} "
          f"({100.0 * st[9] / max(st[8], 1):.2f} %), to global memory after the last round {st[10]} (by level above the tile's finest: {st[12:16]}); "
          f"tiles cut short as hopeless {st[11]}")

if a.dump:
    gl, gg = capi.mipmap_grid_sampler_2d_backward(go, tex, grid, jac, 8, 1, 0)
    fw = capi.mipmap_grid_sampler_2d(tex, grid, jac, 8, 1, 0)
    th.save([t[:1].cpu() for t in gl[:3]] + [gg[:1].cpu(), fw[:1].cpu()], a.dump)

if a.leftover_dump:
    import ctypes

    import numpy as np

    L = capi.lib()
    buf = (ctypes.c_uint * (4 << 20))()
    cnt = ctypes.c_uint(0)
    L.drtk_amd_debug_read_mip_dump(buf, ctypes.byref(cnt))  # clear
    L.drtk_amd_debug_set_flags(64)
    capi.mipmap_grid_sampler_2d_backward(go, tex, grid, jac, 8, 1, 0)
    th.cuda.synchronize()
    L.drtk_amd_debug_read_mip_dump(buf, ctypes.byref(cnt))
    L.drtk_amd_debug_set_flags(0)
    d = np.frombuffer(buf, dtype=np.uint32, count=4 * cnt.value).reshape(-1, 4).copy()
    np.savez_compressed(a.leftover_dump, view=d[:, 0] >> 24, tile=d[:, 0] & 0xFFFFFF, thread=d[:, 1] >> 16, level=d[:, 1] & 0xFFFF,
                        x=d[:, 2].astype(np.int32), y=d[:, 3].astype(np.int32), H=a.res, W=a.res)
    print(f"leftover dump: {cnt.value} pairs -> {a.leftover_dump}")

if a.tile_times:
    import ctypes

    import numpy as np

    L = capi.lib()
    buf = (ctypes.c_uint * (4 << 20))()
    cnt = ctypes.c_uint(0)
    L.drtk_amd_debug_read_mip_dump(buf, ctypes.byref(cnt))  # clear
    L.drtk_amd_debug_set_flags((1 << 21) if a.tile_phases else (1 << 20))
    capi.mipmap_grid_sampler_2d_backward(go, tex, grid, jac, 8, 1, 0)
    th.cuda.synchronize()
    L.drtk_amd_debug_read_mip_dump(buf, ctypes.byref(cnt))
    L.drtk_amd_debug_set_flags(0)
    d = np.frombuffer(buf, dtype=np.uint32, count=4 * cnt.value).reshape(-1, 4).copy()
    np.savez_compressed(a.tile_times, view=d[:, 0] >> 24, tile=d[:, 0] & 0xFFFFFF, rounds=d[:, 1], t0=d[:, 2], t1=d[:, 3], H=a.res, W=a.res)
    print(f"tile times: {cnt.value} tiles -> {a.tile_times}")

"""Wide net over the rasterizer's binning machinery (bin_count / bin_scan / bin_fill / tile_raster work lists) at sizes
the small-scene fuzzers never reach: image sizes off every tile multiple up to 2048 x 1536, 1e3 - 3e5 triangles per view
in a random mix of tiny (1-4 px), medium (10-60 px) and large (200-2000 px) ones, depths quantised so that exact ties
occur (lowest id wins).  index_img and depth_img against the oracle, bit for bit.
usage: python tests/fuzz_raster_large.py [--first S] [--cases K]"""
import argparse
import os

os.environ.setdefault("DRTK_CAPI_POISON", "1")  # outputs of the ctypes binding pre-filled with NaN / sentinels (drtk_amd/capi.py _out)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch as th  # noqa: E402

import oracle as O  # noqa: E402
from drtk_amd import capi  # noqa: E402

DEV = "cuda:0"
SIZES = [(512, 512), (700, 1000), (1024, 1024), (1536, 2048), (2047, 333), (129, 2048), (1080, 1920), (17, 4096)]
COUNTS = [1000, 5000, 20000, 60000, 100000, 300000]


def make_case(seed):
    g = th.Generator().manual_seed(seed)
    r = lambda lo, hi: int(th.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    H, W = SIZES[r(0, len(SIZES) - 1)]
    N = r(1, 2)
    ntri = COUNTS[r(0, len(COUNTS) - 1)]
    n_large = r(0, 40)
    frac_medium = [0.0, 0.05, 0.3, 0.9][r(0, 3)]
    size = th.empty(N, ntri, 1, 1)
    u = th.rand(N, ntri, 1, 1, generator=g)
    tiny = 1.0 + 3.0 * th.rand(N, ntri, 1, 1, generator=g)
    medium = 10.0 + 50.0 * th.rand(N, ntri, 1, 1, generator=g)
    size = th.where(u < frac_medium, medium, tiny)
    if n_large:
        size[:, :n_large] = 200.0 + 1800.0 * th.rand(N, n_large, 1, 1, generator=g)
    ctr = th.rand(N, ntri, 1, 2, generator=g) * th.tensor([W + 40.0, H + 40.0]) - 20.0
    if r(0, 2) == 0:  # clustered: most triangles in a few tiles -> long per-tile lists
        k = r(1, 6)
        hot = th.rand(N, k, 1, 2, generator=g) * th.tensor([float(W), float(H)])
        pick = th.randint(0, k, (N, ntri), generator=g)
        ctr = th.where(th.rand(N, ntri, 1, 1, generator=g) < 0.8, hot[th.arange(N)[:, None], pick] + 30.0 * th.randn(N, ntri, 1, 2, generator=g), ctr)
    xy = ctr + (th.rand(N, ntri, 3, 2, generator=g) - 0.5) * size
    levels = [4, 64, 100000][r(0, 2)]
    z = 1.0 + th.randint(0, levels, (N, ntri, 1, 1), generator=g).float() / levels + (0.0 if r(0, 1) else 1.0) * 0.01 * th.rand(N, ntri, 3, 1, generator=g)
    v = th.cat([xy, z.expand(N, ntri, 3, 1)], -1).reshape(N, ntri * 3, 3).contiguous()
    vi = th.arange(ntri * 3, dtype=th.int32).view(ntri, 3)
    if r(0, 3) == 0:
        vi = vi[th.randperm(ntri, generator=g)].contiguous()  # ids not in storage order
    return dict(N=N, H=H, W=W, ntri=ntri, n_large=n_large, v=v, vi=vi, levels=levels)


def describe(c):
    return f"N={c['N']} H={c['H']} W={c['W']} triangles={c['ntri']} large={c['n_large']} depth levels={c['levels']}"


def run_case(c):
    d_o, i_o = O.rasterize(c["v"], c["vi"], c["H"], c["W"], nthreads=0)
    d_g, i_g = capi.rasterize(c["v"].to(DEV), c["vi"].to(DEV), c["H"], c["W"])
    th.cuda.synchronize()
    nd = int((i_g.cpu() != i_o).sum())
    assert nd == 0, f"index_img differs at {nd} pixels"
    assert th.equal(d_g.cpu(), d_o), "depth_img differs"
    return float((i_o != -1).float().mean())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=20)
    ap.add_argument("--first", type=int, default=0)
    a = ap.parse_args()
    bad, t0, cov = 0, time.time(), []
    for seed in range(a.first, a.first + a.cases):
        c = make_case(seed)
        try:
            cov.append(run_case(c))
        except Exception as e:
            bad += 1
            print(f"FAIL seed {seed}: {describe(c)}: {type(e).__name__}: {str(e)[:200]}", flush=True)
    print(f"coverage of the images: min {min(cov):.2f} max {max(cov):.2f}; {time.time() - t0:.0f} s" if cov else "")
    print(f"{a.cases - bad}/{a.cases} cases passed")
    sys.exit(1 if bad else 0)

"""The mipmap sampler on STRUCTURED sample positions: grid coordinates exactly on texel centres, texel edges and the
texture border of a power-of-two texture (and up to 1.5 textures outside, where the padding modes act), footprints at
exact powers of two (integer mip levels) and exact integer anisotropy ratios.  floor() of the unnormalised coordinate
picks the cell, floor(in / span) counts the reflections, ceil(ratio) the taps: every argument lands exactly ON the
step.  Forward, grid gradient and level gradients against the oracle in all padding / interpolation modes.
usage: python tests/fuzz_mipmap_snapped.py [--cases K]"""
import argparse
import os

os.environ.setdefault("DRTK_CAPI_POISON", "1")  # outputs of the ctypes binding pre-filled with NaN / sentinels (drtk_amd/capi.py _out)
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch as th  # noqa: E402

import fuzz_all_ops as FA  # noqa: E402,F401  (import paths)
import oracle as O  # noqa: E402
from f64_distance import assert_within_f64_distance  # noqa: E402
from drtk_amd import capi  # noqa: E402

DEV = "cuda:0"


def make_case(seed):
    g = th.Generator().manual_seed(seed)
    r = lambda lo, hi: int(th.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    tw, th_ = [8, 16, 32, 64][r(0, 3)], [8, 16, 32, 64][r(0, 3)]
    C, H, W = [1, 3, 4, 5][r(0, 3)], r(5, 40), r(5, 40)
    base = th.rand(1, C, th_, tw, generator=g)
    levels = [base]
    while min(levels[-1].shape[-2:]) > 1:
        levels.append(th.nn.functional.avg_pool2d(levels[-1], 2))
    levels = [lv.contiguous() for lv in levels]
    # positions in HALF texels: k/2 texel units -> centres (odd k) and edges (even k); normalised x = (k / tw) - 1
    kx = th.randint(-3 * tw, 5 * tw + 1, (1, H, W), generator=g).float()
    ky = th.randint(-3 * th_, 5 * th_ + 1, (1, H, W), generator=g).float()
    grid = th.stack([kx / tw - 1.0, ky / th_ - 1.0], -1).contiguous()
    # footprints: exact powers of two along one axis, an exact integer multiple along the other
    p = 2.0 ** th.randint(-1, 5, (1, H, W), generator=g).float()
    ratio = th.randint(1, 6, (1, H, W), generator=g).float()
    jac = th.zeros(1, H, W, 2, 2)
    swap = th.rand(1, H, W, generator=g) < 0.5
    jac[..., 0, 0] = th.where(swap, p * ratio, p) / tw
    jac[..., 1, 1] = th.where(swap, p, p * ratio) / th_
    gout = th.rand(1, C, H, W, generator=g) * 2 - 1
    return dict(levels=levels, grid=grid, jac=jac, gout=gout, desc=f"texture {th_}x{tw} C={C} output {H}x{W}")


def run_case(c):
    d = lambda x: x.to(DEV)  # noqa: E731
    for padding in (0, 1, 2):
        for mode in (0, 2):
            for align, force, clip in ((False, False, False), (True, False, True), (False, True, False)):
                args = (4, padding, mode, align, force, clip)
                want = O.mipmap_grid_sampler_2d(c["levels"], c["grid"], c["jac"], *args)
                got = capi.mipmap_grid_sampler_2d([d(t) for t in c["levels"]], d(c["grid"]), d(c["jac"]), *args)
                FA._close(got, want, f"forward padding={padding} mode={mode} flags={align, force, clip}")
                wl, wg = O.mipmap_grid_sampler_2d_backward(c["gout"], c["levels"], c["grid"], c["jac"], *args)
                gl, gg = capi.mipmap_grid_sampler_2d_backward(d(c["gout"]), [d(t) for t in c["levels"]], d(c["grid"]), d(c["jac"]), *args)
                FA._close(gg, wg, f"grad grid padding={padding} mode={mode} flags={align, force, clip}", atol=1e-4)
                # level gradients: against the same sums carried in double, as near as the oracle's own float32 run up to a
                # factor (tests/f64_distance.py).  A coarse level is a handful of texels that each collect thousands of
                # signed terms (a 1 x 1 level: every tap of every pixel), so the magnitude that was ACCUMULATED -- the
                # oracle's backward of |grad_out| -- enters the bound as well; a flipped cell / tap count is 0.1 - 1.
                # (The inputs are exactly representable, so the double run takes the same cells and tap counts.)
                al, _ = O.mipmap_grid_sampler_2d_backward(c["gout"].abs(), c["levels"], c["grid"], c["jac"], *args)
                dd = lambda t: t.double()  # noqa: E731
                wl64, _ = O.mipmap_grid_sampler_2d_backward(dd(c["gout"]), [dd(t) for t in c["levels"]], dd(c["grid"]), dd(c["jac"]), *args)
                for k, (a, b) in enumerate(zip(gl, wl)):
                    assert_within_f64_distance(a, b, wl64[k], f"grad level {k} padding={padding} mode={mode} flags={align, force, clip}",
                                               acc_magnitude=float(al[k].abs().max()))

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--first", type=int, default=0)
    a = ap.parse_args()
    bad = 0
    for seed in range(a.first, a.first + a.cases):
        c = make_case(seed)
        try:
            run_case(c)
        except Exception as e:
            bad += 1
            print(f"FAIL seed {seed}: {c['desc']}: {type(e).__name__}: {str(e)[:220]}", flush=True)
    print(f"{a.cases - bad}/{a.cases} cases passed")
    sys.exit(1 if bad else 0)

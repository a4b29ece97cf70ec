"""Every op of tests/fuzz_all_ops.py (forward bits, gradients at the 1e-5 bar, both edge-grad routes at max_dp_dr = 1e4
and 0.5) on LARGE two-object scenes: two intersecting lobed spheres at 512^2 - 1024^2 (and awkward aspect ratios),
3e3 - 6e4 triangles, 1-2 views; thousands of intersection pixels with |dp/dr| > 100 per case.  A parity check at scale.
It is NOT a test of the square-root correction (profiles/NOTES.md section 3): measured, the library with the old native root
passes these scenes too (60/60) -- a sign decided by rounding needs a pair of faces whose projected normals are parallel
up to rounding, which more pixels do not produce; the small low-poly scenes of fuzz_all_ops do (11 cases in 9000), and
test_edge_grad_sign_decisions_at_near_parallel_normals_follow_the_reference pins those.
usage: python tests/fuzz_large_scenes.py [--first S] [--cases K]"""
import argparse
import os

os.environ.setdefault("DRTK_CAPI_POISON", "1")  # outputs of the ctypes binding pre-filled with NaN / sentinels (drtk_amd/capi.py _out)
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch as th  # noqa: E402

import fuzz_all_ops as FA  # noqa: E402  (sets up the import paths)
from drtk_amd import synthetic as S  # noqa: E402

SIZES = [(512, 512), (768, 1024), (1024, 1024), (1000, 700), (2047, 333), (640, 1536)]


def make_case(seed):
    g = th.Generator().manual_seed(seed)
    r = lambda lo, hi: int(th.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    H, W = SIZES[r(0, len(SIZES) - 1)]
    N = r(1, 2)
    C = [1, 3, 8, 16, 17][r(0, 4)]
    dtype = th.float64 if r(0, 3) == 0 else th.float32
    v, vi = S.sphere_views(N, r(20, 120), r(30, 160), H, W, lobes=0.1 * r(0, 3), second_sphere=True)
    batched_vi = r(0, 5) == 0
    if batched_vi:
        vi = vi[None].repeat(N, 1, 1)
        if N > 1:
            vi[1] = vi[1].flip(-1)
    v = v.to(dtype).contiguous()
    attr = th.rand(N, v.shape[1], C, generator=g).to(dtype)
    go = (th.rand(N, C, H, W, generator=g) * 2 - 1).to(dtype)
    gd = (th.rand(N, H, W, generator=g) * 2 - 1).to(dtype)
    gb = (th.rand(N, 3, H, W, generator=g) * 2 - 1).to(dtype)
    return dict(N=N, H=H, W=W, C=C, dtype=dtype, kind=2, batched_vi=batched_vi, v=v, vi=vi.contiguous(), attr=attr, go=go, gd=gd, gb=gb)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=20)
    ap.add_argument("--first", type=int, default=0)
    a = ap.parse_args()
    bad, t0 = 0, time.time()
    for seed in range(a.first, a.first + a.cases):
        c = make_case(seed)
        try:
            FA.run_case(c)
        except Exception as e:
            bad += 1
            print(f"FAIL seed {seed}: {FA.describe(c)}: {type(e).__name__}: {str(e)[:220]}", flush=True)
    print(f"{a.cases - bad}/{a.cases} cases passed in {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)

"""The all-ops checks of tests/fuzz_all_ops.py on STRUCTURED coordinates: vertex x, y snapped to a 1, 1/2 or 1/4 pixel
grid (vertices on pixel centres and pixel edges, axis-aligned and 45-degree edges, shared edges that pass exactly
through pixel centres) and depths snapped to a few levels (exact ties).  Edge functions are then exactly zero at many
pixels and the reference's on-edge / top-left rules (rasterize_kernel.cu:69-166, edge_grad's pix_in_tri) decide -- the
case random floats almost never produce.  usage: python tests/fuzz_snapped.py [--first S] [--cases K]"""
import argparse
import os

os.environ.setdefault("DRTK_CAPI_POISON", "1")  # outputs of the ctypes binding pre-filled with NaN / sentinels (drtk_amd/capi.py _out)
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch as th  # noqa: E402

import fuzz_all_ops as FA  # noqa: E402


def make_case(seed):
    c = FA.make_case(seed)
    g = th.Generator().manual_seed(seed + 7)
    step = [1.0, 0.5, 0.25][int(th.randint(0, 3, (1,), generator=g))]
    v = c["v"].clone()
    v[..., :2] = th.round(v[..., :2] / step) * step
    if int(th.randint(0, 2, (1,), generator=g)):
        v[..., 2] = th.round(v[..., 2] * 8) / 8  # depth levels 1/8 apart: exact ties between triangles
        v[..., 2].clamp_(min=0.125)
    c["v"] = v.contiguous()
    c["step"] = step
    return c


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--first", type=int, default=0)
    a = ap.parse_args()
    bad = 0
    for seed in range(a.first, a.first + a.cases):
        c = make_case(seed)
        try:
            FA.run_case(c)
        except Exception as e:
            bad += 1
            print(f"FAIL seed {seed}: grid {c['step']} {FA.describe(c)}: {type(e).__name__}: {str(e)[:200]}", flush=True)
    print(f"{a.cases - bad}/{a.cases} cases passed")
    sys.exit(1 if bad else 0)

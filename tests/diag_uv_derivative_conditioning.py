"""Diagnostic: the f32 screen_space_uv_derivative kernel's per-pixel error against the f64 evaluation, measured in units of
the problem's own sensitivity -- how far the f64 result moves when every input coordinate (vertices, uvs) moves by half an
f32 ulp.  An evaluation that is forward-stable stays within a small multiple of that at EVERY pixel, however ill-conditioned
the pixel is; prints the distribution of that multiple over the fuzz cases (kernel and reference composite).
usage: python tests/diag_uv_derivative_conditioning.py [--first S] [--cases K]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
import fuzz_next_ops as FZ
import oracle as O
from drtk_amd import capi, transform
DEV = "cuda:0"


def sensitivity(c, index, bary, mask, truth):
    """max over two sign patterns of |f64 result on inputs moved by +-2^-24 relative - f64 result|, per pixel"""
    dd = lambda x: x.double()  # noqa: E731
    campos, camrot, focal = (dd(t) for t in c["cams"][:3])
    s = th.zeros(truth.shape[:-2], dtype=th.float64)
    for ph in (0, 1):
        sg = lambda t: 1.0 - 2.0 * ((th.arange(t.numel(), dtype=th.float64) + ph) % 2).reshape(t.shape)  # noqa: E731
        moved = O.screen_space_uv_derivative(dd(c["vN"]) * (1 + 2.0 ** -24 * sg(c["vN"])), dd(c["vt"]) * (1 + 2.0 ** -24 * sg(c["vt"])),
                                             c["vi"], c["vi"], index, dd(bary), mask, campos, camrot, focal)
        s = th.maximum(s, (moved - truth).abs().amax((-1, -2)))
    return s


def case_ratios(c):
    d = lambda x: x.to(DEV)  # noqa: E731
    vi, H, W = c["vi"], c["H"], c["W"]
    out = transform(d(c["vN"]), *(d(t) for t in c["cams"]))
    _, index = O.rasterize(out.detach().cpu(), vi, H, W)
    _, bary = O.render(out.detach().cpu(), vi, index)
    mask = (index != -1) & c["mask_keep"]
    if int(mask.sum()) == 0:
        return None
    campos, camrot, focal = c["cams"][0], c["cams"][1], c["cams"][2]
    want = O.screen_space_uv_derivative(c["vN"], c["vt"], vi, vi, index, bary, mask, campos, camrot, focal)
    truth = O.screen_space_uv_derivative(c["vN"].double(), c["vt"].double(), vi, vi, index, bary.double(), mask, campos.double(), camrot.double(), focal.double())
    got = capi.screen_space_uv_derivative(d(c["vN"]), d(c["vt"]), d(vi), d(vi), d(index), d(bary), d(mask), d(campos), d(camrot), d(focal)).cpu()
    sens = sensitivity(c, index, bary, mask, truth)
    px = truth.abs().amax((-1, -2))
    unit = (sens + 2.0 ** -23 * px)[mask].clamp_min(1e-300)  # the sensitivity, and one ulp of the value itself
    rg = ((got.double() - truth).abs().amax((-1, -2)))[mask] / unit
    rr = ((want.double() - truth).abs().amax((-1, -2)))[mask] / unit
    return rg, rr


ap = argparse.ArgumentParser(); ap.add_argument("--first", type=int, default=20000); ap.add_argument("--cases", type=int, default=600)
a = ap.parse_args()
G, R, n = [], [], 0
worst = (0.0, None)
for seed in range(a.first, a.first + a.cases):
    c = FZ.make_case(seed)
    if c["dtype"] != th.float32 or c["batched_vi"]:
        continue
    try:
        r = case_ratios(c)
    except Exception as ex:  # the reference composite raises on degenerate UV faces
        print(f"seed {seed}: reference raised {type(ex).__name__}"); continue
    if r is None:
        continue
    n += 1
    G.append(r[0]); R.append(r[1])
    if float(r[0].max()) > worst[0]:
        worst = (float(r[0].max()), seed)
import numpy as np
g, r = th.cat(G).numpy(), th.cat(R).numpy()
qs = [0.5, 0.9, 0.99, 0.999, 1.0]
print(f"{n} f32 cases, {g.size} pixels.  error / (sensitivity to half-ulp inputs + 1 ulp of the value), quantiles 50 / 90 / 99 / 99.9 / 100 %:")
print("   kernel   ", [round(float(x), 2) for x in np.quantile(g, qs)], f"(worst: seed {worst[1]})")
print("   composite", [round(float(x), 2) for x in np.quantile(r, qs)])

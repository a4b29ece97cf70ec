"""Diagnostic: is the f32 screen_space_uv_derivative kernel as accurate as the reference's f32 PyTorch composite?
For every f32 fuzz case (tests/fuzz_next_ops.py) with >= 20 foreground pixels: relative error against the f64
evaluation, per pixel, for both; prints the distribution over cases of  log10(q90_kernel / q90_composite)  and of the
medians.  Symmetric around 0 = same accuracy, case-to-case luck in either direction; shifted = a systematic gap.
usage: python tests/diag_uv_derivative_accuracy.py [--first S] [--cases K] [--show SEED]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th
import fuzz_next_ops as FZ
import oracle as O
from drtk_amd import capi, transform
DEV = "cuda:0"


def errors(c):
    d = lambda x: x.to(DEV)  # noqa: E731
    vi, H, W = c["vi"], c["H"], c["W"]
    out = transform(d(c["vN"]), *(d(t) for t in c["cams"]))
    v_pix = out.detach().cpu()
    _, index = O.rasterize(v_pix, vi, H, W)
    _, bary = O.render(v_pix, vi, index)
    mask = (index != -1) & c["mask_keep"]
    if int(mask.sum()) < 20:
        return None
    campos, camrot, focal = c["cams"][0], c["cams"][1], c["cams"][2]
    want = O.screen_space_uv_derivative(c["vN"], c["vt"], vi, vi, index, bary, mask, campos, camrot, focal)
    truth = O.screen_space_uv_derivative(c["vN"].double(), c["vt"].double(), vi, vi, index, bary.double(), mask, campos.double(), camrot.double(), focal.double())
    got = capi.screen_space_uv_derivative(d(c["vN"]), d(c["vt"]), d(vi), d(vi), d(index), d(bary), d(mask), d(campos), d(camrot), d(focal)).cpu()
    px = truth.abs().amax((-1, -2)).clamp_min(1e-30)
    rg = ((got.double() - truth).abs().amax((-1, -2)) / px)[mask]
    rr = ((want.double() - truth).abs().amax((-1, -2)) / px)[mask]
    return rg, rr, index[mask]


ap = argparse.ArgumentParser(); ap.add_argument("--first", type=int, default=10000); ap.add_argument("--cases", type=int, default=600)
ap.add_argument("--show", type=int, default=12589); a = ap.parse_args()
lr90, lrmed, n, worse4, better4 = [], [], 0, 0, 0
for seed in range(a.first, a.first + a.cases):
    c = FZ.make_case(seed)
    if c["dtype"] != th.float32 or c["batched_vi"]:
        continue
    try:
        e = errors(c)
    except Exception as ex:  # the reference composite raises on degenerate UV faces
        print(f"seed {seed}: reference raised {type(ex).__name__}"); continue
    if e is None:
        continue
    rg, rr, _ = e
    q90g, q90r = float(th.quantile(rg, 0.9)), float(th.quantile(rr, 0.9))
    mg, mr = float(rg.median()), float(rr.median())
    if min(q90g, q90r, mg, mr) <= 0:
        continue
    n += 1
    lr90.append(th.log10(th.tensor(q90g / q90r))); lrmed.append(th.log10(th.tensor(mg / mr)))
    worse4 += q90g > 4 * q90r; better4 += q90r > 4 * q90g
l9, lm = th.stack(lr90), th.stack(lrmed)
qs = th.tensor([0.01, 0.1, 0.5, 0.9, 0.99])
print(f"{n} f32 cases.  log10(q90 kernel / q90 composite): quantiles 1/10/50/90/99 % = {[round(float(x), 3) for x in th.quantile(l9, qs)]}  mean {float(l9.mean()):+.3f}")
print(f"             log10(median kernel / median composite):                  = {[round(float(x), 3) for x in th.quantile(lm, qs)]}  mean {float(lm.mean()):+.3f}")
print(f"cases where the kernel's q90 is > 4x the composite's: {worse4};  where the composite's is > 4x the kernel's: {better4}")
c = FZ.make_case(a.show); rg, rr, faces = errors(c)
print(f"seed {a.show}: {len(rg)} foreground px over {len(faces.unique())} faces; q90 kernel {float(th.quantile(rg, .9)):.3e} composite {float(th.quantile(rr, .9)):.3e}; "
      f"median kernel {float(rg.median()):.3e} composite {float(rr.median()):.3e}; px where the kernel is closer to f64: {int((rg < rr).sum())}, further: {int((rg > rr).sum())}")
worst = rg.argsort(descending=True)[:max(1, len(rg) // 10)]
fw = faces[worst]
print(f"   the worst 10 % of the kernel's pixels lie on faces {sorted(set(fw.tolist()))} ({len(set(fw.tolist()))} faces)")
for f in sorted(set(fw.tolist()))[:6]:
    m = faces == f
    print(f"   face {f}: {int(m.sum())} px, kernel rel err {float(rg[m].median()):.2e}, composite {float(rr[m].median()):.2e}")

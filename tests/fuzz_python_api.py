"""Wide net over the torch-operator shim and the `drtk_amd.*` Python API (autograd formulas, saved tensors, stride
handling, route selection) on the awkward shapes of tests/fuzz_all_ops.py.  Self-checking, two independent references:
  * forward outputs of the API == the C-ABI results, bit for bit (those are tied to the oracle by fuzz_all_ops);
  * gradients of the whole pipeline rasterize -> render -> interpolate -> mask -> edge_grad_estimator -> loss agree
    between route A (contiguous inputs, FUSED edge-grad backward) and route B (NON-CONTIGUOUS views of the same
    values, UNFUSED reference-shaped backward forced by an identity v_pix_img_hook) -- no backward kernel in common
    on the edge path -- at the 1e-5 bar (1e-10 relative in f64).
usage: python tests/fuzz_python_api.py [--first S] [--cases K]"""
import argparse
import os

os.environ.setdefault("DRTK_CAPI_POISON", "1")  # outputs of the ctypes binding pre-filled with NaN / sentinels (drtk_amd/capi.py _out)
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch as th  # noqa: E402

import drtk_amd  # noqa: E402
import fuzz_all_ops as FA  # noqa: E402
from drtk_amd import capi  # noqa: E402

DEV = "cuda:0"
STATS = []  # per case: (max |grad v_pix|, max |route A - route B|)


def strided(x):
    """The same values as a NON-contiguous view: every other element of the last dimension of a wider buffer."""
    x = x.to(DEV)
    if x.numel() == 0:
        return x
    big = th.zeros(*x.shape[:-1], x.shape[-1] * 2, dtype=x.dtype, device=DEV)
    view = big[..., ::2]
    view.copy_(x)
    assert not view.is_contiguous() or x.shape[-1] == 1
    return view


def pipeline(c, v, attr, vi, hook):
    H, W = c["H"], c["W"]
    index = drtk_amd.rasterize(v, vi, height=H, width=W)
    depth, bary = drtk_amd.render(v, vi, index)
    img = drtk_amd.interpolate(attr, vi, index, bary)
    img_m = img * (index != -1)[:, None]
    out = drtk_amd.edge_grad_estimator(v_pix=v, vi=vi, bary_img=bary, img=img_m, index_img=index, v_pix_img_hook=hook)
    loss = (out * c["go"].to(DEV)).sum() + (depth * c["gd"].to(DEV)).sum() + (bary * c["gb"].to(DEV)).sum()
    gv, ga = th.autograd.grad(loss, (v, attr), allow_unused=True)
    return index, depth, bary, img, gv, ga


def run_case(c):
    tight = c["dtype"] == th.float64
    tol = dict(atol=1e-12, rtol=1e-10) if tight else dict(atol=1e-5, rtol=1e-5)
    vi = c["vi"].to(DEV)
    vA, aA = c["v"].to(DEV).requires_grad_(True), c["attr"].to(DEV).requires_grad_(True)
    iA, dA, bA, imA, gvA, gaA = pipeline(c, vA, aA, vi, None)
    # forward == C ABI, bit for bit
    d_c, i_c = capi.rasterize(vA.detach(), vi, c["H"], c["W"])
    assert th.equal(iA, i_c), "api rasterize != capi"
    rd_c, rb_c = capi.render(vA.detach(), vi, i_c)
    assert th.equal(dA, rd_c) and th.equal(bA, rb_c), "api render != capi"
    assert th.equal(imA, capi.interpolate(aA.detach(), vi, i_c, rb_c)), "api interpolate != capi"
    # route B: non-contiguous leaves, unfused edge-grad backward
    vB, aB = strided(c["v"]).requires_grad_(True), strided(c["attr"]).requires_grad_(True)
    viB = strided(c["vi"])
    seen = []
    iB, dB, bB, imB, gvB, gaB = pipeline(c, vB, aB, viB, lambda g: seen.append(tuple(g.shape)))
    assert th.equal(iA, iB) and th.equal(dA, dB) and th.equal(bA, bB) and th.equal(imA, imB), "forward differs for strided inputs"
    if c["v"].numel() and c["H"] * c["W"]:
        assert seen == [(c["N"], 3, c["H"], c["W"])], f"v_pix_img_hook saw {seen}"
    if os.environ.get("FUZZ_API_MUTATE") and gvB is not None:  # harness self-test: a 1e-3 relative error must be caught
        gvB = gvB * (1 + 1e-3)
    STATS.append((float(gvA.abs().max()) if gvA is not None and gvA.numel() else 0.0,
                  float((gvA - gvB).abs().max()) if gvA is not None and gvA.numel() else 0.0))
    for name, a, b in (("grad v_pix", gvA, gvB), ("grad attr", gaA, gaB)):
        assert (a is None) == (b is None), name
        if a is not None:
            assert a.shape == b.shape and bool(th.isfinite(a).all()), name
            FA._close(a, b.cpu(), f"{name}: fused/contiguous vs unfused/strided", **tol)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--first", type=int, default=0)
    a = ap.parse_args()
    bad = 0
    for seed in range(a.first, a.first + a.cases):
        c = FA.make_case(seed)
        try:
            run_case(c)
        except Exception as e:
            bad += 1
            print(f"FAIL seed {seed}: {FA.describe(c)}: {type(e).__name__}: {str(e)[:220]}", flush=True)
    nz = [s for s in STATS if s[0] > 0]
    if nz:
        rel = sorted(s[1] / s[0] for s in nz)
        print(f"cases with a non-zero vertex gradient: {len(nz)}/{len(STATS)}; max|A-B| / max|grad|: median {rel[len(rel)//2]:.2e}, max {rel[-1]:.2e}")
    print(f"{a.cases - bad}/{a.cases} cases passed")
    sys.exit(1 if bad else 0)

"""Diagnostic (scratch): on the fuzz seeds that failed the edge-grad check, compare GPU f32 / oracle f32 / oracle f64
on identical inputs, element by element, where GPU and oracle-f32 disagree beyond the fuzzer's tolerance."""
import os, sys
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch as th
import fuzz_all_ops as FZ
import oracle as O
from drtk_amd import capi
DEV = "cuda:0"
for seed in (10000, 10151, 10586, 11009, 12587):
    c = FZ.make_case(seed)
    v, vi, H, W, M = c["v"], c["vi"], c["H"], c["W"], 1e4
    d_o, i_o = O.rasterize(v, vi, H, W, nthreads=0)
    rd_o, rb_o = O.render(v, vi, i_o, nthreads=0)
    img = O.interpolate(c["attr"], vi, i_o, rb_o, nthreads=0) * (i_o != -1)[:, None]
    go = c["go"]
    eg32 = O.edge_grad_backward(v, img, i_o, vi, go, M)
    eg64 = O.edge_grad_backward(v.double(), img.double(), i_o, vi, go.double(), M)
    egg = capi.edge_grad_backward(v.to(DEV), img.to(DEV), i_o.to(DEV), vi.to(DEV), go.to(DEV), M).cpu()
    tol = 1e-5 + 1e-5 * float(eg32.abs().max())
    bad = ((egg - eg32).abs() > tol).nonzero()
    print(f"seed {seed}: shape {tuple(eg32.shape)} max|ref| {float(eg32.abs().max()):.4g} tol {tol:.3g}  elements beyond tol: {len(bad)}"
          f" | max|oracle32-truth64| over ALL elements {float((eg32.double()-eg64).abs().max()):.4g}"
          f" | max|gpu-truth64| {float((egg.double()-eg64).abs().max()):.4g}")
    for idx in bad[:6].tolist():
        t = tuple(idx)
        n, y, x = t[0], t[-3] if eg32.dim() == 4 and eg32.shape[-1] <= 3 else None, None
        print(f"   at {t}: gpu32 {float(egg[t]): .7g}   oracle32 {float(eg32[t]): .7g}   truth64 {float(eg64[t]): .7g}")
    # neighbourhood of the first bad element, whatever the layout: report index_img around it
    if len(bad):
        t = bad[0].tolist()
        shape = tuple(eg32.shape)
        # find (n,y,x) from the layout: dims equal to H and W adjacent
        n = t[0]
        hy = [k for k in range(1, len(shape) - 1) if shape[k] == H and shape[k + 1] == W]
        if hy:
            y, x = t[hy[0]], t[hy[0] + 1]
            ys, xs = slice(max(0, y - 1), y + 3), slice(max(0, x - 1), x + 3)
            print(f"   index_img[{n}, {ys.start}:{ys.stop}, {xs.start}:{xs.stop}] =\n{i_o[n, ys, xs]}")
    # fused route
    vg_o32, _ = O.interpolate_backward(eg32, v, vi, i_o, rb_o, True, False)
    vg_o64, _ = O.interpolate_backward(eg64, v.double(), vi, i_o, rb_o.double(), True, False)
    vgg = capi.edge_grad_backward_fused(v.to(DEV), img.to(DEV), i_o.to(DEV), vi.to(DEV), rb_o.to(DEV), go.to(DEV), M).cpu()
    tolf = 1e-5 + 1e-5 * float(vg_o32.abs().max())
    print(f"   fused: max|gpu-oracle32| {float((vgg-vg_o32).abs().max()):.4g} (tol {tolf:.3g})  max|oracle32-truth64| {float((vg_o32.double()-vg_o64).abs().max()):.4g}"
          f"  max|gpu-truth64| {float((vgg.double()-vg_o64).abs().max()):.4g}")

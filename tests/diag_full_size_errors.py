"""Diagnostic (GPU box): how far are the accumulated gradients of ONE full-size view from the truth?

For render backward, the attribute gradient and the fused edge route it prints, per configuration, the largest
|difference| of (a) HIP f32 vs the f32 oracle, (b) HIP f32 vs the f64 oracle, (c) f32 oracle vs f64 oracle, next to
max|ref| and the 1e-5 + 1e-5 * max|ref| bar.  (a) compares two f32 accumulations in different orders; (b) and (c) say
which of the two is nearer the exact sum.      python tests/diag_full_size_errors.py [100k 250k 1M]"""
import os
import sys

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import oracle as O  # noqa: E402
from drtk_amd import capi  # noqa: E402
from drtk_amd import synthetic as S  # noqa: E402

DEV = "cuda:0"
CFG = {"100k": (2048, 16), "250k": (2048, 16), "1M": (4096, 4)}


def main():
    for mesh in (sys.argv[1:] or list(CFG)):
        res, C = CFG[mesh]
        nl, no = S.MESH_SIZES[mesh]
        v, vi = S.sphere_views(1, nl, no, res, res, lobes=0.05)
        g = th.Generator().manual_seed(11)
        attr = th.rand(1, v.shape[1], C, generator=g)
        go = th.rand(1, C, res, res, generator=g) * 2 - 1
        gd = th.rand(1, res, res, generator=g) * 2 - 1
        gbar = th.rand(1, 3, res, res, generator=g) * 2 - 1
        _, idx = O.rasterize(v, vi, res, res, nthreads=0)
        _, bary = O.render(v, vi, idx, nthreads=0)
        img = O.interpolate(attr, vi, idx, bary, nthreads=0) * (idx != -1)[:, None]
        d = lambda t: t.to(DEV)  # noqa: E731
        v64, attr64, go64, gd64, gbar64 = (t.double() for t in (v, attr, go, gd, gbar))
        _, bary64 = O.render(v64, vi, idx, nthreads=0)
        img64 = O.interpolate(attr64, vi, idx, bary64, nthreads=0) * (idx != -1)[:, None]
        rows = []
        rows.append(("render_backward", capi.render_backward(d(v), d(vi), d(idx), d(gd), d(gbar)).cpu(),
                     O.render_backward(v, vi, idx, gd, gbar, nthreads=0), O.render_backward(v64, vi, idx, gd64, gbar64, nthreads=0)))
        ag_g, _ = capi.interpolate_backward(d(go), d(attr), d(vi), d(idx), d(bary))
        rows.append(("attr_grad", ag_g.cpu(), O.interpolate_backward(go, attr, vi, idx, bary, nthreads=0)[0],
                     O.interpolate_backward(go64, attr64, vi, idx, bary.double(), nthreads=0)[0]))
        eg32 = O.edge_grad_backward(v, img, idx, vi, go, nthreads=0)
        eg64 = O.edge_grad_backward(v64, img.double(), idx, vi, go64, nthreads=0)
        rows.append(("fused_edge", capi.edge_grad_backward_fused(d(v), d(img), d(idx), d(vi), d(bary), d(go)).cpu(),
                     O.interpolate_backward(eg32, v, vi, idx, bary, True, False, nthreads=0)[0],
                     O.interpolate_backward(eg64, v64, vi, idx, bary.double(), True, False, nthreads=0)[0]))
        for name, hip, o32, o64 in rows:
            m = float(o64.abs().max())
            e_a = float((hip.double() - o32.double()).abs().max())
            e_b = float((hip.double() - o64).abs().max())
            e_c = float((o32.double() - o64).abs().max())
            print(f"{mesh:5s} {name:16s} max|ref| {m:10.3f}  bar {1e-5 + 1e-5 * m:.3e}  hip-o32 {e_a:.3e}  hip-o64 {e_b:.3e}  o32-o64 {e_c:.3e}", flush=True)


if __name__ == "__main__":
    main()

"""Diagnostic (GPU): how many units of u * A_i (A_i = magnitude accumulated into element i, oracle.accumulated_magnitudes)
the HIP kernels and the float32 oracle are away from the double oracle, per backward operator, over the scenes of
tests/test_gpu_f64_distance.py -- the measurement behind the per-operator `ulps` of the element-wise bound
(tests/f64_distance.py).   python tests/diag_elementwise_ratios.py"""
import os
import sys

import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import oracle as O  # noqa: E402
from drtk_amd import capi  # noqa: E402
from drtk_amd import synthetic as S  # noqa: E402

U = 2.0 ** -24
DEV = "cuda:0"
worst = {}
details = []
CURRENT = ""


def note(name, got, o32, o64, A):
    got, o32, o64, A = (t.detach().cpu().double() for t in (got, o32, o64, A))
    m = A > 0
    k = float(((got - o64).abs()[m] / (U * A[m])).max()) if bool(m.any()) else 0.0
    o = float(((o32 - o64).abs()[m] / (U * A[m])).max()) if bool(m.any()) else 0.0
    # ... and with the k * |oracle_f32 - f64| term of the bound: what remains for the A term
    own = (o32 - o64).abs()
    rest = (got - o64).abs() - th.clamp(3 * own, min=1e-7 * float(o64.abs().max()))  # beyond 3 own and the absolute floor 1e-7 max|f64|
    r = float((rest[m] / (U * A[m])).max()) if bool(m.any()) else 0.0
    dead = float((got - o64).abs()[~m].max()) if bool((~m).any()) else 0.0
    w = worst.setdefault(name, [0.0, 0.0, 0.0, 0.0])
    w[0], w[1], w[2], w[3] = max(w[0], k), max(w[1], o), max(w[2], r), max(w[3], dead)
    if bool(m.any()) and r > 64.0 and len(details) < 12:
        q = th.where(m, rest / (U * A).clamp(min=1e-300), th.zeros_like(rest)).flatten()
        i = int(q.argmax())
        details.append(f"{name} [{CURRENT}]: element {i}: kernel {float(got.flatten()[i]):.9e} oracle32 {float(o32.flatten()[i]):.9e} f64 {float(o64.flatten()[i]):.9e} "
                       f"A {float(A.flatten()[i]):.3e}  -> (|k - f64| - 3 |o32 - f64|) / uA = {float(q[i]):.3e}")
    if dead > 0 and len(details) < 24:
        e = th.where(~m, (got - o64).abs(), th.zeros_like(got)).flatten()
        i = int(e.argmax())
        details.append(f"{name} [{CURRENT}]: A = 0 at element {i}: kernel {float(got.flatten()[i]):.9e} oracle32 {float(o32.flatten()[i]):.9e} f64 {float(o64.flatten()[i]):.9e}")


def d(x):
    return x.double() if x.is_floating_point() else x


def path_ops(v, vi, attr, go, gd, gb, H, W):
    _, idx = O.rasterize(v, vi, H, W, nthreads=0)
    _, bary = O.render(v, vi, idx, nthreads=0)
    g = lambda t: t.to(DEV)  # noqa: E731
    r32, r64 = O.render_backward(v, vi, idx, gd, gb), O.render_backward(d(v), vi, idx, d(gd), d(gb))
    with O.accumulated_magnitudes():
        rA = O.render_backward(d(v), vi, idx, d(gd), d(gb))
    note("render_backward", capi.render_backward(g(v), g(vi), g(idx), g(gd), g(gb)), r32, r64, rA)
    ag, bg = capi.interpolate_backward(g(go), g(attr), g(vi), g(idx), g(bary))
    a32, b32 = O.interpolate_backward(go, attr, vi, idx, bary)
    a64, b64 = O.interpolate_backward(d(go), d(attr), vi, idx, d(bary))
    with O.accumulated_magnitudes():
        aA, bA = O.interpolate_backward(d(go), d(attr), vi, idx, d(bary))
    note("interpolate_backward attr", ag, a32, a64, aA)
    note("interpolate_backward bary", bg, b32, b64, bA)
    img = O.interpolate(attr, vi, idx, bary, nthreads=0) * (idx != -1)[:, None]
    for M in (1e4, 0.5):
        e32, e64 = O.edge_grad_backward(v, img, idx, vi, go, M), O.edge_grad_backward(d(v), d(img), idx, vi, d(go), M)
        with O.accumulated_magnitudes():
            eA = O.edge_grad_backward(d(v), d(img), idx, vi, d(go), M)
        note(f"edge_grad_backward M={M}", capi.edge_grad_backward(g(v), g(img), g(idx), g(vi), g(go), M), e32, e64, eA)
        v32, _ = O.interpolate_backward(e32, v, vi, idx, bary, True, False)
        v64, _ = O.interpolate_backward(e64, d(v), vi, idx, d(bary), True, False)
        vA, _ = O.interpolate_backward(eA, d(v), vi, idx, d(bary).abs(), True, False)  # the plain operator on magnitudes
        note(f"fused edge route M={M}", capi.edge_grad_backward_fused(g(v), g(img), g(idx), g(vi), g(bary), g(go), M), v32, v64, vA)


for shape in [(3, 40, 44, 256, 320, 7), (1, 70, 72, 512, 512, 16), (2, 12, 14, 129, 203, 3), (2, 30, 34, 200, 256, 24)]:
    n, nl, no, H, W, C = shape
    CURRENT = f"spheres {shape}"
    v, vi = S.sphere_views(n, nl, no, H, W, second_sphere=True)
    gen = th.Generator().manual_seed(11)
    attr = th.rand(n, v.shape[1], C, generator=gen)
    gd = th.rand(n, H, W, generator=gen) * 2 - 1
    gb = th.rand(n, 3, H, W, generator=gen) * 2 - 1
    go = th.rand(n, C, H, W, generator=gen) * 2 - 1
    path_ops(v, vi, attr, go, gd, gb, H, W)
import fuzz_all_ops as F  # noqa: E402

for seed in range(500, 580):
    for wide in (False, True):
        c = F.make_case(seed, wide_channels=wide)
        if c["dtype"] == th.float32:
            CURRENT = f"fuzz_all_ops seed {seed} wide={wide}"
            path_ops(c["v"], c["vi"], c["attr"], c["go"], c["gd"], c["gb"], c["H"], c["W"])
print(f"{'operator':36s} {'kernel/uA':>12s} {'oracle32/uA':>12s} {'beyond/uA':>16s} {'err where A=0':>14s}")
for k, w in worst.items():
    print(f"{k:36s} {w[0]:12.1f} {w[1]:12.1f} {w[2]:16.1f} {w[3]:14.3e}")
print("\n".join(details))

"""Randomised sweep of mipmap_grid_sampler_2d forward/backward through the C ABI against the CPU oracle
(not collected by pytest -- run on a GPU box: `python tests/fuzz_mipmap.py --cases 200`; a fixed subset
runs as tests/test_gpu_mipmap.py::test_randomised_mipmap_cases).  Non-square / odd textures and outputs,
1..11 levels, 1..6 channels (C <= 4 bilinear f32 takes the tiled backward kernel, the rest the direct one),
smooth and incoherent uv fields, footprints from magnification to beyond the coarsest level, every
padding / interpolation mode, align_corners, force_max_aniso, clip_grad."""
import argparse
import os

os.environ.setdefault("DRTK_CAPI_POISON", "1")  # outputs of the ctypes binding pre-filled with NaN / sentinels (drtk_amd/capi.py _out)
import sys

import torch as th

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))

DEV = "cuda:0"


def _close(a, ref, what, atol=1e-5, rtol=1e-5):
    a = a.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert a.shape == ref.shape, (what, a.shape, ref.shape)
    tol = atol + rtol * float(ref.abs().max()) if ref.numel() else atol
    err = float((a - ref).abs().max()) if ref.numel() else 0.0
    assert err <= tol, f"{what}: max abs err {err:.3e} > tol {tol:.3e}"


def make_case(seed):
    g = th.Generator().manual_seed(1000 + seed)
    r = lambda lo, hi: int(th.randint(lo, hi + 1, (1,), generator=g))  # noqa: E731
    N, C = r(1, 3), r(1, 6)
    dtype = th.float64 if r(0, 4) == 0 else th.float32
    h0 = [1, 2, 5, 16, 17, 31, 48, 64, 100, 128][r(0, 9)]
    w0 = [1, 3, 8, 16, 23, 32, 64, 65, 96, 256][r(0, 9)]
    tex = [th.rand(N, C, h0, w0, generator=g, dtype=th.float64).to(dtype)]
    want_levels = r(1, 11)
    while len(tex) < want_levels and min(tex[-1].shape[-2:]) > 1:
        tex.append(th.nn.functional.avg_pool2d(tex[-1], 2))
    H = [1, 3, 16, 17, 40, 64, 75][r(0, 6)]
    W = [1, 2, 16, 31, 48, 64, 90][r(0, 6)]
    if r(0, 1) == 0:  # smooth uv field (an affine map plus a ripple): neighbouring pixels share texels
        ys, xs = th.meshgrid(th.arange(H, dtype=th.float64), th.arange(W, dtype=th.float64), indexing="ij")
        a = (th.rand(N, 2, 3, generator=g, dtype=th.float64) - 0.5) * th.tensor([0.08, 0.08, 2.0], dtype=th.float64)
        grid = th.stack([a[:, 0, 0, None, None] * xs + a[:, 0, 1, None, None] * ys + a[:, 0, 2, None, None],
                         a[:, 1, 0, None, None] * xs + a[:, 1, 1, None, None] * ys + a[:, 1, 2, None, None]], -1)
        grid = grid + 0.02 * th.sin(xs * 0.7 + ys * 0.3)[None, ..., None]
        jac = th.stack([th.stack([a[:, 0, 0], a[:, 1, 0]], -1), th.stack([a[:, 0, 1], a[:, 1, 1]], -1)], -2)  # [N,2,2]
        jac = (jac[:, None, None] * 0.5).expand(N, H, W, 2, 2).clone()
        jac = jac * (1 + 0.2 * th.rand(N, H, W, 1, 1, generator=g, dtype=th.float64))
    else:
        grid = th.rand(N, H, W, 2, generator=g, dtype=th.float64) * 2.6 - 1.3
        jac = th.randn(N, H, W, 2, 2, generator=g, dtype=th.float64) * 0.05
        jac[..., 0, :] *= th.rand(N, H, W, 1, generator=g, dtype=th.float64) * 4 + 0.05
    jac = jac * [0.02, 0.3, 1.0, 3.0, 20.0][r(0, 4)]
    gout = th.rand(N, C, H, W, generator=g, dtype=th.float64) * 2 - 1
    if r(0, 2) == 0:
        gout = gout * (th.rand(N, 1, H, W, generator=g, dtype=th.float64) > 0.4)  # masked upstream gradient
    return dict(tex=tex, grid=grid.to(dtype), jac=jac.to(dtype), gout=gout.to(dtype), max_aniso=[1, 2, 4, 8, 16][r(0, 4)],
                padding=r(0, 2), mode=[0, 0, 2][r(0, 2)], align=bool(r(0, 1)), force=r(0, 3) == 0, clip=bool(r(0, 1)),
                dtype=dtype)


def run_case(c, place=None):
    """One case; with DRTK_CAPI_GUARD=g in the environment also: nothing was written outside an output or a workspace."""
    from drtk_amd import capi

    out = _run_case(c, place)
    capi.check_guards()
    return out


def _run_case(c, place=None):
    """`place(tensor)` puts an input on the device (default: a plain copy; fuzz_all_ops.misaligned for odd pointers)."""
    import oracle as O
    from drtk_amd import capi

    d = place or (lambda x: x.to(DEV))
    args = (c["max_aniso"], c["padding"], c["mode"], c["align"], c["force"], c["clip"])
    f64 = c["dtype"] == th.float64
    tol = dict(atol=1e-11, rtol=1e-10) if f64 else dict(atol=2e-5, rtol=2e-5)
    want = O.mipmap_grid_sampler_2d(c["tex"], c["grid"], c["jac"], *args)
    got = capi.mipmap_grid_sampler_2d([d(t) for t in c["tex"]], d(c["grid"]), d(c["jac"]), *args)
    _close(got, want, "forward", **tol)
    wl, wg = O.mipmap_grid_sampler_2d_backward(c["gout"], c["tex"], c["grid"], c["jac"], *args)
    gl, gg = capi.mipmap_grid_sampler_2d_backward(d(c["gout"]), [d(t) for t in c["tex"]], d(c["grid"]), d(c["jac"]), *args)
    gtol = dict(atol=1e-10, rtol=1e-9) if f64 else dict(atol=1e-4, rtol=1e-4)
    _close(gg, wg, "grad grid", **gtol)
    for i, (a, b) in enumerate(zip(gl, wl)):
        _close(a, b, f"grad level {i}", **gtol)


def describe(c):
    t = c["tex"]
    return (f"N={t[0].shape[0]} C={t[0].shape[1]} tex={tuple(t[0].shape[-2:])} levels={len(t)} out={tuple(c['grid'].shape[1:3])} "
            f"{str(c['dtype']).split('.')[-1]} aniso={c['max_aniso']} pad={c['padding']} mode={c['mode']} align={c['align']} "
            f"force={c['force']} clip={c['clip']}")


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
        except AssertionError as e:
            bad += 1
            print(f"FAIL seed {seed}: {describe(c)}: {e}", flush=True)
    print(f"{a.cases - bad}/{a.cases} cases passed")
    sys.exit(1 if bad else 0)

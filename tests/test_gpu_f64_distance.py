"""-m gpu: every backward operator and every sampler mode held to ONE bound (tests/f64_distance.py):

    |HIP_f32 - oracle_f64|  <=  max(1e-5 * max|ref|, 3 * |oracle_f32 - oracle_f64|)      (+ the accumulation term where
                                                                                          an output has a handful of elements)

on the committed fixtures' scenes, on the shapes the fuzz generators draw, and on the three seeds whose hand-tuned
bounds were loosened in round 3 (fuzz_mipmap_snapped 340826, fuzz_next_ops 450324 and 202311).  The oracle in double is
the same restatement run on inputs cast to double, with the DISCRETE inputs (index_img, topology) kept from the float32
pipeline, so both precisions differentiate the same image."""
import os
import sys

import pytest
import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(x):
    return x.to(DEV) if isinstance(x, th.Tensor) else x


def dbl(x):
    return x.double() if isinstance(x, th.Tensor) and x.is_floating_point() else x


def _path_ops(v, vi, attr, go, gd, gb, H, W, what):
    """render / interpolate / edge_grad backward (unfused and fused) of one float32 scene."""
    import oracle as O
    from drtk_amd import capi
    from f64_distance import assert_within_f64_distance as within

    _, idx = O.rasterize(v, vi, H, W, nthreads=0)
    _, bary = O.render(v, vi, idx, nthreads=0)
    v6, a6, go6, gd6, gb6, bary6 = (dbl(t) for t in (v, attr, go, gd, gb, bary))
    within(capi.render_backward(dev(v), dev(vi), dev(idx), dev(gd), dev(gb)), O.render_backward(v, vi, idx, gd, gb),
           O.render_backward(v6, vi, idx, gd6, gb6), f"{what}: render backward")
    ag, bg = capi.interpolate_backward(dev(go), dev(attr), dev(vi), dev(idx), dev(bary))
    ag32, bg32 = O.interpolate_backward(go, attr, vi, idx, bary)
    ag64, bg64 = O.interpolate_backward(go6, a6, vi, idx, bary6)
    within(ag, ag32, ag64, f"{what}: interpolate backward, attributes")
    within(bg, bg32, bg64, f"{what}: interpolate backward, barycentrics")
    img = O.interpolate(attr, vi, idx, bary, nthreads=0) * (idx != -1)[:, None]
    for M in (1e4, 0.5):
        eg32 = O.edge_grad_backward(v, img, idx, vi, go, M)
        eg64 = O.edge_grad_backward(v6, dbl(img), idx, vi, go6, M)
        within(capi.edge_grad_backward(dev(v), dev(img), dev(idx), dev(vi), dev(go), M), eg32, eg64, f"{what}: edge_grad backward, max_dp_dr={M}")
        vg32, _ = O.interpolate_backward(eg32, v, vi, idx, bary, True, False)
        vg64, _ = O.interpolate_backward(eg64, v6, vi, idx, bary6, True, False)
        within(capi.edge_grad_backward_fused(dev(v), dev(img), dev(idx), dev(vi), dev(bary), dev(go), M), vg32, vg64,
               f"{what}: fused edge_grad route, max_dp_dr={M}")


@pytest.mark.parametrize("shape", [(3, 40, 44, 256, 320, 7), (1, 70, 72, 512, 512, 16), (2, 12, 14, 129, 203, 3), (2, 30, 34, 200, 256, 24)])
def test_path_backward_ops_on_seeded_scenes(shape):
    from drtk_amd import synthetic as S

    n, nl, no, H, W, C = shape
    v, vi = S.sphere_views(n, nl, no, H, W, second_sphere=True)
    g = th.Generator().manual_seed(11)
    attr = th.rand(n, v.shape[1], C, generator=g)
    gd = th.rand(n, H, W, generator=g) * 2 - 1
    gb = th.rand(n, 3, H, W, generator=g) * 2 - 1
    go = th.rand(n, C, H, W, generator=g) * 2 - 1
    _path_ops(v, vi, attr, go, gd, gb, H, W, f"spheres {shape}")


@pytest.mark.parametrize("block", range(2))
def test_path_backward_ops_on_fuzz_shapes(block):
    import fuzz_all_ops as F

    done = 0
    for seed in range(500 + 40 * block, 500 + 40 * block + 40):
        for wide in (False, True):
            c = F.make_case(seed, wide_channels=wide)
            if c["dtype"] != th.float32:
                continue
            _path_ops(c["v"], c["vi"], c["attr"], c["go"], c["gd"], c["gb"], c["H"], c["W"], f"fuzz_all_ops seed {seed} wide={wide}: {F.describe(c)}")
            done += 1
    assert done >= 20


def _sampler(levels, grid, jac, gout, args, what, small_levels_acc=True):
    import oracle as O
    from drtk_amd import capi
    from f64_distance import assert_within_f64_distance as within

    gl, gg = capi.mipmap_grid_sampler_2d_backward(dev(gout), [dev(t) for t in levels], dev(grid), dev(jac), *args)
    l32, g32 = O.mipmap_grid_sampler_2d_backward(gout, levels, grid, jac, *args)
    l64, g64 = O.mipmap_grid_sampler_2d_backward(dbl(gout), [dbl(t) for t in levels], dbl(grid), dbl(jac), *args)
    # the tap count and the level are discontinuous in the footprint: a pixel where float32 and double take different ones
    # moves both float32 results away from the double one by the same O(1) amount -- the bound follows
    within(gg, g32, g64, f"{what}: grid gradient")
    al, _ = O.mipmap_grid_sampler_2d_backward(gout.abs(), levels, grid, jac, *args)
    for k in range(len(levels)):
        within(gl[k], l32[k], l64[k], f"{what}: gradient of level {k}", acc_magnitude=float(al[k].abs().max()) if small_levels_acc else None)
    fw = capi.mipmap_grid_sampler_2d([dev(t) for t in levels], dev(grid), dev(jac), *args)
    within(fw, O.mipmap_grid_sampler_2d(levels, grid, jac, *args), O.mipmap_grid_sampler_2d([dbl(t) for t in levels], dbl(grid), dbl(jac), *args),
           f"{what}: forward")


@pytest.mark.parametrize("mode", [0, 2], ids=["bilinear", "bicubic"])
@pytest.mark.parametrize("padding", [0, 1, 2], ids=["zeros", "border", "reflection"])
def test_sampler_backward_modes_on_fuzz_shapes(mode, padding):
    import fuzz_mipmap as FM

    done = 0
    for seed in range(700, 760):
        c = FM.make_case(seed)
        if c["dtype"] != th.float32:
            continue
        args = (c["max_aniso"], padding, mode, c["align"], c["force"], c["clip"])
        _sampler(c["tex"], c["grid"], c["jac"], c["gout"], args, f"fuzz_mipmap seed {seed} mode={mode} padding={padding}: {FM.describe(c)}")
        done += 1
    assert done >= 30


def test_the_three_seeds_whose_bounds_were_loosened_in_round_3():
    """fuzz_mipmap_snapped 340826 (bicubic, a 1 x 1 level collecting 874 pixels x 4 taps x 16 weights), fuzz_next_ops 450324
    (screen_space_uv_derivative's median on one row of 64 pixels) and 202311 (an A^T A entry summing ~24 000 products): the
    fuzzers themselves, which now state their bounds through tests/f64_distance.py."""
    import fuzz_mipmap_snapped as FS
    import fuzz_next_ops as FN

    FS.run_case(FS.make_case(340826))
    for seed in (450324, 202311):
        FN.run_case(FN.make_case(seed))


def test_sparse_operators_and_transform_on_fuzz_shapes():
    """interpolation_normal_matrix_values (+ backward), interpolation_matrix backward and the pinhole transform's VJP."""
    import fuzz_next_ops as FN

    for seed in range(800, 840):
        c = FN.make_case(seed)
        try:
            FN.run_case(c)
        except AssertionError as e:
            raise AssertionError(f"seed {seed}: {FN.describe(c)}: {e}") from e

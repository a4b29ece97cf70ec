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


# ---- per element (round 6): tests/f64_distance.py assert_elementwise_within ----------------------------------------------
# ulps of the magnitude accumulated into an element (oracle.accumulated_magnitudes) that the kernel may be away from the double
# result, per operator -- measured by tests/diag_elementwise_ratios.py over these scenes (the largest excess beyond 3 x the float32
# oracle's own distance and the absolute floor, in units of u A_i): interpolate backward 0.7, the edge routes 0.6, render
# backward 0.6 -- 108 before its magnitudes counted the cancellation INSIDE the terms (the gradient of p0 is minus the sum of
# the others', b12 = -b0 + b1, ...: oracle/drtk_oracle_body.inc, the block under drtk_oracle_abs_accumulate).
ELEMENT_ULPS = {"render": 8.0, "interpolate": 4.0, "edge": 4.0}
FLOOR = 1e-7  # x max|f64|: what is not resolved per element (the north star's 1e-5 of the output's scale, tightened 100 x)


def _elementwise_path_ops(v, vi, attr, go, gd, gb, H, W, what, mutate=False):
    """Every element of every gradient of the path within max(3 |oracle_f32 - f64|, ulps u A_i, 1e-7 max|f64|) of the double
    oracle.  mutate: returns the kernel outputs, the oracle triples and magnitudes instead, for the sensitivity test."""
    import oracle as O
    from drtk_amd import capi
    from f64_distance import assert_elementwise_within as within

    _, idx = O.rasterize(v, vi, H, W, nthreads=0)
    _, bary = O.render(v, vi, idx, nthreads=0)
    v6, a6, go6, gd6, gb6, bary6 = (dbl(t) for t in (v, attr, go, gd, gb, bary))
    cases = []
    r32, r64 = O.render_backward(v, vi, idx, gd, gb), O.render_backward(v6, vi, idx, gd6, gb6)
    with O.accumulated_magnitudes():
        rA = O.render_backward(v6, vi, idx, gd6, gb6)
    cases.append(("render backward", capi.render_backward(dev(v), dev(vi), dev(idx), dev(gd), dev(gb)), r32, r64, rA, ELEMENT_ULPS["render"]))
    ag, bg = capi.interpolate_backward(dev(go), dev(attr), dev(vi), dev(idx), dev(bary))
    a32, b32 = O.interpolate_backward(go, attr, vi, idx, bary)
    a64, b64 = O.interpolate_backward(go6, a6, vi, idx, bary6)
    with O.accumulated_magnitudes():
        aA, bA = O.interpolate_backward(go6, a6, vi, idx, bary6)
    cases.append(("interpolate backward, attributes", ag, a32, a64, aA, ELEMENT_ULPS["interpolate"]))
    cases.append(("interpolate backward, barycentrics", bg, b32, b64, bA, ELEMENT_ULPS["interpolate"]))
    img = O.interpolate(attr, vi, idx, bary, nthreads=0) * (idx != -1)[:, None]
    e32, e64 = O.edge_grad_backward(v, img, idx, vi, go, 1e4), O.edge_grad_backward(v6, dbl(img), idx, vi, go6, 1e4)
    with O.accumulated_magnitudes():
        eA = O.edge_grad_backward(v6, dbl(img), idx, vi, go6, 1e4)
    cases.append(("edge_grad backward", capi.edge_grad_backward(dev(v), dev(img), dev(idx), dev(vi), dev(go), 1e4), e32, e64, eA, ELEMENT_ULPS["edge"]))
    v32, _ = O.interpolate_backward(e32, v, vi, idx, bary, True, False)
    v64, _ = O.interpolate_backward(e64, v6, vi, idx, bary6, True, False)
    vA, _ = O.interpolate_backward(eA, v6, vi, idx, bary6.abs(), True, False)  # (the plain operator on magnitudes)
    cases.append(("fused edge_grad route", capi.edge_grad_backward_fused(dev(v), dev(img), dev(idx), dev(vi), dev(bary), dev(go), 1e4), v32, v64, vA, ELEMENT_ULPS["edge"]))
    if mutate:
        return cases
    for name, got, o32, o64, A, ulps in cases:
        within(got, o32, o64, A, ulps, f"{what}: {name}", floor_rel=FLOOR)


def _sphere_case(shape):
    from drtk_amd import synthetic as S

    n, nl, no, H, W, C = shape
    v, vi = S.sphere_views(n, nl, no, H, W, second_sphere=True)
    g = th.Generator().manual_seed(11)
    attr = th.rand(n, v.shape[1], C, generator=g)
    gd = th.rand(n, H, W, generator=g) * 2 - 1
    gb = th.rand(n, 3, H, W, generator=g) * 2 - 1
    go = th.rand(n, C, H, W, generator=g) * 2 - 1
    return v, vi, attr, go, gd, gb, H, W


@pytest.mark.parametrize("shape", [(3, 40, 44, 256, 320, 7), (1, 70, 72, 512, 512, 16), (2, 12, 14, 129, 203, 3), (2, 30, 34, 200, 256, 24)])
def test_every_gradient_element_within_its_own_accumulated_magnitude(shape):
    _elementwise_path_ops(*_sphere_case(shape), f"spheres {shape}")


@pytest.mark.parametrize("block", range(2))
def test_every_gradient_element_on_fuzz_shapes(block):
    import fuzz_all_ops as F

    done = 0
    for seed in range(500 + 40 * block, 500 + 40 * block + 40):
        for wide in (False, True):
            c = F.make_case(seed, wide_channels=wide)
            if c["dtype"] != th.float32:
                continue
            _elementwise_path_ops(c["v"], c["vi"], c["attr"], c["go"], c["gd"], c["gb"], c["H"], c["W"], f"fuzz_all_ops seed {seed} wide={wide}: {F.describe(c)}")
            done += 1
    assert done >= 20


def test_the_elementwise_bound_sees_a_small_vertex_that_lost_a_contribution():
    """What the max-norm bars cannot see and this one must: ONE low-magnitude element that is wrong by the size of what was
    accumulated into it.  The kernels' own outputs are mutated -- the contributions of one run of pixels dropped from the
    element with the SMALLEST non-negligible accumulated magnitude (its value replaced by what the double oracle has without
    the largest half of that magnitude, i.e. off by A_i / 2) -- and every operator's mutated output must (a) still pass the
    max-norm bound of tests/f64_distance.py (that is the blind spot) and (b) fail the element-wise one."""
    from f64_distance import elementwise_excess, f64_distance_bound

    cases = _elementwise_path_ops(*_sphere_case((2, 30, 34, 200, 256, 24)), "mutation", mutate=True)
    for name, got, o32, o64, A, ulps in cases:
        got = got.detach().cpu().clone()
        Af = A.detach().cpu().double().flatten()
        scale = float(o64.abs().max())
        # the smallest accumulated magnitude that is still well above the floor (20 x) and well below the max-norm bar's reach
        cand = th.where((Af > 20 * FLOOR * scale) & (Af < 0.2 * 1e-5 * scale / 0.5), Af, th.full_like(Af, float("inf")))
        i = int(cand.argmin())
        if not bool(th.isfinite(cand[i])):
            continue  # (no such element in this operator's output on this scene: nothing to demonstrate)
        assert elementwise_excess(got, o32, o64, A, ulps, floor_rel=FLOOR)[0] <= 1.0, name
        g = got.flatten()
        g[i] = g[i] - 0.5 * float(Af[i]) * (1.0 if float(o64.flatten()[i]) >= 0 else -1.0)  # half of what was accumulated is gone
        bound, _ = f64_distance_bound(o32, o64)
        assert float((got.double() - o64.double()).abs().max()) <= bound, f"{name}: the mutation is visible to the max-norm bound already -- pick a smaller element"
        excess, worst, _ = elementwise_excess(got, o32, o64, A, ulps, floor_rel=FLOOR)
        assert excess > 1.0 and worst == i, f"{name}: a vertex that lost half of its accumulated magnitude ({float(Af[i]):.3e}) passes the element-wise bound"
        return
    pytest.skip("no operator offered a low-magnitude element on this scene")


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

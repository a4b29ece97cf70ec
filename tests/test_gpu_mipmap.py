"""GPU (-m gpu): the HIP anisotropic mipmap grid sampler through the C ABI and through
`drtk_amd.mipmap_grid_sample`, against the fixtures of the reference's pure-PyTorch model
(force_max_aniso=True, clip_grad=False) and against the CPU restatement in every mode.

Tolerance: |d| <= 1e-5 + 1e-5 * max|ref| per tensor (f32); f64 1e-10."""
import pytest
import torch as th
from conftest import MIPMAP_ADAPTIVE_CASES, MIPMAP_CASES, load_mipmap, mipmap_inputs

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def close(a, ref, what, atol=1e-5, rtol=1e-5):
    a, ref = a.detach().cpu().double(), ref.detach().cpu().double()
    assert a.shape == ref.shape, (what, a.shape, ref.shape)
    tol = atol + rtol * float(ref.abs().max()) if ref.numel() else atol
    err = float((a - ref).abs().max()) if ref.numel() else 0.0
    assert err <= tol, f"{what}: max abs err {err:.3e} > tol {tol:.3e}"


def dev(x):
    return [t.to(DEV) for t in x] if isinstance(x, list) else x.to(DEV)


@pytest.mark.parametrize("name", MIPMAP_CASES)
def test_capi_matches_reference_model_fixture(name):
    from drtk_amd import capi

    c = load_mipmap(name)
    tex, grid, vt = dev(c["tex"]), dev(c["grid"]), dev(c["vt"])
    out = capi.mipmap_grid_sampler_2d(tex, grid, vt, c["max_aniso"], c["padding"], c["mode"], False, True, False)
    close(out, c["out"], "forward")
    glv, gg = capi.mipmap_grid_sampler_2d_backward(
        dev(c["grad_out"]), tex, grid, vt, c["max_aniso"], c["padding"], c["mode"], False, True, False)
    close(gg, c["grad_grid"], "grad grid")
    for i, (g, ref) in enumerate(zip(glv, c["grad_tex"])):
        close(g, ref, f"grad level {i}")


@pytest.mark.parametrize("name", MIPMAP_ADAPTIVE_CASES)
def test_adaptive_tap_count_matches_reference_model_fixture(name):
    """force_max_aniso=False (the default of drtk.mipmap_grid_sample): kernel through the C ABI and through the Python
    API with autograd, against the reference model's outputs assembled per tap-count class
    (tests/test_mipmap_oracle.py::test_oracle_adaptive_tap_count_matches_reference_model_fixture explains the fixture)."""
    import drtk_amd
    from drtk_amd import capi

    c = load_mipmap(name)
    tex, grid, vt = dev(c["tex"]), dev(c["grid"]), dev(c["vt"])
    out = capi.mipmap_grid_sampler_2d(tex, grid, vt, c["max_aniso"], c["padding"], c["mode"], False, False, False)
    close(out, c["out"], "forward")
    glv, gg = capi.mipmap_grid_sampler_2d_backward(
        dev(c["grad_out"]), tex, grid, vt, c["max_aniso"], c["padding"], c["mode"], False, False, False)
    close(gg, c["grad_grid"], "grad grid")
    for i, (g, ref) in enumerate(zip(glv, c["grad_tex"])):
        close(g, ref, f"grad level {i}")
    texr = [t.to(DEV).requires_grad_(True) for t in c["tex"]]
    gridr = c["grid"].to(DEV).requires_grad_(True)
    mode = "bilinear" if c["mode"] == 0 else "bicubic"
    padding = ["zeros", "border", "reflection"][c["padding"]]
    o2 = drtk_amd.mipmap_grid_sample(texr, gridr, vt, c["max_aniso"], mode=mode, padding_mode=padding)
    close(o2, c["out"], "forward (python api, default force_max_aniso)")
    o2.backward(dev(c["grad_out"]))
    close(gridr.grad, c["grad_grid"], "grad grid (autograd)")
    for i, (t, ref) in enumerate(zip(texr, c["grad_tex"])):
        close(t.grad, ref, f"grad level {i} (autograd)")


@pytest.mark.parametrize("name", MIPMAP_CASES)
def test_python_api_and_autograd_match_reference_model_fixture(name):
    import drtk_amd

    c = load_mipmap(name)
    tex = [t.to(DEV).requires_grad_(True) for t in c["tex"]]
    grid = c["grid"].to(DEV).requires_grad_(True)
    mode = "bilinear" if c["mode"] == 0 else "bicubic"
    padding = ["zeros", "border", "reflection"][c["padding"]]
    out = drtk_amd.mipmap_grid_sample(tex, grid, dev(c["vt"]), c["max_aniso"], mode=mode, padding_mode=padding,
                                      force_max_aniso=True)
    close(out, c["out"], "forward")
    out.backward(dev(c["grad_out"]))
    close(grid.grad, c["grad_grid"], "grad grid")
    for i, (t, ref) in enumerate(zip(tex, c["grad_tex"])):
        close(t.grad, ref, f"grad level {i}")


@pytest.mark.parametrize("mode", [0, 2])
@pytest.mark.parametrize("padding", [0, 1, 2])
@pytest.mark.parametrize("flags", [(False, False, False), (False, False, True), (True, True, False), (True, False, True)])
def test_capi_matches_oracle_in_every_mode(mode, padding, flags):
    """Adaptive tap count, clip_grad and align_corners have no executable reference: HIP vs the CPU
    restatement on seeded inputs (anisotropy 0.05..4x, footprints beyond the coarsest level)."""
    import oracle as O
    from drtk_amd import capi

    align, force, clip = flags
    tex, grid, vt, gout = mipmap_inputs(11 + mode + 3 * padding, 2, 3, 32, 3, 24, 28)
    want = O.mipmap_grid_sampler_2d(tex, grid, vt * 3, 6, padding, mode, align, force, clip)
    got = capi.mipmap_grid_sampler_2d(dev(tex), dev(grid), dev(vt * 3), 6, padding, mode, align, force, clip)
    close(got, want, "forward")
    wl, wg = O.mipmap_grid_sampler_2d_backward(gout, tex, grid, vt * 3, 6, padding, mode, align, force, clip)
    gl, gg = capi.mipmap_grid_sampler_2d_backward(dev(gout), dev(tex), dev(grid), dev(vt * 3), 6, padding, mode, align, force, clip)
    close(gg, wg, "grad grid", atol=2e-5)
    for i, (a, b) in enumerate(zip(gl, wl)):
        close(a, b, f"grad level {i}")


@pytest.mark.parametrize("padding", [0, 1, 2])
@pytest.mark.parametrize("C", [3, 4])
def test_masked_upstream_gradients_magnified_pixels_and_tap_counts_beyond_the_tap_table(padding, C):
    """What the sampler kernels skip or look up must not change a result (oracle: nothing is skipped there).
    * upstream gradient zero in a whole 16 x 16 tile (the tiled backward leaves early), in part of a tile and in
      scattered pixels (dead pixels neither place the LDS windows nor run their taps), and in one channel everywhere;
    * a magnified band (footprint < 1 texel: a == 0, the second level has weight zero and is skipped by both passes)
      next to a minified one;
    * max_aniso = 12 with footprints of ratio up to ~20: tap counts above the 8 rows of the LDS table of tap offsets."""
    import oracle as O
    from drtk_amd import capi

    H, W = 40, 48
    tex, grid, vt, gout = mipmap_inputs(300 + 7 * padding + C, 2, C, 64, 5, H, W, jscale=0.03)
    vt = vt.clone()
    vt[:, :, : W // 3] *= 0.02           # magnified: every pixel on level 0 with a == 0
    vt[:, :, W // 3: 2 * W // 3, 0] *= 5  # strongly anisotropic: many taps
    gout = gout.clone()
    gout[:, :, :16, :16] = 0              # a dead tile
    gout[:, :, 16:32, 20:40] = 0          # parts of four tiles
    gout[:, 1] = 0                        # a dead channel
    g = th.Generator().manual_seed(5)
    gout *= (th.rand(2, 1, H, W, generator=g) > 0.2)  # scattered dead pixels
    for align, force, clip in [(False, False, False), (True, False, True), (False, True, False)]:
        want = O.mipmap_grid_sampler_2d(tex, grid, vt, 12, padding, 0, align, force, clip)
        got = capi.mipmap_grid_sampler_2d(dev(tex), dev(grid), dev(vt), 12, padding, 0, align, force, clip)
        close(got, want, "forward")
        wl, wg = O.mipmap_grid_sampler_2d_backward(gout, tex, grid, vt, 12, padding, 0, align, force, clip)
        gl, gg = capi.mipmap_grid_sampler_2d_backward(dev(gout), dev(tex), dev(grid), dev(vt), 12, padding, 0, align, force, clip)
        close(gg, wg, "grad grid", atol=2e-5)
        assert float(gg[:, :16, :16].abs().max()) == 0.0
        for i, (a, b) in enumerate(zip(gl, wl)):
            close(a, b, f"grad level {i}")


@pytest.mark.parametrize("levels", [2, 6])
@pytest.mark.parametrize("padding", [0, 1])
@pytest.mark.parametrize("C", [1, 3, 4, 6])
def test_taps_of_a_pixel_far_apart_on_a_smooth_uv_field(padding, C, levels):
    """The further rounds of the tiled backward place the windows on ONE tap of a tile's pixels when everything pending
    does not fit (the limb of a sphere: a pixel's eight taps spread over hundreds of texels, while tap i of neighbouring
    pixels stay neighbours).  Smooth uv fields that produce exactly that, checked against the oracle:
    * taps along a diagonal (the corner of the bounding box of everything pending holds nothing),
    * a seam through the tiles (one half samples one end of the texture, the other half the other end),
    * a footprint that grows 30x across the image (a tile's pixels on three or four levels, one to eight taps),
    * a cluster of one tap wider than the widest window slot (the window is centred on the seed tap).
    With a pyramid of two levels the level of detail is clipped and the eight taps of a pixel lie 30-60 texels apart
    (spread over 230-460 texels: more than any window); with six levels they are 1-2 texels apart on the pixel's own level
    while the tiles' pixels sit on three different ones."""
    import oracle as O
    from drtk_amd import capi

    H, W, size = 48, 80, 512
    g = th.Generator().manual_seed(77 + 3 * padding + C)
    tex = [th.rand(2, C, size, size, generator=g)]
    for _ in range(levels - 1):
        tex.append(th.nn.functional.avg_pool2d(tex[-1], 2))
    yy, xx = th.meshgrid(th.arange(H, dtype=th.float32), th.arange(W, dtype=th.float32), indexing="ij")
    u = -0.55 + 1.1 * xx / W + 0.02 * yy / H
    v = -0.5 + 1.0 * yy / H + 0.03 * xx / W
    grid = th.stack([th.stack([u, v], -1), th.stack([v * 0.9, -u * 0.8], -1)])  # [2, H, W, 2]
    grid[0, :, W // 2:, 0] -= 1.0                                             # a seam: the right half wraps to the other end
    grid[0, :, W // 2:, 0] += 2.0 * (grid[0, :, W // 2:, 0] < -1).float()
    jac = th.zeros(2, H, W, 2, 2)
    grow = 0.01 * (1.0 + 29.0 * xx / W)                                        # footprint grows 30x from left to right
    jac[0, ..., 0, 0] = 0.55 * grow / grow.max() + 0.05                        # view 0: long along +u +v (diagonal taps)
    jac[0, ..., 0, 1] = 0.45 * grow / grow.max() + 0.04
    jac[0, ..., 1, 0] = -0.004
    jac[0, ..., 1, 1] = 0.005
    jac[1, ..., 1, 0] = 0.9                                                    # view 1: along u, clusters 40+ texels wide
    jac[1, ..., 1, 1] = 0.002
    jac[1, ..., 0, 0] = 0.003 + 0.1 * yy / H                                   # (and a second axis that makes one tap's cluster wide)
    jac[1, ..., 0, 1] = 0.02
    gout = th.rand(2, C, H, W, generator=g) * 2 - 1
    gout[:, :, 8:20, 30:50] = 0                                                # part of the tiles masked
    for force, clip in [(False, False), (True, False), (False, True)]:
        want = O.mipmap_grid_sampler_2d(tex, grid, jac, 8, padding, 0, False, force, clip)
        got = capi.mipmap_grid_sampler_2d(dev(tex), dev(grid), dev(jac), 8, padding, 0, False, force, clip)
        close(got, want, "forward")
        wl, wg = O.mipmap_grid_sampler_2d_backward(gout, tex, grid, jac, 8, padding, 0, False, force, clip)
        gl, gg = capi.mipmap_grid_sampler_2d_backward(dev(gout), dev(tex), dev(grid), dev(jac), 8, padding, 0, False, force, clip)
        close(gg, wg, "grad grid", atol=2e-5)
        for i, (a, b) in enumerate(zip(gl, wl)):
            close(a, b, f"grad level {i}")


def test_non_finite_texels_under_a_zero_weight_stay_out_of_the_result():
    """include/drtk_amd.h, "FINITE TEXELS ASSUMED": a magnified pixel (footprint below one texel) blends level 0 with
    weight 1 and level 1 with weight EXACTLY 0.  The reference evaluates 0 * texel, the oracle restatement with it, so a
    NaN in level 1 poisons that pixel there; the HIP kernels skip a level of weight 0 (forward) and every (pixel, level)
    whose weighted upstream gradient is 0 (backward), so the pixel keeps the value it has with a finite level 1 and the
    poisoned level receives a zero gradient.  Pinned so that the difference is a documented decision, not an accident."""
    import oracle as O
    from drtk_amd import capi

    H, W = 24, 32
    tex, grid, vt, gout = mipmap_inputs(912, 1, 3, 32, 3, H, W, jscale=0.03)
    vt = vt * 0.02  # every pixel magnified: level 0 only
    grid = grid.clamp(-0.9, 0.9)
    clean = capi.mipmap_grid_sampler_2d(dev(tex), dev(grid), dev(vt), 4, 1, 0)
    close(clean, O.mipmap_grid_sampler_2d(tex, grid, vt, 4, 1, 0, False, False, False), "finite texture: same as the oracle")
    bad = [tex[0], th.full_like(tex[1], float("nan")), tex[2]]
    assert bool(th.isnan(O.mipmap_grid_sampler_2d(bad, grid, vt, 4, 1, 0, False, False, False)).any()), "the restated reference propagates 0 * NaN"
    got = capi.mipmap_grid_sampler_2d(dev(bad), dev(grid), dev(vt), 4, 1, 0)
    assert th.equal(got, clean), "a level of weight 0 is not read"
    gl, gg = capi.mipmap_grid_sampler_2d_backward(dev(gout), dev(bad), dev(grid), dev(vt), 4, 1, 0)
    gl_clean, gg_clean = capi.mipmap_grid_sampler_2d_backward(dev(gout), dev(tex), dev(grid), dev(vt), 4, 1, 0)
    assert bool(th.isfinite(gg).all()) and th.equal(gg, gg_clean)
    assert float(gl[1].abs().max()) == 0.0
    close(gl[0], gl_clean[0].cpu(), "gradient of level 0 (sums of float atomics: equal to rounding)")


@pytest.mark.parametrize("dtype", [th.float32, th.float64])
@pytest.mark.parametrize("padding", [0, 1])
@pytest.mark.parametrize("C", [1, 3, 4, 5])
def test_a_non_finite_texel_reaches_only_the_pixels_that_sample_it(padding, C, dtype):
    """Round 5's lean kernels let every lane WITHOUT an interior tap (fewer taps than its wave, a border tap, a NaN
    coordinate, a dead level) run its wave's loads at offset 0 of a level with weight 0: a NaN or Inf in texel (0,0) or
    (1,0) of a level became `NaN * 0` in pixels that never sample it.  The reference touches only the texels it samples:
    with a poisoned texel (0,0) on the two finest levels the NaN pattern of the output equals the oracle's, the other
    pixels are equal to rounding, and the gradients of the pixels that do not sample it stay finite.  (Double runs the
    same lean kernels since round 6; its bars are double's.)"""
    import oracle as O
    from drtk_amd import capi

    H, W = 40, 56
    tex, grid, vt, gout = mipmap_inputs(4100 + 10 * padding + C, 2, C, 64, 4, H, W, jscale=0.03)
    tex, grid, vt, gout = [t.to(dtype) for t in tex], grid.to(dtype), vt.to(dtype), gout.to(dtype)
    tol = dict(atol=1e-5, rtol=1e-5) if dtype == th.float32 else dict(atol=1e-12, rtol=1e-12)
    grid = grid.clone()
    grid[:, :, : W // 2] = grid[:, :, : W // 2].abs().clamp(0.05, 1.15)  # left half: away from texel (0,0), some taps beyond the border
    grid[0, :4, :4] = float("nan")                                      # and a few NaN coordinates
    vt = vt.clone()
    vt[:, :, ::3] *= 6.0                                                # tap counts differ between the lanes of a wave
    bad = [t.clone() for t in tex]
    bad[0][:, :, 0, 0] = float("nan")
    bad[0][:, :, 0, 1] = float("inf")
    bad[1][:, :, 0, 0] = float("nan")
    want = O.mipmap_grid_sampler_2d(bad, grid, vt, 8, padding, 0, False, False, False)
    got = capi.mipmap_grid_sampler_2d(dev(bad), dev(grid), dev(vt), 8, padding, 0, False, False, False).cpu()
    wn, gn = ~th.isfinite(want), ~th.isfinite(got)
    assert bool(wn.any()) and not bool(wn[:, :, :, : W // 2].any()), "scene: some pixels sample the poisoned texel, the left half does not"
    # (a level of weight exactly 0 is skipped by the kernels and multiplied in by the reference: only ever MORE finite pixels here)
    assert not bool((gn & ~wn).any()), f"{int((gn & ~wn).sum())} pixels are non-finite that never sample the poisoned texel"
    ok = ~wn & ~gn
    close(got[ok], want[ok], "forward, pixels that do not sample the poisoned texel", **tol)
    clean = capi.mipmap_grid_sampler_2d(dev(tex), dev(grid), dev(vt), 8, padding, 0, False, False, False).cpu()
    assert th.equal(got[:, :, :, : W // 2], clean[:, :, :, : W // 2]), "pixels away from the poisoned texel: bit-equal to the finite texture's"
    # backward: the grid gradient of a pixel that samples the texel is non-finite in the reference too; the others are finite
    gout = gout.clone()
    gout[:, :, :, W // 2:] = 0  # upstream gradient only where the poisoned texels are not sampled
    wl, wg = O.mipmap_grid_sampler_2d_backward(gout, bad, grid, vt, 8, padding, 0, False, False, False)
    gl, gg = capi.mipmap_grid_sampler_2d_backward(dev(gout), dev(bad), dev(grid), dev(vt), 8, padding, 0, False, False, False)
    # (where the upstream gradient is zero the reference still forms 0 * texel for the grid gradient: NaN on the pixels that
    # sample the poisoned texel; the kernels count the texels of a zero upstream gradient as 0 -- include/drtk_amd.h)
    assert bool(th.isfinite(wg[:, :, : W // 2]).all()), "scene: the reference's grid gradient is finite where the upstream gradient lives"
    close(gg[:, :, : W // 2], wg[:, :, : W // 2], "grad grid", atol=2 * tol["atol"], rtol=tol["rtol"])
    assert float(gg[:, :, W // 2:].abs().max()) == 0.0
    for i, (a, b) in enumerate(zip(gl, wl)):
        close(a, b, f"grad level {i}", **tol)


@pytest.mark.parametrize("dtype", [th.float32, th.float64])
@pytest.mark.parametrize("C", [1, 2, 3, 4, 5])
def test_a_one_by_one_last_level_followed_by_poison_in_memory(C, dtype):
    """A full pyramid (16, 8, 4, 2, 1) carved from ONE buffer, the 1 x 1 level last, NaN (and the bits of int32 -1) right
    behind it: on a 1 x 1 level the pair at offset 0 of the last channel of the last view ends one element beyond the
    tensor -- round 5's lean kernels read it on every lane without an interior tap (every tap of a 1 x 1 level is a border
    tap).  Footprints of 1/4 to 4 texture widths put the pixels on the last levels."""
    import oracle as O
    from drtk_amd import capi

    N, H, W = 2, 24, 40
    sizes = [16, 8, 4, 2, 1]
    g = th.Generator().manual_seed(900 + C)
    n_el = [N * C * s * s for s in sizes]
    ibits = th.int32 if dtype == th.float32 else th.int64
    tol = dict(atol=1e-5, rtol=1e-5) if dtype == th.float32 else dict(atol=1e-12, rtol=1e-12)
    for poison in (float("nan"), th.tensor([-1], dtype=ibits).view(dtype).item()):
        buf = th.full((sum(n_el) + 64,), poison, dtype=dtype)
        tex, off = [], 0
        for s, n in zip(sizes, n_el):
            buf[off: off + n] = th.rand(n, generator=th.Generator().manual_seed(17 * s + C)).to(dtype)
            tex.append(buf[off: off + n].view(N, C, s, s))
            off += n
        grid = (th.rand(N, H, W, 2, generator=g) * 2.2 - 1.1).to(dtype)
        vt = (th.randn(N, H, W, 2, 2, generator=g) * (0.25 + 4.0 * th.rand(N, H, W, 1, 1, generator=g))).to(dtype)
        gout = (th.rand(N, C, H, W, generator=g) * 2 - 1).to(dtype)
        dbuf = buf.to(DEV)
        dtex, off = [], 0
        for s, n in zip(sizes, n_el):
            dtex.append(dbuf[off: off + n].view(N, C, s, s))
            off += n
        for padding in (0, 1):
            for force in (False, True):
                want = O.mipmap_grid_sampler_2d([t.clone() for t in tex], grid, vt, 4, padding, 0, False, force, False)
                got = capi.mipmap_grid_sampler_2d(dtex, dev(grid), dev(vt), 4, padding, 0, False, force, False)
                assert bool(th.isfinite(got).all()), "the poison behind the 1 x 1 level leaked into the output"
                close(got, want, "forward", **tol)
                wl, wg = O.mipmap_grid_sampler_2d_backward(gout, [t.clone() for t in tex], grid, vt, 4, padding, 0, False, force, False)
                gl, gg = capi.mipmap_grid_sampler_2d_backward(dev(gout), dtex, dev(grid), dev(vt), 4, padding, 0, False, force, False)
                close(gg, wg, "grad grid", atol=2 * tol["atol"], rtol=tol["rtol"])
                for i, (a, b) in enumerate(zip(gl, wl)):
                    close(a, b, f"grad level {i}", **tol)


@pytest.mark.parametrize("layout", ["pixel_major", "channel_major"])
@pytest.mark.parametrize("C", [5, 8, 13, 16])
def test_wide_textures_take_the_tiled_backward_once_per_block_of_four_channels(C, layout):
    """C > 4: the lean tile kernel is launched once per block of four channels (tails of 1, 2, 3 channels included); every launch
    after the first ADDS its part of the grid gradient to what the previous ones stored (read-modify-write, in stream order --
    include/drtk_amd.h: grad_grid must not alias an input).  Both layouts of the grid gradient: [N,H,W,2] pairs, and the
    channel-first image seen through permute(0, 2, 3, 1) (two strided stores per pixel); zeros, border and reflection
    padding, the adaptive tap count; against the oracle."""
    import oracle as O
    from drtk_amd import capi

    H, W = 36, 52
    tex, grid, vt, gout = mipmap_inputs(5100 + C, 2, C, 64, 4, H, W, jscale=0.04)
    dgrid = dev(grid)
    if layout == "channel_major":
        dgrid = dgrid.permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1)  # [N,2,H,W] storage
        assert dgrid.stride(3) == H * W and th.equal(dgrid.cpu(), grid)
    for padding in (0, 1, 2):
        want = O.mipmap_grid_sampler_2d(tex, grid, vt, 8, padding, 0, False, False, False)
        close(capi.mipmap_grid_sampler_2d(dev(tex), dgrid, dev(vt), 8, padding, 0, False, False, False), want, f"forward, padding {padding}")
        wl, wg = O.mipmap_grid_sampler_2d_backward(gout, tex, grid, vt, 8, padding, 0, False, False, False)
        gl, gg = capi.mipmap_grid_sampler_2d_backward(dev(gout), dev(tex), dgrid, dev(vt), 8, padding, 0, False, False, False)
        assert gg.stride() == dgrid.stride()
        close(gg, wg, f"grad grid, padding {padding}", atol=2e-5)
        for i, (a, b) in enumerate(zip(gl, wl)):
            close(a, b, f"grad level {i}, padding {padding}")


def test_one_texture_shared_by_all_views_is_sampled_in_place():
    """A [1,C,h,w] pyramid expanded to N views (batch stride 0) -- one texture, many cameras -- gives the results of its
    materialised copy, through the C ABI (`level_sN` = 0) and through the torch op, forward and backward; the level
    gradients come back as contiguous [N,C,h,w] tensors (autograd sums them over the views), and a pyramid whose views are
    spaced wider than C*h*w (a slice of a larger batch) works too."""
    import drtk_amd
    from drtk_amd import capi

    tex1, grid, vt, gout = mipmap_inputs(77, 3, 3, 32, 4, 20, 24)
    tex1 = [t[:1].to(DEV) for t in tex1]
    grid, vt, gout = dev(grid), dev(vt), dev(gout)
    shared = [t.expand(3, -1, -1, -1) for t in tex1]
    copies = [t.contiguous() for t in shared]
    assert shared[0].stride(0) == 0
    want = capi.mipmap_grid_sampler_2d(copies, grid, vt, 6, 1, 0)
    got = capi.mipmap_grid_sampler_2d(shared, grid, vt, 6, 1, 0)
    assert th.equal(got, want)
    wl, wg = capi.mipmap_grid_sampler_2d_backward(gout, copies, grid, vt, 6, 1, 0)
    gl, gg = capi.mipmap_grid_sampler_2d_backward(gout, shared, grid, vt, 6, 1, 0)
    assert th.equal(gg, wg)
    for a, b in zip(gl, wl):
        assert a.is_contiguous() and a.shape == b.shape
        close(a, b, "grad level (atomics: order differs)")
    # views spaced wider than one view: every second entry of a batch of 6
    wide = [th.cat([t, t * 0 - 7.0], 1).reshape(6, *t.shape[1:])[::2] for t in copies]
    assert wide[0].stride(0) == 2 * copies[0].stride(0) and th.equal(wide[0], copies[0])
    assert th.equal(capi.mipmap_grid_sampler_2d(wide, grid, vt, 6, 1, 0), want)
    # the torch op, with autograd: gradient of the shared leaf = sum over the views
    leaves = [t.clone().requires_grad_(True) for t in tex1]
    out = drtk_amd.mipmap_grid_sample([t.expand(3, -1, -1, -1) for t in leaves], grid, vt, 6, padding_mode="border")
    assert th.equal(out, want)
    out.backward(gout)
    for leaf, b in zip(leaves, wl):
        close(leaf.grad, b.sum(0, keepdim=True), "gradient of the shared texture")


@pytest.mark.parametrize("dtype", [th.float32, th.float64])
def test_channel_first_uv_image_is_read_in_place(dtype):
    """grid = uv_img.permute(0, 2, 3, 1) -- how every caller builds it from `interpolate`'s [N,2,H,W] output -- is read
    through its strides (C ABI: grid_layout {2HW, 1, HW}) with the results of its contiguous copy, the grid gradient
    comes back in the same layout (so that it is channel-first contiguous after the permute's backward), a uv image that
    is a channel slice of a wider tensor works too, and layouts the kernels do not take (rows with padding) are copied."""
    import drtk_amd
    from drtk_amd import capi

    tol = dict(atol=1e-5, rtol=1e-5) if dtype == th.float32 else dict(atol=1e-11, rtol=1e-11)
    for C, mode in ((3, 0), (5, 0), (2, 2)):  # tiled backward, direct backward, bicubic
        tex, grid, vt, gout = mipmap_inputs(90 + C, 2, C, 32, 4, 18, 28, dtype)
        tex, grid, vt, gout = dev(tex), dev(grid), dev(vt), dev(gout)
        uv_img = grid.permute(0, 3, 1, 2).contiguous()          # [N,2,H,W]
        cf = uv_img.permute(0, 2, 3, 1)
        assert not cf.is_contiguous() and th.equal(cf, grid)
        want = capi.mipmap_grid_sampler_2d(tex, grid, vt, 6, 1, mode)
        assert th.equal(capi.mipmap_grid_sampler_2d(tex, cf, vt, 6, 1, mode), want)
        wl, wg = capi.mipmap_grid_sampler_2d_backward(gout, tex, grid, vt, 6, 1, mode)
        gl, gg = capi.mipmap_grid_sampler_2d_backward(gout, tex, cf, vt, 6, 1, mode)
        assert gg.stride() == cf.stride() and th.equal(gg, wg)
        for a, b in zip(gl, wl):
            close(a, b, "grad level", **tol)
        wide = th.cat([uv_img * 0 + 9, uv_img, uv_img * 0 - 9], 1)[:, 2:4].permute(0, 2, 3, 1)  # channels 2..3 of 6
        assert th.equal(capi.mipmap_grid_sampler_2d(tex, wide, vt, 6, 1, mode), want)
        padded = th.cat([grid, grid * 0], 2)[:, :, : grid.shape[2]]                              # rows with padding
        assert not padded.is_contiguous() and th.equal(capi.mipmap_grid_sampler_2d(tex, padded, vt, 6, 1, mode), want)
        # the torch op with autograd: the gradient reaches the channel-first leaf contiguous and equal
        leaf = uv_img.clone().requires_grad_(True)
        out = drtk_amd.mipmap_grid_sample(tex, leaf.permute(0, 2, 3, 1), vt, 6, mode="bilinear" if mode == 0 else "bicubic",
                                          padding_mode="border")
        assert th.equal(out, want)
        out.backward(gout)
        assert leaf.grad.is_contiguous() and th.equal(leaf.grad, wg.permute(0, 3, 1, 2))


def test_f64_and_odd_channel_counts_match_oracle():
    import oracle as O
    from drtk_amd import capi

    for C in (1, 5, 9):
        tex, grid, vt, gout = mipmap_inputs(40 + C, 1, C, 16, 2, 9, 7, th.float64)
        want = O.mipmap_grid_sampler_2d(tex, grid, vt, 3, 1, 0, False, False, False)
        got = capi.mipmap_grid_sampler_2d(dev(tex), dev(grid), dev(vt), 3, 1, 0, False, False, False)
        close(got, want, f"forward C={C}", atol=1e-12, rtol=1e-12)
        wl, wg = O.mipmap_grid_sampler_2d_backward(gout, tex, grid, vt, 3, 1, 0, False, False, False)
        gl, gg = capi.mipmap_grid_sampler_2d_backward(dev(gout), dev(tex), dev(grid), dev(vt), 3, 1, 0, False, False, False)
        close(gg, wg, "grad grid", atol=1e-11, rtol=1e-11)
        for a, b in zip(gl, wl):
            close(a, b, "grad level", atol=1e-11, rtol=1e-11)


def test_full_size_properties_and_errors():
    """1024^2 texture with a full pyramid sampled at 2 x 1024^2 pixels: a constant texture is
    reproduced, texture gradients sum to sum(grad_out) (border padding), and a uniformly minified
    lookup equals the matching mip level."""
    import drtk_amd

    g = th.Generator(device=DEV).manual_seed(0)
    N, C, S, H, W = 2, 3, 1024, 1024, 1024
    tex = [th.rand(N, C, S, S, device=DEV, generator=g)]
    while tex[-1].shape[-1] > 1:
        tex.append(th.nn.functional.avg_pool2d(tex[-1], 2))
    assert len(tex) == 11
    grid = th.rand(N, H, W, 2, device=DEV, generator=g) * 2 - 1
    vt = th.randn(N, H, W, 2, 2, device=DEV, generator=g) * 0.004
    const = [th.full_like(t, 0.5) for t in tex]
    out = drtk_amd.mipmap_grid_sample(const, grid, vt, 8, padding_mode="border")
    close(out, th.full_like(out, 0.5), "constant texture")
    leaves = [t.clone().requires_grad_(True) for t in tex]
    out = drtk_amd.mipmap_grid_sample(leaves, grid, vt, 8, padding_mode="border")
    gout = th.rand(out.shape, device=DEV, generator=g)
    out.backward(gout)
    total = sum(float(t.grad.double().sum()) for t in leaves)
    assert abs(total - float(gout.double().sum())) <= 1e-4 * float(gout.double().sum())
    # isotropic footprint of exactly 4 texels per pixel -> level 2 only (a = 0), bilinear lookup of it
    iso = th.zeros(N, H, W, 2, 2, device=DEV)
    iso[..., 0, 0] = 4.0 / S
    iso[..., 1, 1] = 4.0 / S
    got = drtk_amd.mipmap_grid_sample(tex, grid, iso, 1, padding_mode="border")
    want = th.nn.functional.grid_sample(tex[2], grid, mode="bilinear", padding_mode="border", align_corners=False)
    close(got, want, "isotropic minification = grid_sample of level 2", atol=2e-5)

    with pytest.raises(ValueError, match="only 'bilinear' and 'bicubic'"):
        drtk_amd.mipmap_grid_sample(tex, grid, vt, 2, mode="nearest")
    with pytest.raises(ValueError, match="expected padding_mode"):
        drtk_amd.mipmap_grid_sample(tex, grid, vt, 2, padding_mode="wrap")
    with pytest.raises(RuntimeError, match="same batch size"):
        drtk_amd.mipmap_grid_sample(tex, grid[:1], vt, 2)
    with pytest.raises(RuntimeError, match="at least one mipmap level"):
        th.ops.mipmap_grid_sampler_ext.mipmap_grid_sampler_2d([], grid, vt, 2, 0, 0, False, False, False)


@pytest.mark.parametrize("tag", ["f32", "f64"])
def test_uv_derivative_matches_reference_composite_fixture(tag):
    import drtk_amd
    from conftest import load_golden
    from drtk_amd import capi

    i, o = load_golden("uv_derivative_" + tag)
    d = {k: (x.to(DEV) if isinstance(x, th.Tensor) else x) for k, x in i.items()}
    mask = d["index_img"] != -1
    ref = o["vt_dxdy_img"]
    tol = dict(atol=2e-6, rtol=2e-5) if tag == "f32" else dict(atol=1e-12, rtol=1e-11)
    got = capi.screen_space_uv_derivative(d["v"], d["vt"], d["vi"], d["vti"], d["index_img"], d["bary_img"], mask, d["campos"],
                                          d["camrot"], d["focal"])
    close(got, ref, "C ABI", **tol)
    got = drtk_amd.screen_space_uv_derivative(d["v"], d["vt"], d["vi"], d["vti"], d["index_img"], d["bary_img"], mask, d["campos"],
                                              d["camrot"], d["focal"])
    close(got, ref, "python api", **tol)
    # view-shared geometry as a stride-0 batch gives the same result
    v1 = d["v"][:1].expand(d["v"].shape[0], -1, -1)
    vt1 = d["vt"][:1].expand(d["vt"].shape[0], -1, -1)
    close(drtk_amd.screen_space_uv_derivative(v1, vt1, d["vi"], d["vti"], d["index_img"], d["bary_img"], mask, d["campos"],
                                              d["camrot"], d["focal"]), ref, "shared geometry", **tol)
    with pytest.raises(NotImplementedError):
        drtk_amd.screen_space_uv_derivative(d["v"], d["vt"], d["vi"], d["vti"], d["index_img"], d["bary_img"], mask, d["campos"],
                                            d["camrot"], d["focal"], dist_mode=["fisheye"], dist_coeff=d["focal"])


def test_uv_derivative_under_autograd_behaves_like_the_reference():
    """The reference's composite cannot be differentiated: backward() through it raises (in-place mask on the output of
    linalg.inv_ex -- recorded from the reference by oracle/gen_golden_refpy.py).  Same here: the result is part of the
    graph, a loss that only hands it to mipmap_grid_sample (no gradient defined for vt_dxdy_img) is unaffected, a
    gradient that really reaches it raises."""
    import numpy as np
    import drtk_amd
    from conftest import GOLDEN as GOLDEN_DIR, load_golden

    z = np.load(GOLDEN_DIR + "/refpy_uv_derivative_autograd.npz")
    assert bool(z["backward_raises"]) and "inplace operation" in str(z["message"]) and "inplace operation" in str(z["message_all_true_mask"])
    i, o = load_golden("uv_derivative_f32")
    d = {k: (x.to(DEV) if isinstance(x, th.Tensor) else x) for k, x in i.items()}
    mask = d["index_img"] != -1
    v = d["v"].clone().requires_grad_(True)
    vt = d["vt"].clone().requires_grad_(True)
    jac = drtk_amd.screen_space_uv_derivative(v, vt, d["vi"], d["vti"], d["index_img"], d["bary_img"], mask, d["campos"], d["camrot"], d["focal"])
    assert jac.requires_grad
    close(jac, o["vt_dxdy_img"], "forward under autograd", atol=2e-6, rtol=2e-5)
    with pytest.raises(RuntimeError, match="not differentiable"):
        jac.sum().backward()
    # through its consumer: gradients flow to the texture and to the uv image, none is asked of the Jacobians
    N, H, W = d["index_img"].shape
    tex = [th.rand(N, 3, 16 >> l, 16 >> l, device=DEV, requires_grad=True) for l in range(3)]
    uv_img = drtk_amd.interpolate(vt[:, d["vti"].long().reshape(-1)].contiguous(), th.arange(d["vti"].numel(), dtype=th.int32, device=DEV).view(-1, 3),
                                  d["index_img"], d["bary_img"])
    jac = drtk_amd.screen_space_uv_derivative(v, vt, d["vi"], d["vti"], d["index_img"], d["bary_img"], mask, d["campos"], d["camrot"], d["focal"])
    out = drtk_amd.mipmap_grid_sample(tex, (uv_img * 2 - 1).permute(0, 2, 3, 1), jac, 4, padding_mode="border")
    out.square().sum().backward()
    assert all(t.grad is not None and bool(th.isfinite(t.grad).all()) for t in tex) and vt.grad is not None and v.grad is None
    # no gradient required anywhere: a plain tensor
    with th.no_grad():
        assert not drtk_amd.screen_space_uv_derivative(v, vt, d["vi"], d["vti"], d["index_img"], d["bary_img"], mask, d["campos"], d["camrot"], d["focal"]).requires_grad


def test_uv_derivative_equals_finite_differences_of_the_uv_image():
    """Property at full size (2 x 1024^2 views of the 10k sphere): inside a triangle the analytic
    Jacobian matches central differences of the perspective-correct uv image."""
    import drtk_amd
    from drtk_amd import synthetic as S

    N, H, W = 2, 1024, 1024
    nl, no = S.MESH_SIZES["10k"]
    v, vi = S.uv_sphere(nl, no, device=DEV)
    campos, camrot, focal, princpt = S.ring_cameras(N, W, H, device=DEV)
    vN = v[None].expand(N, -1, -1)
    v_pix = drtk_amd.transform(vN, campos, camrot, focal, princpt)
    index = drtk_amd.rasterize(v_pix, vi, H, W)
    _, bary = drtk_amd.render(v_pix, vi, index)
    vid = th.arange(v.shape[0], device=DEV)
    vt = th.stack([(vid % no).float() / no * 0.8 + 0.1, (vid // no).float() / nl * 0.8 + 0.1], -1)[None].expand(N, -1, -1)
    mask = index != -1
    jac = drtk_amd.screen_space_uv_derivative(vN, vt, vi, vi, index, bary, mask, campos, camrot, focal)
    uv = drtk_amd.interpolate(vt.contiguous(), vi, index, bary).permute(0, 2, 3, 1)
    same_x = (index[:, :, 2:] == index[:, :, :-2]) & (index[:, :, 1:-1] == index[:, :, 2:]) & mask[:, :, 1:-1]
    fdx = (uv[:, :, 2:] - uv[:, :, :-2]) / 2
    err = (jac[:, :, 1:-1, 0, :] - fdx).abs()[same_x]
    scale = float(jac[mask].abs().max())
    assert same_x.sum() > 100000 and float(err.max()) < 2e-2 * scale and float(err.mean()) < 1e-3 * scale
    same_y = (index[:, 2:] == index[:, :-2]) & (index[:, 1:-1] == index[:, 2:]) & mask[:, 1:-1]
    fdy = (uv[:, 2:] - uv[:, :-2]) / 2
    err = (jac[:, 1:-1, :, 1, :] - fdy).abs()[same_y]
    assert float(err.max()) < 2e-2 * scale and float(err.mean()) < 1e-3 * scale
    assert float(jac[~mask].abs().sum()) == 0.0


def test_textured_shading_with_fp16_attributes_under_autocast():
    """BASELINE.json configs[4] in small: rasterize -> render -> interpolate(uv) -> uv Jacobian ->
    mipmap_grid_sample -> edge_grad_estimator with the uv attributes and the texture pyramid stored in
    fp16, run under autocast.  Like the reference (Autocast keys: cached_cast to float32 at every op,
    interpolate_module.cpp:584-600, mipmap_grid_sampler_module.cpp autocast wrapper) the ops compute in f32:
    outputs equal the f32 pipeline on the same (fp16-representable) values bit for bit, and the leaves
    receive fp16 gradients equal to the rounded f32 ones."""
    import drtk_amd
    from drtk_amd import synthetic as S

    N, H, W = 2, 192, 256
    nl, no = 24, 28
    v, vi = S.uv_sphere(nl, no, device=DEV)
    campos, camrot, focal, princpt = S.ring_cameras(N, W, H, device=DEV)
    vN = v[None].expand(N, -1, -1).contiguous()
    vid = th.arange(v.shape[0], device=DEV)
    vt16 = th.stack([(vid % no).float() / no * 0.8 + 0.1, (vid // no).float() / nl * 0.8 + 0.1], -1)[None].expand(N, -1, -1).half()
    g = th.Generator(device=DEV).manual_seed(2)
    tex16 = [th.rand(N, 3, 64, 64, device=DEV, generator=g).half()]
    while tex16[-1].shape[-1] > 4:
        tex16.append(th.nn.functional.avg_pool2d(tex16[-1].float(), 2).half())

    def pipeline(vt, tex, v_in, autocast):
        with th.autocast("cuda", dtype=th.float16, enabled=autocast):
            v_pix = drtk_amd.transform(v_in, campos, camrot, focal, princpt)
            index = drtk_amd.rasterize(v_pix, vi, H, W)
            _, bary = drtk_amd.render(v_pix, vi, index)
            uv = drtk_amd.interpolate(vt, vi, index, bary)
            mask = index != -1
            jac = drtk_amd.screen_space_uv_derivative(v_in, vt.float(), vi, vi, index, bary, mask, campos, camrot, focal)
            grid = (uv.permute(0, 2, 3, 1) * 2 - 1) * mask[..., None]
            img = drtk_amd.mipmap_grid_sample(tex, grid, jac, 4, padding_mode="border") * mask[:, None]
            img = drtk_amd.edge_grad_estimator(v_pix=v_pix, vi=vi, bary_img=bary, img=img, index_img=index)
            return index, uv, img

    vt_h = vt16.clone().requires_grad_(True)
    tex_h = [t.clone().requires_grad_(True) for t in tex16]
    v_h = vN.clone().requires_grad_(True)
    index_h, uv_h, img_h = pipeline(vt_h, tex_h, v_h, True)
    assert uv_h.dtype == th.float32 and img_h.dtype == th.float32
    (img_h.square().mean()).backward()

    vt_f = vt16.float().requires_grad_(True)
    tex_f = [t.float().requires_grad_(True) for t in tex16]
    v_f = vN.clone().requires_grad_(True)
    index_f, uv_f, img_f = pipeline(vt_f, tex_f, v_f, False)
    (img_f.square().mean()).backward()

    assert th.equal(index_h, index_f) and th.equal(uv_h, uv_f) and th.equal(img_h, img_f)
    assert int((index_f != -1).sum()) > 0.3 * index_f.numel() and float(img_f.detach().abs().max()) > 0.1
    assert vt_h.grad.dtype == th.float16 and all(t.grad.dtype == th.float16 for t in tex_h)
    close(vt_h.grad, vt_f.grad, "uv attribute gradient", atol=1e-6, rtol=2e-3)  # fp16 rounding of the f32 gradient
    for a, b in zip(tex_h, tex_f):
        close(a.grad, b.grad, "texture gradient", atol=1e-7, rtol=2e-3)
    close(v_h.grad, v_f.grad, "vertex gradient", atol=1e-7, rtol=1e-4)


@pytest.mark.parametrize("block", range(3))
def test_randomised_mipmap_cases(block):
    """20 seeded cases per block from tests/fuzz_mipmap.py (odd / non-square textures and outputs, 1..11
    levels, 1..6 channels, smooth and incoherent uv fields, every mode flag) against the oracle."""
    import fuzz_mipmap as F

    for seed in range(20 * block, 20 * block + 20):
        c = F.make_case(seed)
        try:
            F.run_case(c)
        except AssertionError as e:
            raise AssertionError(f"seed {seed}: {F.describe(c)}: {e}") from e


def near_isotropic_case(seed=0, H=128, W=128, tex=64):
    """Footprints within a few ulp of ISOTROPIC (what a surface seen face-on produces): du/dx = p/tex * (1 + k1 ulp),
    dv/dy = p/tex * (1 + k2 ulp), k in -3..3.  The tap count ceil(p_max / p_min) jumps from 1 to 2 as that ratio
    crosses 1 -- measured on the oracle, 0.2 in the output of a [0,1] texture for a 3e-7 change of one length."""
    g = th.Generator().manual_seed(seed)
    base = th.rand(1, 3, tex, tex, generator=g)
    levels = [base]
    while levels[-1].shape[-1] > 1:
        levels.append(th.nn.functional.avg_pool2d(levels[-1], 2))
    levels = [lv.contiguous() for lv in levels]
    grid = th.rand(1, H, W, 2, generator=g) * 1.8 - 0.9
    p = 0.5 + 7.5 * th.rand(1, H, W, generator=g)
    ulp = 2.0 ** -23
    k1 = th.randint(-3, 4, (1, H, W), generator=g).float()
    k2 = th.randint(-3, 4, (1, H, W), generator=g).float()
    jac = th.zeros(1, H, W, 2, 2)
    jac[..., 0, 0] = (p / tex) * (1 + k1 * ulp)
    jac[..., 1, 1] = (p / tex) * (1 + k2 * ulp)
    return levels, grid, jac


def test_tap_count_at_near_isotropic_footprints_follows_the_reference():
    """The anisotropic tap count is a DISCONTINUOUS function of the two footprint lengths (sqrt of a sum of squares
    each): with a square root that is only 1 ulp accurate (`__fsqrt_rn` = the native root on this toolchain) two
    lengths that round to the same float can come out different, the ratio leaves 1 and the kernel takes 2 taps where
    the reference takes 1.  sqrt, /, ceil are IEEE on both sides now (profiles/NOTES.md section 3); the level selection, by
    contrast, is continuous across integer levels (floor + blend weight) and needs no such care."""
    import oracle as O
    from drtk_amd import capi

    for seed in range(4):
        levels, grid, jac = near_isotropic_case(seed)
        for mode in (0, 2):  # bilinear, bicubic
            want = O.mipmap_grid_sampler_2d(levels, grid, jac, 4, 0, mode)
            got = capi.mipmap_grid_sampler_2d([t.to(DEV) for t in levels], grid.to(DEV), jac.to(DEV), 4, 0, mode).cpu()
            err = (got - want).abs().amax(1)
            nbad = int((err > 1e-5).sum())
            assert nbad == 0, f"seed {seed} mode {mode}: {nbad} of {err.numel()} pixels differ, worst {float(err.max()):.3e} (a tap-count flip is ~1e-1)"


def test_zero_sized_dimensions_of_the_texture_and_sparse_ops():
    """mipmap_grid_sample, transform, screen_space_uv_derivative and the sparse interpolation operators with each
    dimension in turn set to zero, forward and backward through the Python API: right shapes, finite values, no
    error -- an empty tensor's null pointer is not a missing argument (the backward of the normal-matrix values
    rejected a mesh without faces while its forward accepted it).  Faces WITHOUT vertices are rejected on purpose:
    every index would be out of range."""
    import drtk_amd
    from drtk_amd import synthetic as S

    def finite(*ts):
        for t in ts:
            d = t.to_dense() if t.layout != th.strided else t
            assert d.numel() == 0 or bool(th.isfinite(d.float()).all())

    for zero in ("N", "C", "H", "W"):
        d = dict(N=2, C=3, H=5, W=7)
        d[zero] = 0
        N, C, H, W = d["N"], d["C"], d["H"], d["W"]
        for mode in ("bilinear", "bicubic"):
            tex = [th.rand(N, C, 16, 16, device=DEV).requires_grad_(True), th.rand(N, C, 8, 8, device=DEV).requires_grad_(True)]
            grid = (th.rand(N, H, W, 2, device=DEV) * 2 - 1).requires_grad_(True)
            jac = th.randn(N, H, W, 2, 2, device=DEV) * 0.05
            out = drtk_amd.mipmap_grid_sample(tex, grid, jac, 4, mode=mode, padding_mode="border")
            assert tuple(out.shape) == (N, C, H, W)
            out.sum().backward()
            assert grid.grad.shape == grid.shape and tex[0].grad.shape == tex[0].shape and tex[1].grad.shape == tex[1].shape
            finite(out, grid.grad, tex[0].grad, tex[1].grad)

    v0, vi = S.uv_sphere(6, 8, device=DEV)
    tri = vi.shape[0] // 2  # a mid-latitude triangle: the first row touches the pole and has zero area (singular uv Jacobian)
    for zero in ("N", "H", "W", "V", "F"):
        N, H, W = (0 if zero == "N" else 2), (0 if zero == "H" else 12), (0 if zero == "W" else 16)
        v = v0[:0] if zero == "V" else v0
        f = vi[:0] if zero == "F" else vi
        V = v.shape[0]
        cams = S.ring_cameras(N, max(W, 1), max(H, 1), device=DEV)
        vN = v[None].expand(N, -1, -1).contiguous().requires_grad_(True)
        v_pix = drtk_amd.transform(vN, *cams)
        assert tuple(v_pix.shape) == (N, V, 3)
        v_pix.sum().backward()
        assert vN.grad.shape == vN.shape
        finite(v_pix, vN.grad)

        index = th.full((N, H, W), -1, dtype=th.int32, device=DEV)
        if f.shape[0] and V and index.numel():
            index.view(-1)[::2] = tri
        bary = th.rand(N, 3, H, W, device=DEV)
        vt = th.rand(N, V, 2, device=DEV)
        uv_args = (vN.detach(), vt, f, f, index, bary, index != -1, cams[0], cams[1], cams[2])
        if zero == "V":
            with pytest.raises(RuntimeError, match="invalid argument"):
                drtk_amd.screen_space_uv_derivative(*uv_args)
            with pytest.raises(RuntimeError, match="expected num_vertices to be positive"):
                drtk_amd.interpolation_normal_matrix(f, index, bary, V)
            continue
        jac = drtk_amd.screen_space_uv_derivative(*uv_args)
        assert tuple(jac.shape) == (N, H, W, 2, 2)
        finite(jac)
        b = bary.clone().requires_grad_(True)
        A = drtk_amd.interpolation_matrix(f, index, b, V)
        M = drtk_amd.interpolation_normal_matrix(f, index, b, V)
        assert A.shape[1] == V and tuple(M.shape) == (V, V)
        (A.values().sum() + M.values().sum()).backward()
        assert b.grad.shape == b.shape
        finite(A, M, b.grad)
        if zero == "F":
            assert float(b.grad.abs().sum()) == 0.0 and A.shape[0] == 0


def test_randomised_mipmap_cases_with_inputs_at_odd_element_offsets():
    """The sampler reads `grid` as 8-byte and `vt_dxdy_img` as 16-byte vectors per pixel; with contiguous inputs that
    are only element-aligned (views one element into a flat buffer) the results are those of the aligned call
    (platform support for dword-aligned wide accesses; include/drtk_amd.h, alignment)."""
    import fuzz_all_ops as FA
    import fuzz_mipmap as F

    for seed in range(15):
        c = F.make_case(seed)
        try:
            F.run_case(c, place=FA.misaligned)
        except Exception as e:
            raise AssertionError(f"seed {seed}: {F.describe(c)}: {type(e).__name__}: {e}") from e


@pytest.mark.parametrize("block", range(3))
def test_randomised_sparse_uv_and_transform_cases(block):
    """16 seeded cases per block from tests/fuzz_next_ops.py: the sparse interpolation operators (structure and
    interpolation-matrix values bit-exact, the device-built A^T A pattern identical to the restated one, accumulated
    values at the usual bar), screen_space_uv_derivative (f64: 1e-10 against the restated composite; f32: as accurate
    against the f64 result as the reference formulation itself -- the op is ill-conditioned at grazing triangles) and
    transform + its gradient against the f64 PyTorch formulation; awkward image sizes, shared / per-view topology."""
    import fuzz_next_ops as F

    for seed in range(16 * block, 16 * block + 16):
        c = F.make_case(seed)
        try:
            F.run_case(c)
        except Exception as e:
            raise AssertionError(f"seed {seed}: {F.describe(c)}: {type(e).__name__}: {e}") from e


def test_inconsistent_shapes_and_dtypes_are_rejected_by_the_sampler_and_the_sparse_ops():
    """The relations the reference checks between the tensors of one call (mipmap_grid_sampler_kernel.cu:911-990,
    interpolate_kernel.cu:703-836), with its messages -- plus two it does not check and would index out of bounds on:
    the spatial size of vt_dxdy_img against grid, and vertex indices beyond num_vertices in the A^T A pattern."""
    import drtk_amd  # noqa: F401  (registers the operators)

    N, C, H, W, V, F = 2, 3, 6, 10, 9, 6
    g = th.Generator(device=DEV).manual_seed(0)

    def tex(n=N, c=C, dt=th.float32):
        return [th.rand(n, c, 16, 16, device=DEV, dtype=dt), th.rand(n, c, 8, 8, device=DEV, dtype=dt)]

    grid = th.rand(N, H, W, 2, device=DEV) * 2 - 1
    jac = th.randn(N, H, W, 2, 2, device=DEV) * 0.05
    M = th.ops.mipmap_grid_sampler_ext.mipmap_grid_sampler_2d
    m = lambda t, gr, j: M(t, gr, j, 4, 1, 0, False, False, False)  # noqa: E731
    vi = th.randint(0, V, (N, F, 3), device=DEV, generator=g).int()
    index = th.randint(-1, F, (N, H, W), device=DEV, generator=g).int()
    bary = th.rand(N, 3, H, W, device=DEV, generator=g)
    pair = th.zeros(N, F, 9, dtype=th.int32, device=DEV)
    IM, NM = th.ops.interpolate_ext.interpolation_matrix, th.ops.interpolate_ext.interpolation_normal_matrix
    NV = th.ops.interpolate_ext.interpolation_normal_matrix_values
    one = lambda t: t[:1].contiguous()  # noqa: E731
    S = r"mipmap_aniso_grid_sampler_2d\(\): "
    cases = [
        (S + "expected input to have at least one mipmap level", lambda: m([], grid, jac)),
        (S + "at most 11 mipmap levels", lambda: m([th.rand(N, C, 4, 4, device=DEV)] * 12, grid, jac)),
        (S + "expected grid, vt_dxdy_img and input to have same batch size", lambda: m(tex(), one(grid), one(jac))),
        (S + "expected grid, vt_dxdy_img and input to have same batch size", lambda: m(tex(), grid, one(jac))),
        (S + "expected grid to have size 2 in last dimension", lambda: m(tex(), th.rand(N, H, W, 3, device=DEV), jac)),
        (S + "expected vt_dxdy_img to have size 2 in last two dimension", lambda: m(tex(), grid, th.rand(N, H, W, 2, 3, device=DEV))),
        (S + "expected 4D input and grid with same number of dimensions and 5D vt_dxdy_img", lambda: m(tex(), grid, jac[..., 0])),
        (S + "expected 4D input and grid with same number of dimensions and 5D vt_dxdy_img", lambda: m(tex(), grid[..., 0], jac)),
        (S + "expected input and grid to have same dtype", lambda: m(tex(dt=th.float64), grid, jac)),
        (S + "expected all inputs to have same device, dtype, layout", lambda: m([tex()[0], th.rand(1, C, 8, 8, device=DEV)], grid, jac)),
        (S + "expected all inputs to have same device, dtype, layout", lambda: m([tex()[0], th.rand(N, C + 1, 8, 8, device=DEV)], grid, jac)),
        (S + "expected all inputs to have same device, dtype, layout", lambda: m([tex()[0], th.rand(N, C, 8, 8, device=DEV, dtype=th.float64)], grid, jac)),
        (r"grid_sampler\(\): expected input to have non-empty spatial dimensions", lambda: m([th.rand(N, C, 0, 16, device=DEV)], grid, jac)),
        (S + "expected vt_dxdy_img to match grid in device, dtype and spatial size", lambda: m(tex(), grid, jac[:, : H - 2].contiguous())),
        (S + "expected vt_dxdy_img to match grid in device, dtype and spatial size", lambda: m(tex(), grid[:, : H - 2].contiguous(), jac)),
        (r"interpolation_matrix\(\): expected vi, index_img and bary_img shapes to agree", lambda: IM(vi, one(index), bary)),
        (r"interpolation_matrix\(\): expected vi, index_img and bary_img shapes to agree", lambda: IM(vi, index[:, : H - 1].contiguous(), bary)),
        (r"interpolation_matrix\(\): expected vi, index_img and bary_img shapes to agree", lambda: IM(vi, index, th.rand(N, 4, H, W, device=DEV))),
        (r"interpolation_matrix\(\): expected vi, index_img and bary_img shapes to agree", lambda: IM(th.zeros(N, F, 4, dtype=th.int32, device=DEV), index, bary)),
        (r"interpolation_matrix\(\): expected vi, index_img and bary_img shapes to agree", lambda: IM(one(vi), index, bary)),
        (r"interpolation_matrix\(\): expected bary_img to have floating point type", lambda: IM(vi, index, bary.int())),
        (r"interpolation_normal_matrix\(\): expected vi, index_img and bary_img shapes to agree", lambda: NM(vi, index[:, :, : W - 1].contiguous(), bary, V)),
        (r"interpolation_normal_matrix\(\): vi contains a vertex index outside", lambda: NM(vi, index, bary, 3)),
        (r"interpolation_normal_matrix_values\(\): expected pair_indices \[N,F,9\]", lambda: NV(pair[..., :8].contiguous(), index, bary, 10)),
        (r"interpolation_normal_matrix_values\(\): expected pair_indices, index_img and bary_img shapes to agree", lambda: NV(one(pair), index, bary, 10)),
        (r"interpolation_normal_matrix_values\(\): expected pair_indices to have int32 type", lambda: NV(pair.long(), index, bary, 10)),
        (r"interpolation_normal_matrix_values\(\): expected pair_indices, index_img and bary_img shapes to agree", lambda: NV(pair, index[:, : H - 1].contiguous(), bary, 10)),
        (r"interpolation_normal_matrix_values\(\): expected nnz to be non-negative", lambda: NV(pair, index, bary, -1)),
    ]
    for pattern, fn in cases:
        with pytest.raises(RuntimeError, match=pattern):
            fn()
    assert tuple(m(tex(), grid, jac).shape) == (N, C, H, W)


def test_kernel_gives_the_hand_derived_known_answers():
    """Adaptive tap count, level selection / clipping, clip_grad and degenerate footprints of the HIP sampler against
    answers derived on paper from the CUDA source (tests/mipmap_known_answers.py) -- through the C ABI and through
    drtk_amd.mipmap_grid_sample."""
    import drtk_amd
    import mipmap_known_answers as K
    from drtk_amd import capi

    def via_capi(levels, grid, jac, max_aniso, padding, interp, align, force, clip):
        return capi.mipmap_grid_sampler_2d([dev(t) for t in levels], dev(grid), dev(jac), max_aniso, padding, interp, align, force, clip).cpu()

    def via_python_api(levels, grid, jac, max_aniso, padding, interp, align, force, clip):
        assert padding == 1 and interp == 0
        return drtk_amd.mipmap_grid_sample([dev(t) for t in levels], dev(grid), dev(jac), max_aniso, padding_mode="border",
                                           align_corners=align, force_max_aniso=force, clip_grad=clip).cpu()

    for dt in (th.float32, th.float64):
        K.run_A(via_capi, dt)
        K.run_B(via_capi, dt)
    K.run_A(via_python_api)
    K.run_B(via_python_api)

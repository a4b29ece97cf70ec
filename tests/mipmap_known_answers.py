"""TEST INFRASTRUCTURE -- hand-derived known answers for the discrete decisions of mipmap_grid_sampler_2d that the
reference's pure-PyTorch model does not cover (it equals the CUDA kernel only for force_max_aniso=True, clip_grad=False,
drtk/mipmap_grid_sample.py:140-146): the ADAPTIVE tap count, the level selection incl. its clipping to the pyramid,
`clip_grad`, degenerate footprints.  Derived on paper from mipmap_grid_sampler_kernel.cu:455-507:

    px = |(du/dx * W, dv/dx * H)|,  py = |(du/dy * W, dv/dy * H)|            (texels; + 1e-12 under the root)
    N  = min(ceil(p_max / p_min), max_aniso);  N = 1 if p_min == 0
    lambda = log2(p_max / N);  0 if nan / inf
    l  = max(min(lambda, mipmaps - 1 - 1e-6), 0);  d1 = floor(l);  a = l - d1
    out = sum over taps i < N_int of  (1 - a) / N_int * level[d1](tap_i)  +  a / N_int * level[d1 + 1](tap_i)
    tap_i = (u, v) + (du/dx, dv/dx) * ((i + 1) / (N_int + 1) * 2 - 1)   along the major axis;  N_int = max_aniso if force_max_aniso
    clip_grad and lambda > mipmaps - 1:  the Jacobian is scaled by 2^l * N / p_max before the taps are placed

Case family A -- level l of the pyramid is the CONSTANT image l: whatever the taps, out = l, i.e. the output IS the
selected (fractional) level, and N enters it through lambda.  Case family B -- a one-level texture with value x^2 in
texel column x and taps that land on (or a quarter texel off) texel centres: the output is a mean of known numbers
and depends on where the taps are, which is what clip_grad changes."""
import math

import torch as th


def constant_pyramid(size, levels, dtype=th.float32, channels=2):
    return [th.full((1, channels, size >> k, size >> k), float(k), dtype=dtype) for k in range(levels)]


# (name, texels along x per pixel step [du/dx * W], texels along y [dv/dy * H], max_aniso, force, levels) -> expected level
A_CASES = [
    # isotropic footprints: N = 1, lambda = log2(p)
    ("iso_1", 1, 1, 4, False, 5, 0.0),
    ("iso_2", 2, 2, 4, False, 5, 1.0),
    ("iso_4", 4, 4, 8, False, 5, 2.0),
    ("iso_half", 0.5, 0.5, 4, False, 5, 0.0),               # lambda = -1 -> clamped to 0 (magnification)
    # anisotropic 4 : 1 -- the tap count decides the level: N = min(4, max_aniso)
    ("aniso4_max1", 4, 1, 1, False, 5, 2.0),                # N = 1: log2(4)
    ("aniso4_max2", 4, 1, 2, False, 5, 1.0),                # N = 2: log2(2)
    ("aniso4_max3", 4, 1, 3, False, 5, math.log2(4 / 3)),   # N = 3
    ("aniso4_max4", 4, 1, 4, False, 5, 0.0),                # N = 4: log2(1)
    ("aniso4_max8", 4, 1, 8, False, 5, 0.0),                # N = ceil(4) = 4, not 8
    ("aniso_y_major", 1, 4, 8, False, 5, 0.0),              # major axis y
    # non-integer ratio: N = ceil(3 / 2) = 2 -> log2(1.5);  ratio 2.5 -> N = 3 -> log2(5 / 3) - 0 with p_max = 5, p_min = 2
    ("ratio_1p5", 3, 2, 8, False, 5, math.log2(3 / 2)),
    ("ratio_2p5", 5, 2, 8, False, 5, math.log2(5 / 3)),
    # force_max_aniso changes the number of TAPS, not N in lambda (:500-503): same level as aniso4_max8
    ("forced_taps_same_level", 4, 1, 8, True, 5, 0.0),
    # level clipped to the pyramid: lambda = log2(64) = 6 > levels - 1 -> l = levels - 1 - 1e-6
    ("clipped_to_pyramid", 64, 64, 4, False, 3, 2.0 - 1e-6),
    # degenerate footprint: p_min = 0 (+1e-12 under the root: 1e-6 texels) -> ceil(p_max / p_min) is huge -> N = max_aniso
    ("zero_minor_axis", 4, 0, 4, False, 5, 0.0),
    ("zero_footprint", 0, 0, 4, False, 5, 0.0),             # p = 1e-6 both: N = 1, lambda = log2(1e-6) < 0 -> 0
]


def run_A(sample, dtype=th.float32):
    """sample(levels, grid [1,H,W,2], vt_dxdy_img [1,H,W,2,2], max_aniso, padding_mode=1 (border), interp 0, align, force, clip) -> [1,C,H,W]"""
    size = 64
    for name, tx, ty, max_aniso, force, levels, want in A_CASES:
        tex = constant_pyramid(size, levels, dtype)
        grid = th.tensor([[[[0.1, -0.2], [0.5, 0.5]]]], dtype=dtype)  # two pixels, anywhere inside
        jac = th.zeros(1, 1, 2, 2, 2, dtype=dtype)
        jac[..., 0, 0] = tx / size  # du/dx
        jac[..., 1, 1] = ty / size  # dv/dy
        for clip in (False, True):  # constant levels: clip_grad moves taps only, the level is the same
            out = sample(tex, grid, jac, max_aniso, 1, 0, False, force, clip)
            assert out.shape == (1, 2, 1, 2)
            err = float((out.double() - want).abs().max())
            assert err <= 2e-6, f"{name} (clip_grad={clip}): selected level {float(out[0, 0, 0, 0]):.7f}, derived {want:.7f}"
    # rotated footprint: px = |(3, 4)| = 5 texels, py = |(-0.8, 0.6)| * 2.5 = 2.5 -> N = 2, lambda = log2(2.5)
    tex = constant_pyramid(size, 5, dtype)
    jac = th.zeros(1, 1, 1, 2, 2, dtype=dtype)
    jac[0, 0, 0, 0] = th.tensor([3.0, 4.0], dtype=dtype) / size          # (du/dx, dv/dx)
    jac[0, 0, 0, 1] = th.tensor([-2.0, 1.5], dtype=dtype) / size         # (du/dy, dv/dy)
    out = sample(tex, th.zeros(1, 1, 1, 2, dtype=dtype), jac, 8, 1, 0, False, False, False)
    assert abs(float(out[0, 0, 0, 0]) - math.log2(2.5)) <= 2e-6


def run_B(sample, dtype=th.float32):
    """One 16 x 16 level, value x^2 in column x; pixel at the centre of texel column 8 (u = 1/16 with
    align_corners=False: ix = ((u + 1) * 16 - 1) / 2 = 8), row centre of texel 5 (v = -5/16).  du/dx = 1/4 -> px = 4 texels;
    dv/dy = 1/16 -> py = 1; max_aniso = 3 -> N = 3, lambda = log2(4 / 3) > 0 = mipmaps - 1 -> clipped, a = 0, d1 = 0.
    Taps at u + du/dx * (-1/2, 0, 1/2) = u -+ 1/8 in grid units = -+ 1 texel: columns 7, 8, 9 -> (49 + 64 + 81) / 3.
    clip_grad: scaling = 2^(-1e-6) * 3 / 4 -> -+ 0.75 texel: bilinear x^2 at 7.25 = 52.75, at 8.75 = 76.75 ->
    (52.75 + 64 + 76.75) / 3 = 64.5 (the 2^(-1e-6) moves it by < 1e-4)."""
    x = th.arange(16, dtype=dtype)
    tex = [(x * x)[None, None, None, :].expand(1, 1, 16, 16).contiguous()]
    grid = th.tensor([[[[1.0 / 16, -5.0 / 16]]]], dtype=dtype)
    jac = th.zeros(1, 1, 1, 2, 2, dtype=dtype)
    jac[..., 0, 0] = 0.25
    jac[..., 1, 1] = 1.0 / 16
    plain = float(sample(tex, grid, jac, 3, 1, 0, False, False, False)[0, 0, 0, 0])
    clipped = float(sample(tex, grid, jac, 3, 1, 0, False, False, True)[0, 0, 0, 0])
    assert abs(plain - (49 + 64 + 81) / 3) <= 1e-4, plain
    assert abs(clipped - 64.5) <= 2e-4, clipped
    # force_max_aniso with max_aniso = 7: taps at -+ (3/4, 1/2, 1/4) * 2 texels and 0 -> columns 6.5 .. 9.5 in half-texel steps
    forced = float(sample(tex, grid, jac, 7, 1, 0, False, True, False)[0, 0, 0, 0])
    cols = [8 + 2 * ((i + 1) / 8 * 2 - 1) for i in range(7)]  # 6.5, 7, 7.5, 8, 8.5, 9, 9.5
    bil = lambda c: (1 - (c - math.floor(c))) * math.floor(c) ** 2 + (c - math.floor(c)) * (math.floor(c) + 1) ** 2  # noqa: E731
    assert abs(forced - sum(bil(c) for c in cols) / 7) <= 1e-4, forced

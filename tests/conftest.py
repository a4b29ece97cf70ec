import os
import sys

# the ctypes binding pre-fills every output it allocates with NaN / sentinels (drtk_amd/capi.py _out): an element a kernel
# forgot to write cannot pass for a value
os.environ.setdefault("DRTK_CAPI_POISON", "1")

import numpy as np
import pytest
import torch as th

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True)
def _capi_guards_intact():
    """With DRTK_CAPI_GUARD=g in the environment (a diagnostic run of the whole GPU suite: every output and workspace of the
    ctypes binding sits between g sentinel elements and is merely element-aligned), check after each test that no kernel
    wrote outside what it was given.  Without the variable: nothing."""
    yield
    if os.environ.get("DRTK_CAPI_GUARD", "0") not in ("", "0"):
        capi = sys.modules.get("drtk_amd.capi")
        if capi is not None and th.cuda.is_available():
            capi.check_guards()


def load_golden(name):
    """Returns (inputs, outputs) dicts of torch tensors / python ints from tests/golden/<name>.npz."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    ins, outs = {}, {}
    for k in z.files:
        a = z[k]
        val = int(a) if (a.ndim == 0 and a.dtype.kind in "iu") else th.from_numpy(np.ascontiguousarray(a))
        (ins if k.startswith("in_") else outs)[k.split("_", 1)[1]] = val
    return ins, outs


GOLDEN_SCENES = [
    "two_triangles_f32", "tutorial3_f32", "tutorial3_f64", "spheres_f32", "spheres_f64", "spheres_c16_f32",
    "ragged_f32", "edge_cases_f32", "edge_cases_f64",
]


SPARSE_SCENES = ["spheres_f32", "ragged_f32", "tutorial3_f64"]


def load_sparse(name):
    """(vi [N,F,3] contiguous, index_img, bary_img, V, seeded grads, expected outputs) for tests/golden/sparse_<name>.npz;
    the images are the reference outputs stored in the scene's own fixture."""
    i, o = load_golden(name)
    gi, go = load_golden("sparse_" + name)
    vi, index = i["vi"], o["index_img"]
    vib = (vi[None].expand(index.shape[0], -1, -1) if vi.ndim == 2 else vi).contiguous()
    return vib, index, o["render_bary"], i["v"].shape[1], gi, go


MIPMAP_CASES = [
    "bilinear_border_a4", "bilinear_zeros_a1", "bilinear_reflection_a2", "bicubic_border_a3", "bicubic_zeros_a2_f64",
    "bilinear_border_a8_f64", "single_level_a2",
]
# force_max_aniso=False: the reference model's outputs assembled per tap-count class (oracle/gen_golden_mipmap.py --adaptive)
MIPMAP_ADAPTIVE_CASES = [
    "adaptive_bilinear_border_a4", "adaptive_bilinear_zeros_a8", "adaptive_bicubic_border_a3",
    "adaptive_bilinear_reflection_a6_f64", "adaptive_bicubic_zeros_a5_f64",
]


def load_mipmap(name):
    """tests/golden/mipmap_<name>.npz -> dict(tex=[levels], grid, vt, grad_out, max_aniso, mode, padding,
    out, grad_tex=[...], grad_grid): inputs and the outputs / autograd gradients of the reference's
    pure-PyTorch model (oracle/gen_golden_mipmap.py; force_max_aniso=True, clip_grad=False)."""
    z = np.load(os.path.join(GOLDEN, "mipmap_" + name + ".npz"))
    L = int(z["in_levels"])
    t = lambda k: th.from_numpy(np.ascontiguousarray(z[k]))  # noqa: E731
    return dict(
        tex=[t(f"in_tex{i}") for i in range(L)], grid=t("in_grid"), vt=t("in_vt_dxdy_img"), grad_out=t("in_grad_out"),
        max_aniso=int(z["in_max_aniso"]), mode=int(z["in_mode"]), padding=int(z["in_padding"]), out=t("out_out"),
        grad_tex=[t(f"out_grad_tex{i}") for i in range(L)], grad_grid=t("out_grad_grid"))


def mipmap_inputs(seed, N, C, size, levels, H, W, dtype=th.float32, jscale=0.05):
    """Seeded synthetic inputs of the same family as the fixtures (for oracle-vs-HIP comparisons in
    the modes the reference model cannot run)."""
    g = th.Generator().manual_seed(seed)
    tex = [th.rand(N, C, size, size, generator=g, dtype=th.float64).to(dtype)]
    for _ in range(levels - 1):
        tex.append(th.nn.functional.avg_pool2d(tex[-1], 2))
    grid = (th.rand(N, H, W, 2, generator=g, dtype=th.float64) * 2.4 - 1.2).to(dtype)
    jac = th.randn(N, H, W, 2, 2, generator=g, dtype=th.float64) * jscale
    jac[..., 0, :] *= th.rand(N, H, W, 1, generator=g, dtype=th.float64) * 4 + 0.05
    gout = (th.rand(N, C, H, W, generator=g, dtype=th.float64) * 2 - 1).to(dtype)
    return tex, grid, jac.to(dtype), gout


@pytest.fixture(scope="session")
def oracle_ops():
    from backends import OracleBackend, make_ops

    return make_ops(OracleBackend(nthreads=1))


def has_gpu():
    return th.cuda.is_available()

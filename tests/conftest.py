import os
import sys

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


@pytest.fixture(scope="session")
def oracle_ops():
    from backends import OracleBackend, make_ops

    return make_ops(OracleBackend(nthreads=1))


def has_gpu():
    return th.cuda.is_available()

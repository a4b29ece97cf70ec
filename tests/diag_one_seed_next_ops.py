"""Diagnostic (run from the repository root on a GPU box): one seed of tests/fuzz_next_ops.py against a given build of the
library, and how ill-conditioned its screen_space_uv_derivative is (the oracle against itself on inputs one ulp apart).
usage: python tests/diag_one_seed_next_ops.py SEED [product | path/to/variant.so]"""
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, "oracle"); sys.path.insert(0, ".")
from drtk_amd import capi
if len(sys.argv) > 2 and sys.argv[2] != "product":
    capi.use_profiling_library(os.path.abspath(sys.argv[2]))
import torch as th
import fuzz_next_ops as F
import oracle as O
seed = int(sys.argv[1])
c = F.make_case(seed)
try:
    F.run_case(c)
    print("seed", seed, sys.argv[2:], "PASS")
except Exception as e:
    print("seed", seed, sys.argv[2:], "FAIL", str(e)[:200])
# how ill-conditioned is it: the oracle in double against the oracle with inputs perturbed by 1 ulp
d = lambda x: x.to("cuda:0")
from drtk_amd.transform import transform
out = transform(d(c["vN"]), *(d(t) for t in c["cams"]))
_, index = O.rasterize(out.cpu(), c["vi"], c["H"], c["W"])
_, bary = O.render(out.cpu(), c["vi"], index)
mask = (index != -1) & c["mask_keep"]
want = O.screen_space_uv_derivative(c["vN"], c["vt"], c["vi"], c["vi"], index, bary, mask, c["cams"][0], c["cams"][1], c["cams"][2])
got = capi.screen_space_uv_derivative(d(c["vN"]), d(c["vt"]), d(c["vi"]), d(c["vi"]), d(index), d(bary), d(mask), d(c["cams"][0]), d(c["cams"][1]), d(c["cams"][2])).cpu()
err = (got - want).abs().amax((-1, -2))
k = int(err.flatten().argmax())
print("worst pixel error", float(err.flatten()[k]), "value there", float(want.abs().amax((-1,-2)).flatten()[k]), "max |want|", float(want.abs().max()), "median |want|", float(want.abs().amax((-1,-2))[mask].median()))
vt2 = c["vt"] * (1 + 2.2e-16)
want2 = O.screen_space_uv_derivative(c["vN"], vt2, c["vi"], c["vi"], index, bary, mask, c["cams"][0], c["cams"][1], c["cams"][2])
print("oracle vs oracle with vt scaled by 1 ulp: max diff", float((want2 - want).abs().max()))

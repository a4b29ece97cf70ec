#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- tests/golden/fastmath_owner_changes_{100k,250k}.npz: where the reference's rasterizer
built with ITS OWN flags (`-O3 --fast-math`, setup.py:23-24 -> oracle/_ref/libdrtk_ref_fast.so) and the same source
built strict-IEEE (libdrtk_ref_strict.so -- the build the oracle and every fixture follow) disagree about the owner of
a pixel, on one full-resolution benchmark view (drtk_amd.synthetic.sphere_views(1, ...), 2048^2).

Stored: the projected vertices of the view (the topology is drtk_amd.synthetic.uv_sphere's, integers), the pixels (flat indices) where index_img differs, both builds' index and depth there, the number of covered
pixels, the largest relative depth difference over the covered pixels, and a SHA-256 of either build's full index_img --
so that the depth-ordering policy ("source-order IEEE; the shipped flags move a handful of exact near-ties") is pinned
by DATA: a GPU test can check that the HIP image equals the strict image everywhere (hash), and becomes the fast image
when exactly the listed pixels are replaced (hash), without either reference library on the box.

    python oracle/gen_golden_fastmath.py        # needs /root/reference (oracle/_ref is built from it)"""
import hashlib
import os
import sys

import numpy as np
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.numpy()).tobytes()).hexdigest()


def main():
    from backends import RefBackend

    from drtk_amd import synthetic as S  # (pure PyTorch helpers; the package needs its libraries only to import)

    strict, fast = RefBackend("strict"), RefBackend("fast")
    for mesh, res in (("100k", 2048), ("250k", 2048)):
        nl, no = S.MESH_SIZES[mesh]
        v, vi = S.sphere_views(1, nl, no, res, res, lobes=0.05)
        d_s, i_s = strict.rasterize(v, vi, res, res)
        d_f, i_f = fast.rasterize(v, vi, res, res)
        assert th.equal(i_s >= 0, i_f >= 0), "coverage is exact arithmetic on both sides"
        cov = i_s >= 0
        rel = ((d_s.double() - d_f.double()).abs() / d_f.double().clamp(min=1e-30))[cov]
        px = (i_s != i_f).flatten().nonzero().flatten()
        arrs = {
            "pixels": px.numpy().astype(np.int64), "index_strict": i_s.flatten()[px].numpy(), "index_fast": i_f.flatten()[px].numpy(),
            "depth_strict": d_s.flatten()[px].numpy(), "depth_fast": d_f.flatten()[px].numpy(),
            "covered": np.array(int(cov.sum())), "max_rel_depth_difference": np.array(float(rel.max())),
            "depth_lsb_moved": np.array(int((d_s != d_f).sum())),
            "sha256_index_strict": np.array(sha(i_s)), "sha256_index_fast": np.array(sha(i_f)), "sha256_depth_strict": np.array(sha(d_s)),
            "res": np.array(res),
            # the projected vertices themselves: torch's float32 sin / cos differ in the last bit between CPU models (the
            # GPU box is not this container), and a last-bit change of a vertex changes which pixels are near-ties
            "v": v[0].numpy(),
        }
        path = os.path.join(ROOT, "tests", "golden", f"fastmath_owner_changes_{mesh}.npz")
        np.savez_compressed(path, **arrs)
        print(f"{path}: {px.numel()} of {int(cov.sum())} covered pixels change owner; depth differs at {int((d_s != d_f).sum())} pixels, "
              f"largest relative difference {float(rel.max()):.2e}")


if __name__ == "__main__":
    main()

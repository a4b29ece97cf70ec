#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/mipmap_*.npz from the REFERENCE's own
pure-PyTorch model of the anisotropic mipmap sampler, `drtk.mipmap_grid_sample_ref`
(drtk/mipmap_grid_sample.py:130-243), imported from /root/reference (build container only).

The native kernel is CUDA-only in the reference, so this model is the only executable reference of
the op here; the reference documents it as equal to the kernel for force_max_aniso=True,
clip_grad=False (and high_quality=False on the model side).  Each fixture holds the inputs, the
model's output and its autograd gradients wrt every mip level and the grid.

    python oracle/gen_golden_mipmap.py      # rewrites tests/golden/mipmap_*.npz
"""
import builtins
import os
import sys
import types

import numpy as np
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"


def import_reference_model():
    # the reference tree is read-only: no bytecode; its missing native extensions are tolerated the
    # way its documentation build tolerates them (drtk/utils/load_torch_ops.py:22-26, filter2d.py:30-44)
    sys.dont_write_bytecode = True
    builtins.__sphinx_build__ = True
    sys.modules.setdefault("sphinx", types.ModuleType("sphinx"))
    sys.path.insert(0, REF)
    from drtk.mipmap_grid_sample import mipmap_grid_sample_ref  # noqa: E402

    sys.path.remove(REF)
    return mipmap_grid_sample_ref


def pyramid(gen, N, C, size, levels, dtype):
    out = [th.rand(N, C, size, size, generator=gen, dtype=th.float64).to(dtype)]
    for _ in range(levels - 1):
        out.append(th.nn.functional.avg_pool2d(out[-1], 2))
    return out


CASES = {
    # name: (N, C, tex size, levels, H, W, max_aniso, mode, padding, dtype, jacobian scale)
    "bilinear_border_a4": (2, 3, 64, 4, 24, 20, 4, "bilinear", "border", th.float32, 0.05),
    "bilinear_zeros_a1": (1, 2, 32, 3, 16, 16, 1, "bilinear", "zeros", th.float32, 0.08),
    "bilinear_reflection_a2": (1, 3, 32, 2, 16, 12, 2, "bilinear", "reflection", th.float32, 0.1),
    "bicubic_border_a3": (1, 2, 32, 3, 12, 16, 3, "bicubic", "border", th.float32, 0.05),
    "bicubic_zeros_a2_f64": (1, 2, 16, 2, 10, 10, 2, "bicubic", "zeros", th.float64, 0.1),
    "bilinear_border_a8_f64": (1, 3, 64, 5, 12, 12, 8, "bilinear", "border", th.float64, 0.03),
    "single_level_a2": (1, 3, 32, 1, 12, 12, 2, "bilinear", "border", th.float32, 0.05),
}


def main():
    if not os.path.isdir(REF):
        raise SystemExit("needs /root/reference (build container only)")
    model = import_reference_model()
    th.set_num_threads(1)
    os.makedirs(OUT, exist_ok=True)
    for name, (N, C, size, levels, H, W, aniso, mode, padding, dtype, jscale) in CASES.items():
        gen = th.Generator().manual_seed(abs(hash(name)) % (2**31) if False else sum(map(ord, name)))
        tex = [t.requires_grad_(True) for t in pyramid(gen, N, C, size, levels, dtype)]
        # uv mostly inside [-1,1] with some samples outside (padding modes); smooth-ish jacobians of
        # mixed anisotropy, including pixels whose footprint exceeds the coarsest level
        grid = ((th.rand(N, H, W, 2, generator=gen, dtype=th.float64) * 2.4 - 1.2)).to(dtype).requires_grad_(True)
        jac = (th.randn(N, H, W, 2, 2, generator=gen, dtype=th.float64) * jscale)
        jac[..., 0, :] *= th.rand(N, H, W, 1, generator=gen, dtype=th.float64) * 4 + 0.05
        jac = jac.to(dtype)
        out = model(tex, grid, jac, aniso, mode=mode, padding_mode=padding, align_corners=False)
        gout = (th.rand(out.shape, generator=gen, dtype=th.float64) * 2 - 1).to(dtype)
        grads = th.autograd.grad(out, tex + [grid], gout)
        arrs = {"in_grid": grid.detach().numpy(), "in_vt_dxdy_img": jac.numpy(), "in_grad_out": gout.numpy(),
                "in_max_aniso": np.asarray(aniso), "in_mode": np.asarray(0 if mode == "bilinear" else 2),
                "in_padding": np.asarray({"zeros": 0, "border": 1, "reflection": 2}[padding]),
                "in_levels": np.asarray(levels), "out_out": out.detach().numpy(), "out_grad_grid": grads[-1].numpy()}
        for i, t in enumerate(tex):
            arrs[f"in_tex{i}"] = t.detach().numpy()
            arrs[f"out_grad_tex{i}"] = grads[i].numpy()
        path = os.path.join(OUT, f"mipmap_{name}.npz")
        np.savez_compressed(path, **arrs)
        print(f"  {path}: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()

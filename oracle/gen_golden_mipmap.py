#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/mipmap_*.npz from the REFERENCE's own
pure-PyTorch model of the anisotropic mipmap sampler, `drtk.mipmap_grid_sample_ref`
(drtk/mipmap_grid_sample.py:130-243), imported from /root/reference (build container only).

The native kernel is CUDA-only in the reference, so this model is the only executable reference of
the op here; the reference documents it as equal to the kernel for force_max_aniso=True,
clip_grad=False (and high_quality=False on the model side).  Each fixture holds the inputs, the
model's output and its autograd gradients wrt every mip level and the grid.

    python oracle/gen_golden_mipmap.py      # rewrites tests/golden/mipmap_*.npz
    python oracle/gen_golden_mipmap.py --adaptive  # rewrites tests/golden/mipmap_adaptive_*.npz (force_max_aniso=False)
    python oracle/gen_golden_mipmap.py --uv # rewrites tests/golden/uv_derivative_*.npz (own process: it binds
                                            # interpolate_ext::interpolate to the reference's CPU kernel)
"""
import builtins
import os
import sys
import types

import numpy as np
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
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


# ---- the ADAPTIVE tap count (force_max_aniso=False), pinned with data the reference's own model produced ------------
# mipmap_grid_sample_ref always takes `max_aniso` taps per pixel, but it selects the mip level from
# N = min(ceil(p_max / p_min), max_aniso) (drtk/mipmap_grid_sample.py:252-258) -- exactly the N the CUDA kernel also uses as
# its tap COUNT when force_max_aniso is false (mipmap_grid_sampler_kernel.cu:459-462, :496-499).  So on a pixel whose
# N is k, the kernel's adaptive result with max_aniso = M equals the model called with max_aniso = k: same N, same
# lambda, the same k taps at (j + 1) / (k + 1) * 2 - 1.  The fixture is assembled pixel class by pixel class from M runs
# of the model (k = 1 .. M), its gradients by linearity from the upstream gradient masked to each class.
# PREDICATE (stated here, enforced below): no pixel's ratio p_max / p_min lies within ADAPTIVE_MARGIN of an integer below
# M, so that ceil() is the same whatever the rounding of the footprint lengths (the kernel adds 1e-12 under the root and
# evaluates in its own order); ratios at or beyond M need no margin (N = M on either side).  Pixels are re-drawn until the
# predicate holds.  clip_grad stays False (the model has no such mode).
ADAPTIVE_MARGIN = 0.05
ADAPTIVE_CASES = {
    # name: (N, C, tex size, levels, H, W, max_aniso, mode, padding, dtype, jacobian scale)
    "adaptive_bilinear_border_a4": (2, 3, 64, 4, 24, 20, 4, "bilinear", "border", th.float32, 0.05),
    "adaptive_bilinear_zeros_a8": (1, 2, 64, 5, 20, 24, 8, "bilinear", "zeros", th.float32, 0.04),
    "adaptive_bicubic_border_a3": (1, 2, 32, 3, 12, 16, 3, "bicubic", "border", th.float32, 0.05),
    "adaptive_bilinear_reflection_a6_f64": (1, 3, 32, 3, 16, 12, 6, "bilinear", "reflection", th.float64, 0.08),
    "adaptive_bicubic_zeros_a5_f64": (1, 2, 32, 4, 10, 14, 5, "bicubic", "zeros", th.float64, 0.06),
}


def tap_class(jac, size, max_aniso):
    """N = min(ceil(p_max / p_min), max_aniso) per pixel and the distance of the ratio from the nearest integer where
    that matters (ratios below max_aniso), in double from the fixture's (dtype-rounded) Jacobian."""
    j = jac.double() * th.tensor([size, size], dtype=th.float64)
    px, py = j[..., 0, :].norm(dim=-1), j[..., 1, :].norm(dim=-1)
    r = th.maximum(px, py) / th.minimum(px, py)
    n = th.clamp(th.ceil(r), max=max_aniso)
    margin = th.where(r < max_aniso, (r - th.round(r)).abs(), th.full_like(r, 1.0))
    return n.long(), margin


def main_adaptive():
    if not os.path.isdir(REF):
        raise SystemExit("needs /root/reference (build container only)")
    model = import_reference_model()
    th.set_num_threads(1)
    for name, (N, C, size, levels, H, W, M, mode, padding, dtype, jscale) in ADAPTIVE_CASES.items():
        gen = th.Generator().manual_seed(sum(map(ord, name)))
        tex = [t.requires_grad_(True) for t in pyramid(gen, N, C, size, levels, dtype)]
        grid = ((th.rand(N, H, W, 2, generator=gen, dtype=th.float64) * 2.4 - 1.2)).to(dtype).requires_grad_(True)

        def draw():
            j = th.randn(N, H, W, 2, 2, generator=gen, dtype=th.float64) * jscale
            j[..., 0, :] *= th.rand(N, H, W, 1, generator=gen, dtype=th.float64) * (M + 1) + 0.05
            swap = th.rand(N, H, W, generator=gen) < 0.5  # either axis may be the major one
            j = th.where(swap[..., None, None], j.flip(-2), j)
            return j.to(dtype)

        jac = draw()
        for _ in range(200):
            _, margin = tap_class(jac, size, M)
            bad = margin < ADAPTIVE_MARGIN
            if not bad.any():
                break
            jac = th.where(bad[..., None, None], draw(), jac)
        cls, margin = tap_class(jac, size, M)
        assert float(margin.min()) >= ADAPTIVE_MARGIN, name
        gout = (th.rand(N, C, H, W, generator=gen, dtype=th.float64) * 2 - 1).to(dtype)
        out = th.zeros(N, C, H, W, dtype=dtype)
        grads = [th.zeros_like(t) for t in tex] + [th.zeros_like(grid)]
        for k in range(1, M + 1):
            m = (cls == k)
            if not m.any():
                continue
            o_k = model(tex, grid, jac, k, mode=mode, padding_mode=padding, align_corners=False)
            out = th.where(m[:, None], o_k.detach(), out)
            g_k = th.autograd.grad(o_k, tex + [grid], gout * m[:, None].to(dtype))
            grads = [a + b for a, b in zip(grads, g_k)]
        hist = th.bincount(cls.reshape(-1), minlength=M + 1)[1:].tolist()
        arrs = {"in_grid": grid.detach().numpy(), "in_vt_dxdy_img": jac.numpy(), "in_grad_out": gout.numpy(),
                "in_max_aniso": np.asarray(M), "in_mode": np.asarray(0 if mode == "bilinear" else 2),
                "in_padding": np.asarray({"zeros": 0, "border": 1, "reflection": 2}[padding]),
                "in_levels": np.asarray(levels), "out_out": out.numpy(), "out_grad_grid": grads[-1].numpy(),
                "info_tap_class": cls.numpy().astype(np.int8), "info_min_margin": np.asarray(float(margin.min()))}
        for i, t in enumerate(tex):
            arrs[f"in_tex{i}"] = t.detach().numpy()
            arrs[f"out_grad_tex{i}"] = grads[i].numpy()
        path = os.path.join(OUT, f"mipmap_{name}.npz")
        np.savez_compressed(path, **arrs)
        print(f"  {path}: {os.path.getsize(path) / 1024:.1f} KiB, pixels per tap count 1..{M}: {hist}, min margin {float(margin.min()):.3f}")


if __name__ == "__main__" and "--adaptive" in sys.argv:
    main_adaptive()
elif __name__ == "__main__" and "--uv" not in sys.argv:
    main()


# ---- screen_space_uv_derivative: the reference's own PyTorch composite, run on CPU ---------------------
def gen_uv_derivative():
    """drtk.screen_space_uv_derivative (drtk/screen_space_uv_derivative.py:15-80) calls
    th.ops.interpolate_ext.interpolate; in this process that op is bound to the reference's own CPU kernel
    (oracle/_ref, built from /root/reference/src/interpolate) so the whole composite is the reference's code."""
    import ref_build

    ref_build.build(verbose=False)
    rs = ref_build.load("strict")
    lib = th.library.Library("interpolate_ext", "DEF")
    lib.define("interpolate(Tensor vert_attributes, Tensor vi, Tensor index_img, Tensor bary_img) -> Tensor")
    lib.impl("interpolate", lambda a, vi, idx, bary: rs.interpolate(a.contiguous(), vi.contiguous(), idx, bary), "CPU")
    import_reference_model()
    sys.path.insert(0, REF)
    from drtk.screen_space_uv_derivative import screen_space_uv_derivative  # noqa: E402

    sys.path.remove(REF)
    for name, dtype in (("f32", th.float32), ("f64", th.float64)):
        g = th.Generator().manual_seed(5)
        n, gsz, H, W = 2, 7, 40, 48
        # a perturbed grid mesh in front of two cameras, separate uv topology (vti != vi)
        ys, xs = th.meshgrid(th.linspace(-1, 1, gsz, dtype=th.float64), th.linspace(-1, 1, gsz, dtype=th.float64), indexing="ij")
        v = th.stack([xs, ys, th.zeros_like(xs)], -1).reshape(-1, 3)
        v = v + th.randn(v.shape, generator=g, dtype=th.float64) * th.tensor([0.04, 0.04, 0.15], dtype=th.float64)
        quads = [(r * gsz + c, r * gsz + c + 1, (r + 1) * gsz + c, (r + 1) * gsz + c + 1) for r in range(gsz - 1) for c in range(gsz - 1)]
        vi = th.tensor([t for a, b, c2, d in quads for t in ((a, b, c2), (b, d, c2))], dtype=th.int32)
        vti = th.arange(vi.numel(), dtype=th.int32).view(-1, 3)  # every corner has its own uv
        uv_v = (v[:, :2] * 0.4 + 0.5) + th.randn(v.shape[0], 2, generator=g, dtype=th.float64) * 0.02
        vt = uv_v[vi.long().reshape(-1)] + th.randn(vi.numel(), 2, generator=g, dtype=th.float64) * 0.003
        ang = th.tensor([0.25, -0.4], dtype=th.float64)
        camrot = th.stack([th.tensor([[th.cos(a), 0, th.sin(a)], [0, 1, 0], [-th.sin(a), 0, th.cos(a)]], dtype=th.float64) for a in ang])
        campos = -(camrot.transpose(1, 2) @ th.tensor([0.0, 0.0, 3.0], dtype=th.float64))
        focal = th.stack([th.eye(2, dtype=th.float64) * 1.1 * W, th.tensor([[1.2 * W, 3.0], [0.0, 1.0 * W]], dtype=th.float64)])
        princpt = th.tensor([[W / 2, H / 2], [W / 2 + 2, H / 2 - 1]], dtype=th.float64)
        v, vt, camrot, campos, focal, princpt = (t.to(dtype) for t in (v, vt, camrot, campos, focal, princpt))
        vN, vtN = v[None].expand(n, -1, -1).contiguous(), vt[None].expand(n, -1, -1).contiguous()
        v_cam = (camrot[:, None] @ (vN - campos[:, None])[..., None])[..., 0]
        v_pix = th.cat([(focal[:, None] @ (v_cam[..., :2] / v_cam[..., 2:3])[..., None])[..., 0] + princpt[:, None], v_cam[..., 2:3]], -1).contiguous()
        vib = vi[None].expand(n, -1, -1).contiguous()
        _, index = rs.rasterize(v_pix, vib, H, W)
        _, bary = rs.render(v_pix, vib, index)
        mask = index != -1
        out = screen_space_uv_derivative(vN, vtN, vi, vti, index, bary, mask, campos, camrot, focal)
        arrs = {"in_v": vN.numpy(), "in_vt": vtN.numpy(), "in_vi": vi.numpy(), "in_vti": vti.numpy(), "in_index_img": index.numpy(),
                "in_bary_img": bary.numpy(), "in_campos": campos.numpy(), "in_camrot": camrot.numpy(), "in_focal": focal.numpy(),
                "out_vt_dxdy_img": out.numpy()}
        path = os.path.join(OUT, f"uv_derivative_{name}.npz")
        np.savez_compressed(path, **arrs)
        print(f"  {path}: {os.path.getsize(path) / 1024:.1f} KiB, covered {int(mask.sum())} px, max |J| {float(out.abs().max()):.3f}")


if __name__ == "__main__" and "--uv" in sys.argv:
    gen_uv_derivative()

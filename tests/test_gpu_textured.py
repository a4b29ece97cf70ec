"""GPU (-m gpu): BASELINE.json configs[4] -- "1M-tri scene, 4096x4096, edge_grad + mipmap_grid_sampler textured
shading, fp16 attributes" -- as ONE PIPELINE on one full-size view against the CPU oracle (the four hot-path ops +
the restatements of the mipmap sampler and of the uv-Jacobian composite), forward and backward; then the same
pipeline with the uv attributes and the texture pyramid STORED in fp16 under autocast against the f32 run.

    transform -> rasterize -> render -> interpolate(uv; own atlas topology vti) -> screen_space_uv_derivative ->
    mipmap_grid_sample(max_aniso 8, border) -> mask -> edge_grad_estimator -> sum(img^2) + sum(depth) -> backward

Stage by stage, every stage fed with identical inputs on both sides (so that a discrete decision upstream -- a pixel's
owner, a tap count -- cannot turn into an O(1) difference downstream and hide or fake an error):
  * index_img bit-exact; barycentrics and the uv image bit-identical (same operations, same order);
  * uv Jacobian: kernel (closed form) vs the reference's composite as restated by the oracle -- equal to rounding
    except where a triangle is seen edge-on and the 2x2 inverse is ill-conditioned in f32 (profiles/NOTES.md 4): a robust bar
    (median and 99 % quantile of the relative difference, share of pixels beyond 1e-3);
  * sampler forward on the kernel's Jacobian: 1e-5; whole-pipeline loss; gradients wrt the projected vertices, the uv
    attributes and EVERY mip level: 1e-5 + 1e-5 * max|ref|.
"""
import time

import pytest
import torch as th

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def close(a, ref, what, atol=1e-5, rtol=1e-5):
    a = a.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert a.shape == ref.shape, (what, a.shape, ref.shape)
    tol = atol + rtol * float(ref.abs().max())
    err = float((a - ref).abs().max())
    assert err <= tol, f"{what}: max abs err {err:.3e} > tol {tol:.3e}"
    return err, tol


def _oracle_mipmap():
    import oracle as O

    class Mip(th.autograd.Function):
        @staticmethod
        def forward(ctx, grid, jac, max_aniso, *levels):
            ctx.save_for_backward(grid, jac, *levels)
            ctx.max_aniso = max_aniso
            return O.mipmap_grid_sampler_2d([t.detach() for t in levels], grid.detach(), jac, max_aniso, 1, 0)

        @staticmethod
        def backward(ctx, go):
            grid, jac, *levels = ctx.saved_tensors
            gl, gg = O.mipmap_grid_sampler_2d_backward(go.contiguous(), [t.detach() for t in levels], grid.detach(), jac, ctx.max_aniso, 1, 0)
            return (gg, None, None, *gl)

    def sample(tex, grid, jac, max_aniso, padding_mode="border"):
        assert padding_mode == "border"
        return Mip.apply(grid, jac, max_aniso, *tex)

    return sample


def _scene(mesh, res, tex_size, view):
    from drtk_amd import synthetic as S

    nl, no = S.MESH_SIZES[mesh]
    v_world, vi = S.uv_sphere(nl, no, lobes=0.05, device=DEV)
    vt, vti = S.uv_sphere_atlas(nl, no, device=DEV)
    cams = S.ring_cameras(8, res, res, device=DEV)
    campos, camrot, focal, princpt = (c[view:view + 1].contiguous() for c in cams)
    # values that fp16 represents exactly, so that the fp16-stored run and the f32 run (and the oracle) see the same numbers
    vt16 = vt[None].half()
    tex16 = [t.half() for t in S.texture_pyramid(1, 3, tex_size, device=DEV)]
    return v_world, vi, vt16, vti, tex16, (campos, camrot, focal, princpt)


def _run_hip(v_world, vi, vt, vti, tex, cams, res, autocast):
    import drtk_amd
    from drtk_amd import synthetic as S

    campos, camrot, focal, princpt = cams
    v_leaf = v_world[None].clone().requires_grad_(True)
    vt_leaf = vt.clone().requires_grad_(True)
    tex_leaf = [t.clone().requires_grad_(True) for t in tex]
    with th.autocast("cuda", dtype=th.float16, enabled=autocast):
        v_pix = drtk_amd.transform(v_leaf, campos, camrot, focal, princpt)
        v_pix.retain_grad()
        out = S.textured_shading(drtk_amd, v_leaf, v_pix, vi, vt_leaf, vti, tex_leaf, campos, camrot, focal, res, res)
        # sums, not means: O(1) upstream gradients (2 * img and 1 per pixel), so that the 1e-5 bars below bite
        loss = out["img"].square().sum() + out["depth_img"].sum()
    loss.backward()
    return dict(out, loss=loss.detach(), v_pix=v_pix.detach(), g_v_pix=v_pix.grad, g_v_world=v_leaf.grad, g_vt=vt_leaf.grad,
                g_tex=[t.grad for t in tex_leaf])


@pytest.mark.parametrize("mesh,res,tex_size,view", [("10k", 512, 512, 1), ("1M", 4096, 4096, 3)])
def test_textured_pipeline_one_full_size_view_matches_oracle(mesh, res, tex_size, view):
    import oracle as O
    from backends import OracleBackend, make_ops
    from drtk_amd import synthetic as S

    v_world, vi, vt16, vti, tex16, cams = _scene(mesh, res, tex_size, view)
    campos, camrot, focal, princpt = cams
    t0 = time.time()
    hip = _run_hip(v_world, vi, vt16.float(), vti, [t.float() for t in tex16], cams, res, autocast=False)
    th.cuda.synchronize()
    covered = int((hip["index_img"] != -1).sum())
    assert covered > 0.4 * res * res

    # ---- oracle pipeline on the same projected vertices, uv attributes, texture and (kernel-computed) uv Jacobian
    cpu = lambda t: t.detach().cpu()  # noqa: E731
    ops = make_ops(OracleBackend(nthreads=0))
    v_pix_o = cpu(hip["v_pix"]).requires_grad_(True)
    vt_o = cpu(vt16.float()).requires_grad_(True)
    tex_o = [cpu(t.float()).requires_grad_(True) for t in tex16]
    jac_hip = cpu(hip["vt_dxdy_img"])
    ref = S.textured_shading(ops, None, v_pix_o, cpu(vi), vt_o, cpu(vti), tex_o, None, None, None, res, res,
                             uv_jacobian=lambda *a: jac_hip, mipmap=_oracle_mipmap())
    loss_o = ref["img"].square().sum() + ref["depth_img"].sum()
    loss_o.backward()
    t_oracle = time.time() - t0

    assert th.equal(cpu(hip["index_img"]), ref["index_img"]), f"{int((cpu(hip['index_img']) != ref['index_img']).sum())} index px differ"
    assert th.equal(cpu(hip["bary_img"]), ref["bary_img"].detach()), "barycentrics are not bit-identical"
    assert th.equal(cpu(hip["uv_img"]), ref["uv_img"].detach()), "uv image is not bit-identical"
    close(hip["depth_img"], ref["depth_img"], "depth")
    e_img, _ = close(hip["shaded"], ref["shaded"], "mipmap_grid_sample forward")
    close(hip["img"], ref["img"], "shaded + masked image")
    assert abs(float(hip["loss"]) - float(loss_o.detach())) <= 2e-6 * abs(float(loss_o.detach()))  # f32 sums of 1e7 terms
    e_v, tol_v = close(hip["g_v_pix"], v_pix_o.grad, "d loss / d v_pix (render + uv-interpolate + edge_grad routes)")
    e_t, tol_t = close(hip["g_vt"], vt_o.grad, "d loss / d vt (uv attributes)")
    for lvl, (a, b) in enumerate(zip(hip["g_tex"], tex_o)):
        close(a, b.grad, f"d loss / d texture level {lvl}")
    assert all(float(b.grad.abs().max()) > 0 for b in tex_o[:2]), "the finest levels must receive gradient"

    # ---- uv Jacobian: closed-form kernel vs the reference's composite (restated), robustly
    jac_o = O.screen_space_uv_derivative(cpu(v_world)[None], cpu(vt16.float()), cpu(vi), cpu(vti), ref["index_img"], ref["bary_img"].detach(),
                                         ref["index_img"] != -1, cpu(campos), cpu(camrot), cpu(focal))
    m = (ref["index_img"] != -1)
    a, b = jac_hip[m].double().reshape(-1, 4), jac_o[m].double().reshape(-1, 4)
    scale = b.abs().amax(1, keepdim=True).clamp(min=1e-12)
    rel = ((a - b).abs() / scale).amax(1)
    rel = rel[th.isfinite(rel)]
    sub = rel[:: max(1, rel.numel() // 4_000_000)]  # torch.quantile takes at most 2^24 elements
    q50, q99 = float(sub.quantile(0.5)), float(sub.quantile(0.99))
    share = float((rel > 1e-3).double().mean())
    assert q50 < 2e-6 and q99 < 2e-4 and share < 2e-3, (q50, q99, share)
    assert bool((jac_hip[~m] == 0).all())
    print(f"[{mesh}@{res}, texture {tex_size}] covered {covered} px; sampler fwd err {e_img:.2e}; grad v_pix err {e_v:.2e} (bar {tol_v:.2e}); "
          f"grad vt err {e_t:.2e} (bar {tol_t:.2e}); uv-Jacobian rel diff median {q50:.1e}, 99 % {q99:.1e}, beyond 1e-3: {share:.1e}; "
          f"oracle side {t_oracle:.0f} s")


@pytest.mark.parametrize("mesh,res,tex_size,view", [("1M", 4096, 4096, 3)])
def test_textured_pipeline_fp16_stored_attributes_under_autocast_at_full_size(mesh, res, tex_size, view):
    """"fp16 attributes": uv attributes and texture pyramid STORED as fp16 leaves, the step run under autocast.  Like
    the reference (autocast wrappers cast to float32 at every op: interpolate_module.cpp:584-600, mipmap_grid_sampler_
    module.cpp:214-249) every op computes in f32: all forward tensors equal the f32-leaf run bit for bit, and the fp16
    leaves receive the f32 gradients rounded to fp16."""
    v_world, vi, vt16, vti, tex16, cams = _scene(mesh, res, tex_size, view)
    h = _run_hip(v_world, vi, vt16, vti, tex16, cams, res, autocast=True)
    f = _run_hip(v_world, vi, vt16.float(), vti, [t.float() for t in tex16], cams, res, autocast=False)
    for k in ("index_img", "bary_img", "uv_img", "vt_dxdy_img", "shaded", "img"):
        assert h[k].dtype == f[k].dtype and th.equal(h[k], f[k]), k
    assert float(h["loss"]) == float(f["loss"])
    assert h["g_vt"].dtype == th.float16 and all(g.dtype == th.float16 for g in h["g_tex"])
    # the texture / uv gradients are sums of float atomics (order varies run to run by ~1e-7 relative) rounded to fp16
    # (2^-11 relative): equal to the rounded f32 gradient up to one fp16 ulp of the value
    for name, a, b in [("vt", h["g_vt"], f["g_vt"])] + [(f"texture level {i}", a, b) for i, (a, b) in enumerate(zip(h["g_tex"], f["g_tex"]))]:
        a, b = a.float(), b.float()
        tol = b.abs() * 2.0 ** -10 + 6e-8 + 1e-6 * float(b.abs().max())  # 6e-8: fp16 subnormal spacing
        assert bool(((a - b).abs() <= tol).all()), name
    close(h["g_v_world"], f["g_v_world"], "vertex gradient", atol=1e-7, rtol=1e-5)

#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/refpy_*.npz from the REFERENCE'S OWN STACK, top to
bottom: its Python package (/root/reference/drtk/*.py, imported from where it lies, build container only)
on top of its own torch-op modules and CPU kernels (oracle/_ref/<x>_ext.so = src/<x>/<x>_module.cpp +
<x>_kernel_cpu.cpp, built by oracle/ref_build.build_modules()).  Nothing between `drtk.rasterize(...)` and
the CPU kernel is ours here: the `vi` broadcast, `bary_img.detach()`, the hook registration
(drtk/edge_grad_estimator.py:20-180), what each C++ autograd Function saves, which `requires_grad` it
consults and which gradient slots it fills (rasterize_module.cpp:31-71, render_module.cpp:27-72,
interpolate_module.cpp:378-433, edge_grad_module.cpp:30-170) are the reference's code.

It is a process of its own: the reference's modules register the real namespaces (rasterize_ext, ...) and
cannot be loaded next to drtk_amd's shim, so nothing of drtk_amd is imported (scene inputs are read from
the fixtures gen_golden.py wrote).  Single torch thread => deterministic accumulation order.

    python oracle/gen_golden_refpy.py      # rewrites tests/golden/refpy_*.npz

Fixtures:
  refpy_transform_{f32,f64}   drtk.transform (drtk/transform.py:13-119, utils/projection.py:33-53,486-540), pinhole:
                              (campos, camrot, focal, princpt) and (K, Rt 3x4 / 4x4) forms, shared [1,V,3] and per-view
                              vertices, skewed focal, vertices at / behind the z = 0 plane; outputs + VJPs wrt v and
                              the cameras for a stored upstream gradient.
  refpy_step_spheres_f32      the SURVEY 8d step on the spheres scene through drtk.rasterize / render / interpolate /
                              edge_grad_estimator + autograd: no hook; with a v_pix_img_hook that rescales the
                              gradient (and what the hook saw); partial requires_grad patterns.
  refpy_sparse_<scene>        drtk.interpolation_matrix / interpolation_normal_matrix incl. the module's own A^T A pattern builder
  refpy_uv_derivative_autograd  what drtk.screen_space_uv_derivative does under autograd in the reference: backward() raises
                              (in-place mask on the output of linalg.inv_ex); the message is stored.
  refpy_two_triangles         test/two_triangles.py (the reference's only script-level test) at 64x64: iteration-0
                              tensors and the loss at iterations 0, 1, 50, 100, 200 of its Adam loop.
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


def import_reference():
    """`import drtk` from /root/reference on top of oracle/_ref/<x>_ext.so."""
    import ref_build

    assert "drtk_amd" not in sys.modules, "the reference's modules and drtk_amd's shim register the same namespaces"
    ref_build.build_modules(verbose=True)
    for m in ref_build.MODULES:
        th.ops.load_library(ref_build.module_path(m))
    # read-only tree: no bytecode.  The reference resolves `drtk.<x>_ext` as Python modules; the ones that are not
    # built (it has seven extensions, four are on the path) are tolerated the way its documentation build tolerates
    # them (drtk/utils/load_torch_ops.py:22-26, filter2d.py:30-44) -- the four ops are already registered above.
    sys.dont_write_bytecode = True
    builtins.__sphinx_build__ = True
    sys.modules.setdefault("sphinx", types.ModuleType("sphinx"))
    sys.path.insert(0, REF)
    import drtk  # noqa: E402

    sys.path.remove(REF)
    assert drtk.__file__.startswith(REF)
    return drtk


def load_fixture(name):
    z = np.load(os.path.join(OUT, name + ".npz"))
    return {k: (th.from_numpy(z[k]) if z[k].ndim else z[k].item()) for k in z.files}


def save(name, arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().numpy() if isinstance(v, th.Tensor) else np.asarray(v)) for k, v in arrs.items()})
    print(f"  {path}: {os.path.getsize(path) / 1024:.1f} KiB")


# ------------------------------------------------------------------------------------------------ transform
def gen_transform(drtk):
    for tag, dtype in (("f32", th.float32), ("f64", th.float64)):
        g = th.Generator().manual_seed(21)
        N, V = 3, 257
        r = lambda *s: th.rand(*s, generator=g, dtype=th.float64)  # noqa: E731
        v = (r(N, V, 3) * 2 - 1) * th.tensor([1.5, 1.5, 1.0], dtype=th.float64)
        ang = th.tensor([0.0, -0.7, 2.1], dtype=th.float64)
        c, s = th.cos(ang), th.sin(ang)
        z0, o1 = th.zeros(3, dtype=th.float64), th.ones(3, dtype=th.float64)
        ry = th.stack([th.stack([c, z0, s], -1), th.stack([z0, o1, z0], -1), th.stack([-s, z0, c], -1)], 1)
        rx = th.stack([th.stack([o1, z0, z0], -1), th.stack([z0, c, -s], -1), th.stack([z0, s, c], -1)], 1)
        camrot = ry @ rx
        campos = -(camrot.transpose(1, 2) @ th.tensor([0.1, -0.2, 3.0], dtype=th.float64)[None, :, None])[..., 0]
        # view 0: identity rotation, camera at the origin, its vertices 3 units down the z axis -- there v_cam == v in
        # any evaluation order, so vertices exactly on, just in front of and just behind the camera plane (the z clamp
        # of utils/projection.py:48-49) are well-defined inputs instead of rounding noise
        campos[0] = 0
        v[0, :, 2] += 3.0
        v[0, :5] = th.tensor([[0.3, -0.2, 0.0], [0.3, -0.2, 1e-9], [0.3, -0.2, -1e-9], [0.5, 0.4, -2.0], [0.0, 0.0, 1e-3]], dtype=th.float64)
        focal = th.tensor([[[600.0, 0.0], [0.0, 600.0]], [[512.5, 3.25], [0.0, 498.0]], [[300.0, -2.0], [1.5, 310.0]]], dtype=th.float64)
        princpt = th.tensor([[256.0, 256.0], [250.5, 260.25], [128.0, 100.0]], dtype=th.float64)
        v, camrot, campos, focal, princpt = (t.to(dtype) for t in (v, camrot, campos, focal, princpt))
        K = th.zeros(N, 3, 3, dtype=dtype)
        K[:, :2, :2], K[:, :2, 2], K[:, 2, 2] = focal, princpt, 1
        t = -(camrot @ campos[..., None])
        Rt34 = th.cat([camrot, t], dim=2)
        Rt44 = th.cat([Rt34, th.tensor([0, 0, 0, 1], dtype=dtype)[None, None].expand(N, 1, 4)], dim=1)
        gout = (r(N, V, 3) * 2 - 1).to(dtype)
        arrs = dict(in_v=v, in_campos=campos, in_camrot=camrot, in_focal=focal, in_princpt=princpt, in_K=K, in_Rt34=Rt34,
                    in_Rt44=Rt44, in_grad_out=gout)

        def run(vv, **cams):
            leaves = {k: x.clone().requires_grad_(True) for k, x in cams.items()}
            vl = vv.clone().requires_grad_(True)
            out = drtk.transform(vl, **leaves)
            assert out.shape == (N, V, 3)
            out.backward(gout)
            return out.detach(), vl.grad, {k: x.grad for k, x in leaves.items()}

        # per-view vertices, decomposed camera
        out, gv, gc = run(v, campos=campos, camrot=camrot, focal=focal, princpt=princpt)
        arrs.update(out_per_view=out, out_per_view_grad_v=gv, **{f"out_per_view_grad_{k}": x for k, x in gc.items()})
        # shared vertices [1,V,3] (broadcast against the N cameras), decomposed camera
        out, gv, gc = run(v[:1], campos=campos, camrot=camrot, focal=focal, princpt=princpt)
        arrs.update(out_shared=out, out_shared_grad_v=gv)
        # K / Rt forms (3x4 and 4x4 extrinsics)
        out, gv, gc = run(v, K=K, Rt=Rt34)
        arrs.update(out_KRt34=out, out_KRt34_grad_v=gv, out_KRt34_grad_K=gc["K"], out_KRt34_grad_Rt=gc["Rt"])
        out, gv, gc = run(v, K=K, Rt=Rt44)
        arrs.update(out_KRt44=out, out_KRt44_grad_v=gv)
        # mixed: K with (campos, camrot)
        out, gv, gc = run(v, K=K, campos=campos, camrot=camrot)
        arrs.update(out_K_campos=out, out_K_campos_grad_v=gv)
        # explicit "pinhole" spelling of the distortion mode == None (utils/projection.py:561)
        out2 = drtk.transform(v, campos, camrot, focal, princpt, distortion_mode="pinhole", distortion_coeff=th.zeros(N, 4, dtype=dtype))
        assert th.equal(out2, arrs["out_per_view"])
        # the two argument errors (drtk/transform.py:90-94)
        for kw in (dict(campos=campos, camrot=camrot, Rt=Rt34, K=K), dict(Rt=Rt34, K=K, focal=focal, princpt=princpt), dict(K=K), dict(Rt=Rt34)):
            try:
                drtk.transform(v, **kw)
                raise AssertionError("expected ValueError")
            except ValueError:
                pass
        save(f"refpy_transform_{tag}", arrs)


# ------------------------------------------------------------------------------------------------ the step
def step(drtk, v, vi, attr, H, W, hook=None, max_dp_dr=1e4, v_rg=True, a_rg=True):
    """SURVEY 8d step == drtk_amd.synthetic.fwd_bwd_step, spelled with the reference's functions."""
    v = v.clone().requires_grad_(v_rg)
    attr = attr.clone().requires_grad_(a_rg)
    index_img = drtk.rasterize(v, vi, H, W)
    depth_img, bary_img = drtk.render(v, vi, index_img)
    img = drtk.interpolate(attr, vi, index_img, bary_img)
    img = img * (index_img != -1)[:, None]
    img = drtk.edge_grad_estimator(v_pix=v, vi=vi, bary_img=bary_img, img=img, index_img=index_img,
                                   v_pix_img_hook=hook, max_dp_dr=max_dp_dr)
    loss = (img * img).mean() + depth_img.mean()
    loss.backward()
    return loss.detach(), index_img, depth_img.detach(), v.grad, attr.grad


def gen_step(drtk):
    sc = load_fixture("step_spheres_f32")
    v, vi, attr, H, W = sc["in_v"], sc["in_vi"], sc["in_attr"], int(sc["in_H"]), int(sc["in_W"])
    arrs = dict(in_v=v, in_vi=vi, in_attr=attr, in_H=H, in_W=W)
    loss, index, depth, gv, ga = step(drtk, v, vi, attr, H, W)
    arrs.update(out_loss=loss, out_index_img=index, out_v_grad=gv, out_attr_grad=ga)
    # the restated wiring of round 1 (tests/backends.py: make_ops around the reference KERNELS) wrote
    # step_spheres_f32; the reference's own wiring must agree with it bit for bit (same kernels, same order)
    same = th.equal(index, sc["out_index_img"]) and th.equal(gv, sc["out_v_grad"]) and th.equal(ga, sc["out_attr_grad"])
    print(f"  step (no hook): loss {float(loss):.7f}; identical to the restated wiring's fixture: {same}")
    assert same and float(loss) == float(sc["out_loss"])
    # rasterize_with_depth through the reference wrapper
    depth_r, index_r = drtk.rasterize_with_depth(v, vi, H, W)
    assert th.equal(index_r, index)
    arrs.update(out_depth_img=depth_r)
    # hook: sees grad_v_pix_img [N,3,H,W]; its return value replaces the gradient (drtk/edge_grad_estimator.py:178-179)
    seen = {}

    def hook(g):
        seen["g"] = g.clone()
        return g * 0.5

    loss_h, _, _, gv_h, ga_h = step(drtk, v, vi, attr, H, W, hook=hook)
    assert float(loss_h) == float(loss) and th.equal(ga_h, ga)
    arrs.update(out_hook_seen=seen["g"], out_hook_v_grad=gv_h)
    # a hook that returns None leaves the gradient alone
    loss_n, _, _, gv_n, _ = step(drtk, v, vi, attr, H, W, hook=lambda g: None)
    assert th.equal(gv_n, gv)
    # max_dp_dr = 0 (no clamp of dp/dr at intersections) and a small finite clamp
    for name, M in (("noclamp", 0.0), ("clamp10", 10.0)):
        _, _, _, gv_m, _ = step(drtk, v, vi, attr, H, W, max_dp_dr=M)
        arrs[f"out_{name}_v_grad"] = gv_m
    # partial requires_grad: geometry only / attributes only
    _, _, _, gv_g, ga_g = step(drtk, v, vi, attr, H, W, a_rg=False)
    assert ga_g is None
    arrs.update(out_geomonly_v_grad=gv_g)
    _, _, _, gv_a, ga_a = step(drtk, v, vi, attr, H, W, v_rg=False)
    assert gv_a is None
    arrs.update(out_attronly_attr_grad=ga_a)
    # per-view topology [N,F,3] takes the same path as the broadcast [F,3]
    _, idx_b, _, gv_b, _ = step(drtk, v, vi[None].expand(v.shape[0], -1, -1).contiguous(), attr, H, W)
    assert th.equal(idx_b, index) and th.equal(gv_b, gv)
    save("refpy_step_spheres_f32", arrs)


# ------------------------------------------------------------------------------------------------ two triangles
def gen_two_triangles(drtk):
    """test/two_triangles.py:14-92 at 64x64 on CPU: same scene (x, y scaled by 1/8), same shading, same loss, Adam;
    the perturbed start and the learning rate are those of round 1's trajectory fixture (CPU generator; the script's
    `th.cuda.manual_seed(10)` stream does not exist here)."""
    import torch.nn.functional as thf

    old = load_fixture("two_triangles_trajectory")
    vi, vt, tex, v_gt, v0 = old["out_vi"], old["out_vt"], old["out_tex"], old["out_v_gt"], old["out_v0"]
    v = th.nn.Parameter(v0.clone())

    def shade(vv):
        index_img = drtk.rasterize(vv, vi, 64, 64)
        _, bary_img = drtk.render(vv, vi, index_img)
        vt_img = drtk.interpolate(vt, vi, index_img, bary_img).permute(0, 2, 3, 1)
        img = thf.grid_sample(tex, vt_img, padding_mode="border", align_corners=False) * (index_img != -1)[:, None]
        return img, index_img, bary_img

    with th.no_grad():
        img_gt, index_gt, _ = shade(v_gt)
    optim = th.optim.Adam([v], lr=0.05, betas=(0.9, 0.999))
    rec = dict(v_gt=v_gt, v0=v0, vi=vi, vt=vt, tex=tex, img_gt=img_gt, index_gt=index_gt)
    losses = {}
    for it in range(201):
        img, index_img, bary_img = shade(v)
        img = drtk.edge_grad_estimator(v_pix=v, vi=vi, bary_img=bary_img, img=img, index_img=index_img)
        loss = ((img - img_gt) ** 2).mean()
        optim.zero_grad()
        loss.backward()
        if it == 0:
            rec.update(index0=index_img.clone(), bary0=bary_img.detach().clone(), img0=img.detach().clone(), grad0=v.grad.detach().clone())
        if it in (0, 1, 50, 100, 200):
            losses[it] = float(loss)
        optim.step()
    rec["loss_iters"] = np.array(sorted(losses), dtype=np.int64)
    rec["loss_values"] = np.array([losses[k] for k in sorted(losses)], dtype=np.float64)
    same = all(th.equal(rec[k], old["out_" + k]) for k in ("index0", "bary0", "img0", "grad0", "img_gt")) and \
        np.array_equal(rec["loss_values"], np.asarray(old["out_loss_values"]))
    print(f"  two_triangles: losses {dict(zip(rec['loss_iters'].tolist(), rec['loss_values'].tolist()))}; "
          f"identical to the restated wiring's fixture: {same}")
    assert same
    save("refpy_two_triangles", {"out_" + k: x for k, x in rec.items()})


# ------------------------------------------------------------------------------------------------ sparse operators
def gen_sparse(drtk):
    """drtk.interpolation_matrix / interpolation_normal_matrix through the reference's wrappers and its module --
    including the A^T A PATTERN builder (interpolate_module.cpp:28-262: host code in the module's anonymous namespace,
    unreachable from the kernel-level build of round 1, which is why the pattern was only pinned through a property
    then) and the LRU structure cache.  Shared and per-view topology, with gradients wrt bary_img."""
    for scene, tag in (("spheres_f32", "spheres_f32"), ("ragged_f32", "ragged_f32"), ("tutorial3_f64", "tutorial3_f64")):
        sc = load_fixture(scene)
        vi, index, bary = sc["in_vi"], sc["out_index_img"], sc["out_render_bary"]
        V = sc["in_v"].shape[1]
        b = bary.clone().requires_grad_(True)
        A = drtk.interpolation_matrix(vi, index, b, V)
        g = th.Generator().manual_seed(31)
        gA = th.rand(A.values().shape, generator=g, dtype=th.float64).to(bary.dtype)
        (A.values() * gA).sum().backward()
        arrs = dict(in_gA=gA, out_A_crow=A.crow_indices().to(th.int32), out_A_col=A.col_indices().to(th.int32), out_A_values=A.values().detach(),
                    out_A_bary_grad=b.grad.clone())
        b2 = bary.clone().requires_grad_(True)
        M = drtk.interpolation_normal_matrix(vi, index, b2, V)
        gM = th.rand(M.values().shape, generator=g, dtype=th.float64).to(bary.dtype)
        (M.values() * gM).sum().backward()
        arrs.update(in_gM=gM, out_M_crow=M.crow_indices().to(th.int32), out_M_col=M.col_indices().to(th.int32), out_M_values=M.values().detach(),
                    out_M_bary_grad=b2.grad.clone())
        # the reference's own identity: M == A^T A
        err = (A.detach().to_dense().double().T @ A.detach().to_dense().double() - M.detach().to_dense().double()).abs().max().item()
        assert err < (1e-4 if bary.dtype == th.float32 else 1e-11), err
        # second call on the same topology goes through the module's structure cache: same pattern
        M2 = drtk.interpolation_normal_matrix(vi, index, bary, V)
        assert th.equal(M2.crow_indices(), M.crow_indices()) and th.equal(M2.col_indices(), M.col_indices())
        print(f"  sparse {tag}: A {tuple(A.shape)} nnz {A.values().numel()}, A^T A nnz {M.values().numel()}, |A^T A - M| = {err:.1e}")
        save(f"refpy_sparse_{tag}", arrs)


# ------------------------------------------------------------------------- screen_space_uv_derivative and autograd
def gen_uv_derivative_autograd(drtk):
    """What happens in the REFERENCE when a loss is differentiated through screen_space_uv_derivative
    (drtk/screen_space_uv_derivative.py:15-80)?  It is written as a PyTorch composite, but its last statement,
    `vt_dxdy_img[~mask, :, :] = 0` (:79), modifies the output of `th.linalg.inv_ex` in place, and that operator's backward
    needs its own output: `backward()` RAISES ("... modified by an inplace operation").  So upstream the function is in
    effect forward only -- the one consumer on the path, mipmap_grid_sample, defines no gradient for vt_dxdy_img
    (mipmap_grid_sampler_module.cpp), so nobody notices.  Recorded here (the message, and that a mask that is True
    everywhere does not help) so that the replacement's behaviour -- forward only, an error if a gradient really
    arrives -- is pinned against the reference's, not against what its source looks like."""
    sys.path.insert(0, REF)
    from drtk.screen_space_uv_derivative import screen_space_uv_derivative  # noqa: E402

    sys.path.remove(REF)
    z = load_fixture("uv_derivative_f64")
    msgs = []
    for all_true in (False, True):
        names = ("v", "vt", "bary_img", "campos", "camrot", "focal")
        leaf = {k: z["in_" + k].clone().requires_grad_(True) for k in names}
        index = z["in_index_img"]
        mask = th.ones_like(index, dtype=th.bool) if all_true else index != -1
        out = screen_space_uv_derivative(leaf["v"], leaf["vt"], z["in_vi"], z["in_vti"], index, leaf["bary_img"], mask,
                                         leaf["campos"], leaf["camrot"], leaf["focal"])
        assert out.requires_grad
        try:
            out.sum().backward()
            raise SystemExit("the reference's screen_space_uv_derivative differentiated without error: re-read this generator")
        except RuntimeError as e:
            msgs.append(str(e).split(". Hint")[0])
    assert all("inplace operation" in m and "LinalgInvExBackward" in m for m in msgs), msgs
    save("refpy_uv_derivative_autograd", {"backward_raises": np.array(True), "message": np.array(msgs[0]),
                                          "message_all_true_mask": np.array(msgs[1])})


def main():
    if not os.path.isdir(REF):
        raise SystemExit("needs /root/reference (build container only)")
    th.set_num_threads(1)
    drtk = import_reference()
    gen_transform(drtk)
    gen_step(drtk)
    gen_two_triangles(drtk)
    gen_sparse(drtk)
    gen_uv_derivative_autograd(drtk)


if __name__ == "__main__":
    main()

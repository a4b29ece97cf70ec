#!/usr/bin/env python3
"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz from the REFERENCE's own CPU kernels.

Runs only in the build container: it needs oracle/_ref/ (built by oracle/ref_build.py from
/root/reference/src).  Every fixture holds the scene inputs (tests/scenes.py) and the outputs of
the reference's strict-IEEE build for every function on the hot path, single-threaded
(torch.set_num_threads(1) => deterministic accumulation order), plus the index/depth images of the
reference built with its own `-O3 --fast-math` flags so that flag-induced differences stay
documented (SURVEY.md §7 hard part 1).

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz
"""
import os
import sys

import numpy as np
import torch as th

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import ref_build  # noqa: E402
from backends import RefBackend, make_ops  # noqa: E402
from scenes import SCENES  # noqa: E402

from drtk_amd import synthetic as S  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def run_scene(B, Bfast, sc):
    v, vi, H, W = sc["v"], sc["vi"], sc["H"], sc["W"]
    attr, gd, gb, go = sc["attr"], sc["gd"], sc["gb"], sc["go"]
    out = {}
    vi_r = sc.get("vi_raster", vi)
    depth, index = B.rasterize(v, vi_r, H, W)
    depth_f, index_f = Bfast.rasterize(v, vi_r, H, W)
    r_depth, r_bary = B.render(v, vi, index)
    interp = B.interpolate(attr, vi, index, r_bary)
    grad_v = B.render_backward(v, vi, index, gd, gb)
    attr_grad, bary_grad = B.interpolate_backward(go, attr, vi, index, r_bary, True, True)
    img = interp * (index != -1)[:, None]
    eg = B.edge_grad_backward(v, img, index, vi, go, 1e4)
    eg0 = B.edge_grad_backward(v, img, index, vi, go, 0.0)
    # the C=3 interpolate backward that routes edge gradients to v_pix (edge_grad_estimator.py:172)
    vpix_grad, _ = B.interpolate_backward(eg, v, vi, index, r_bary, True, False)
    out.update(
        depth_img=depth, index_img=index, depth_img_fast=depth_f, index_img_fast=index_f,
        render_depth=r_depth, render_bary=r_bary, interp=interp, grad_v=grad_v, attr_grad=attr_grad,
        bary_grad=bary_grad, img=img, edge_grad=eg, edge_grad_noclamp=eg0, v_pix_grad_from_edges=vpix_grad,
    )
    return out


def run_sparse(ref, sc, outs):
    """Sparse interpolation operators (interpolate_kernel_cpu.cpp:411-693) from the reference's
    strict build on a scene's index/bary images.  The A^T A pattern (crow/col/pair_indices) is
    topology-only host code that lives in an anonymous namespace of interpolate_module.cpp and is
    not reachable without the module's CUDA half, so it comes from the oracle's restatement and is
    pinned here by a property of REFERENCE outputs: the reference's normal-matrix values scattered
    through that pattern must equal A^T A of the reference's interpolation matrix."""
    import oracle as O

    vi, index, bary = sc["vi"], outs["index_img"], outs["render_bary"]
    N, V = index.shape[0], sc["v"].shape[1]
    vib = (vi[None].expand(N, -1, -1) if vi.ndim == 2 else vi).contiguous()
    crow, col, values, rows = ref.interpolation_matrix(vib, index, bary)
    g = th.Generator().manual_seed(77)
    g_im = th.rand(values.shape, generator=g, dtype=th.float64).to(values.dtype)
    im_bwd = ref.interpolation_matrix_backward(g_im, vib, index, bary, rows)
    p_crow, p_col, pair = O.normal_matrix_structure(vib, V)
    nnz = p_col.numel()
    nm_values = ref.normal_matrix_values(pair, index, bary, nnz)
    g_nm = th.rand(nnz, generator=g, dtype=th.float64).to(values.dtype)
    nm_bwd = ref.normal_matrix_values_backward(g_nm, pair, index, bary)
    A = th.sparse_csr_tensor(crow, col, values.double(), size=(rows.numel(), V)).to_dense()
    AtA = th.sparse_csr_tensor(p_crow, p_col, nm_values.double(), size=(V, V)).to_dense()
    err = (A.T @ A - AtA).abs().max().item()
    tol = 1e-4 if values.dtype == th.float32 else 1e-11
    assert err < tol, f"pattern pin failed: |A^T A - scatter(values)| = {err}"
    return dict(g_im=g_im, g_nm=g_nm), dict(
        col_indices=col.to(th.int32), values=values, row_pixels=rows.to(th.int32), im_bary_grad=im_bwd,
        nm_crow=p_crow.to(th.int32), nm_col=p_col.to(th.int32), nm_pair=pair, nm_values=nm_values, nm_bary_grad=nm_bwd,
    )


SPARSE_SCENES = (("spheres", th.float32, "f32"), ("ragged", th.float32, "f32"), ("tutorial3", th.float64, "f64"))


def save(name, inputs, outputs):
    arrs = {}
    for k, val in inputs.items():
        arrs["in_" + k] = val.numpy() if isinstance(val, th.Tensor) else np.asarray(val)
    for k, val in outputs.items():
        arrs["out_" + k] = val.numpy() if isinstance(val, th.Tensor) else np.asarray(val)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  {path}: {os.path.getsize(path) / 1024:.1f} KiB")


def frozen_inputs(fixture, generated):
    """Scene inputs are DATA: once a fixture exists its `in_*` arrays are the scene (the projected spheres come out of
    float64 sin / cos / matmul, which this torch build does not reproduce bit for bit from run to run -- 1e-14 jitter in
    `spheres_f64`'s vertices made that fixture the one file a regeneration could not reproduce byte for byte).  With
    `--fresh` the inputs are generated anew."""
    path = os.path.join(OUT, fixture + ".npz")
    if "--fresh" in sys.argv or not os.path.isfile(path):
        return generated
    z = np.load(path)
    out = dict(generated)
    for k in z.files:
        if not k.startswith("in_"):
            continue
        a = z[k]
        name = k[3:]
        if name in generated and isinstance(generated[name], th.Tensor):
            t = th.from_numpy(np.ascontiguousarray(a))
            assert t.shape == generated[name].shape and t.dtype == generated[name].dtype, (fixture, name)
            assert float((t.double() - generated[name].double()).abs().max()) <= 1e-9 * max(1.0, float(t.double().abs().max())), \
                f"{fixture}: stored input {name} is not the scene the generator describes"
            out[name] = t
    return out


def two_triangles_trajectory(ops):
    """The reference's only 'test' (test/two_triangles.py:14-92) at 64x64 on CPU: GT render,
    perturbed start, Adam.  Records iteration-0 tensors and the loss at a few iterations."""
    import torch.nn.functional as thf

    v_gt, vi, vt, tex = S.two_triangles(64, 64)
    g = th.Generator().manual_seed(10)
    noise = th.randn(v_gt.shape, generator=g, dtype=th.float32) * (20.0 / 8.0)
    v = th.nn.Parameter((v_gt + noise).contiguous())

    def shade(vv):
        index_img = ops.rasterize(vv, vi, 64, 64)
        _, bary_img = ops.render(vv, vi, index_img)
        vt_img = ops.interpolate(vt, vi, index_img, bary_img).permute(0, 2, 3, 1)
        img = thf.grid_sample(tex, vt_img, padding_mode="border", align_corners=False) * (index_img != -1)[:, None]
        return img, index_img, bary_img

    with th.no_grad():
        img_gt, index_gt, _ = shade(v_gt)
    optim = th.optim.Adam([v], lr=0.05, betas=(0.9, 0.999))
    rec = {"v_gt": v_gt, "v0": v.detach().clone(), "vi": vi, "vt": vt, "tex": tex, "img_gt": img_gt, "index_gt": index_gt}
    losses = {}
    for it in range(201):
        img, index_img, bary_img = shade(v)
        img = ops.edge_grad_estimator(v_pix=v, vi=vi, bary_img=bary_img, img=img, index_img=index_img)
        loss = ((img - img_gt) ** 2).mean()
        optim.zero_grad()
        loss.backward()
        if it == 0:
            rec.update(index0=index_img.clone(), bary0=bary_img.detach().clone(), img0=img.detach().clone(),
                       grad0=v.grad.detach().clone())
        if it in (0, 1, 50, 100, 200):
            losses[it] = float(loss)
        optim.step()
    rec["loss_iters"] = np.array(sorted(losses), dtype=np.int64)
    rec["loss_values"] = np.array([losses[k] for k in sorted(losses)], dtype=np.float64)
    return rec


def main():
    if not ref_build.available():
        raise SystemExit("needs /root/reference and triton's cuda_runtime.h (build container only)")
    ref_build.build(verbose=True)
    th.set_num_threads(1)
    os.makedirs(OUT, exist_ok=True)
    B, Bfast = RefBackend("strict"), RefBackend("fast")
    for name, fn in SCENES.items():
        for dtype, tag in ((th.float32, "f32"), (th.float64, "f64")):
            if dtype == th.float64 and name not in ("tutorial3", "spheres", "edge_cases"):
                continue
            sc = frozen_inputs(f"{name}_{tag}", fn(dtype))
            outs = run_scene(B, Bfast, sc)
            nd = int((outs["index_img"] != outs["index_img_fast"]).sum())
            print(f"{name}/{tag}: covered {(outs['index_img'] >= 0).sum().item()} px, "
                  f"index px differing under --fast-math: {nd}")
            save(f"{name}_{tag}", sc, outs)
            if (name, dtype, tag) in SPARSE_SCENES:
                save(f"sparse_{name}_{tag}", *run_sparse(ref_build.load("strict"), sc, outs))

    # end-to-end step (SURVEY.md §8d definition) on the sphere scene
    ops = make_ops(B)
    sc = SCENES["spheres"](th.float32)
    v = sc["v"].clone().requires_grad_(True)
    attr = sc["attr"].clone().requires_grad_(True)
    loss, index_img = S.fwd_bwd_step(v, sc["vi"], attr, sc["H"], sc["W"], ops=ops)
    save("step_spheres_f32", dict(v=sc["v"], vi=sc["vi"], attr=sc["attr"], H=sc["H"], W=sc["W"]),
         dict(loss=loss.detach(), index_img=index_img, v_grad=v.grad, attr_grad=attr.grad))

    rec = two_triangles_trajectory(ops)
    print("two_triangles trajectory:", dict(zip(rec["loss_iters"].tolist(), rec["loss_values"].tolist())))
    save("two_triangles_trajectory", {}, rec)


if __name__ == "__main__":
    main()

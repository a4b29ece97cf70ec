"""CPU: the oracle restatement (oracle/drtk_oracle.c) against the committed golden vectors, which
were produced by the reference's own CPU kernels (oracle/gen_golden.py).  Everything is bit-exact:
index/depth images, every forward float tensor, and -- single-threaded, same accumulation order --
every gradient."""
import pytest
import torch as th
from conftest import GOLDEN_SCENES, SPARSE_SCENES, load_golden, load_sparse

import oracle as O


@pytest.mark.parametrize("name", GOLDEN_SCENES)
def test_oracle_matches_reference_fixture(name):
    i, o = load_golden(name)
    v, vi, H, W = i["v"], i["vi"], i["H"], i["W"]
    vi_r = i.get("vi_raster", vi)
    depth, index = O.rasterize(v, vi_r, H, W)
    assert th.equal(index, o["index_img"])
    assert th.equal(depth, o["depth_img"])
    assert depth.dtype == th.float32  # rasterize depth is float32 even for float64 v
    r_depth, r_bary = O.render(v, vi, index)
    assert th.equal(r_depth, o["render_depth"]) and th.equal(r_bary, o["render_bary"])
    interp = O.interpolate(i["attr"], vi, index, r_bary)
    assert th.equal(interp, o["interp"])
    assert th.equal(O.render_backward(v, vi, index, i["gd"], i["gb"]), o["grad_v"])
    ag, bg = O.interpolate_backward(i["go"], i["attr"], vi, index, r_bary)
    assert th.equal(ag, o["attr_grad"]) and th.equal(bg, o["bary_grad"])
    img = interp * (index != -1)[:, None]
    assert th.equal(img, o["img"])
    assert th.equal(O.edge_grad_backward(v, img, index, vi, i["go"], 1e4), o["edge_grad"])
    assert th.equal(O.edge_grad_backward(v, img, index, vi, i["go"], 0.0), o["edge_grad_noclamp"])
    vg, none = O.interpolate_backward(o["edge_grad"], v, vi, index, r_bary, True, False)
    assert none is None and th.equal(vg, o["v_pix_grad_from_edges"])


@pytest.mark.parametrize("name", GOLDEN_SCENES)
def test_fast_math_reference_build_only_moves_depth_lsbs(name):
    """The reference's own flags (-O3 --fast-math) change depth LSBs but, on these scenes, no
    index_img pixel; any future mismatch against a fast-math build must be a depth near-tie."""
    _, o = load_golden(name)
    assert th.equal(o["index_img"], o["index_img_fast"])
    d, df = o["depth_img"].double(), o["depth_img_fast"].double()
    assert ((d - df).abs() <= 4e-7 * d.abs().clamp(min=1.0)).all()


def test_known_answers_edge_cases():
    """Appendix E of SURVEY.md: answers known independently of any build."""
    i, o = load_golden("edge_cases_f32")
    idx = o["index_img"]
    # 0: coincident triangles of opposite winding and equal depth: lower id wins everywhere
    assert set(idx[0].unique().tolist()) == {-1, 0}
    # 1: quad [0,6]^2 split on the diagonal: exactly x,y in 0..5 covered, x >= y -> tri 0, else tri 1
    cov = idx[1] >= 0
    exp = th.zeros(16, 16, dtype=th.bool)
    exp[:6, :6] = True
    assert th.equal(cov, exp)
    yy, xx = th.meshgrid(th.arange(16), th.arange(16), indexing="ij")
    assert th.equal(idx[1][exp], th.where(xx >= yy, 0, 1).to(th.int32)[exp])
    # 2: z = 0 and z = 1e-9 vertices: whole triangles culled (near plane is 1e-8, no clipping)
    assert (idx[2] == -1).all()
    # 3: off-screen, and zero-area (a,a,b)
    assert (idx[3] == -1).all()
    # 4: far-away huge triangle covers the canvas except where the nearer small one wins
    assert (idx[4] >= 0).all() and (idx[4] == 1).sum() > 0
    # empty pixels: depth exactly 0, bary exactly 0, interpolate = +-1 coordinate sweep
    assert (o["depth_img"][idx == -1] == 0).all()
    assert (o["render_bary"].permute(0, 2, 3, 1)[idx == -1] == 0).all()
    bgx = (th.arange(16, dtype=th.float32) * 2 + 1) / 16 - 1
    assert th.equal(o["interp"][2, 0], bgx[None, :].expand(16, 16))
    assert th.equal(o["interp"][2, 1], bgx[:, None].expand(16, 16))


def test_bary_sums_to_one_and_depth_in_range():
    i, o = load_golden("spheres_f32")
    cov = o["index_img"] >= 0
    s = o["render_bary"].sum(1)
    assert ((s[cov] - 1).abs() < 1e-5).all()
    assert (o["render_depth"][cov] > 1.9).all() and (o["render_depth"][cov] < 4.2).all()


def test_two_triangles_trajectory_reproduced_by_oracle(oracle_ops):
    """Config (1) of BASELINE.json: test/two_triangles.py at 64x64 on the CPU path.  The oracle
    pipeline must reproduce the reference's iteration-0 tensors exactly and its loss curve."""
    import numpy as np
    import torch.nn.functional as thf

    _, r = load_golden("two_triangles_trajectory")
    ops = oracle_ops
    vi, vt, tex, img_gt = r["vi"], r["vt"], r["tex"], r["img_gt"]
    v = th.nn.Parameter(r["v0"].clone())
    optim = th.optim.Adam([v], lr=0.05, betas=(0.9, 0.999))
    want = dict(zip(r["loss_iters"].tolist(), r["loss_values"].tolist()))
    for it in range(201):
        index_img = ops.rasterize(v, vi, 64, 64)
        _, bary_img = ops.render(v, vi, index_img)
        vt_img = ops.interpolate(vt, vi, index_img, bary_img).permute(0, 2, 3, 1)
        img = thf.grid_sample(tex, vt_img, padding_mode="border", align_corners=False) * (index_img != -1)[:, None]
        img = ops.edge_grad_estimator(v_pix=v, vi=vi, bary_img=bary_img, img=img, index_img=index_img)
        loss = ((img - img_gt) ** 2).mean()
        optim.zero_grad()
        loss.backward()
        if it == 0:
            assert th.equal(index_img, r["index0"]) and th.equal(bary_img.detach(), r["bary0"])
            assert th.equal(img.detach(), r["img0"]) and th.equal(v.grad, r["grad0"])
        if it in want:
            assert abs(float(loss.detach()) - want[it]) <= 1e-3 * want[it] + 1e-9, (it, float(loss.detach()), want[it])
        optim.step()
    assert float(loss.detach()) < 0.2 * want[0]


@pytest.mark.parametrize("name", SPARSE_SCENES)
def test_oracle_sparse_operators_match_reference_fixture(name):
    """interpolation_matrix / normal-matrix values, forward and backward: bit-exact against the
    reference's CPU kernels (interpolate_kernel_cpu.cpp:411-693)."""
    vi, index, bary, V, gi, go = load_sparse(name)
    crow, col, values, rows = O.interpolation_matrix(vi, index, bary)
    R = rows.numel()
    assert R == int((index != -1).sum()) and th.equal(crow, th.arange(0, 3 * R + 1, 3))
    assert th.equal(col.int(), go["col_indices"]) and th.equal(values, go["values"])
    assert th.equal(rows.int(), go["row_pixels"])
    assert (col.view(-1, 3)[:, 0] < col.view(-1, 3)[:, 1]).all() and (col.view(-1, 3)[:, 1] < col.view(-1, 3)[:, 2]).all()
    assert th.equal(O.interpolation_matrix_backward(gi["g_im"], vi, index, bary, rows), go["im_bary_grad"])
    p_crow, p_col, pair = O.normal_matrix_structure(vi, V)
    assert th.equal(p_crow.int(), go["nm_crow"]) and th.equal(p_col.int(), go["nm_col"]) and th.equal(pair, go["nm_pair"])
    nnz = p_col.numel()
    assert th.equal(O.normal_matrix_values(pair, index, bary, nnz), go["nm_values"])
    assert th.equal(O.normal_matrix_values_backward(gi["g_nm"], pair, index, bary), go["nm_bary_grad"])


@pytest.mark.parametrize("name", SPARSE_SCENES)
def test_sparse_pattern_is_pinned_by_reference_values(name):
    """The A^T A pattern is restated topology code; pin it with reference outputs only: the
    reference's normal-matrix values placed through the pattern equal A^T A of the reference's
    interpolation matrix, the pattern is symmetric, sorted and duplicate-free."""
    vi, index, bary, V, _, go = load_sparse(name)
    R = go["row_pixels"].numel()
    A = th.sparse_csr_tensor(th.arange(0, 3 * R + 1, 3), go["col_indices"].long(), go["values"].double(), size=(R, V))
    crow, col = go["nm_crow"].long(), go["nm_col"].long()
    AtA = th.sparse_csr_tensor(crow, col, go["nm_values"].double(), size=(V, V)).to_dense()
    ref = A.to_dense().T @ A.to_dense()
    tol = 1e-4 if go["values"].dtype == th.float32 else 1e-11
    assert (AtA - ref).abs().max() < tol
    rows = th.repeat_interleave(th.arange(V), crow[1:] - crow[:-1])
    keys = rows * V + col
    assert (keys[1:] > keys[:-1]).all()
    dense_pat = th.zeros(V, V, dtype=th.bool)
    dense_pat[rows, col] = True
    assert th.equal(dense_pat, dense_pat.T)
    # every referenced vertex has its diagonal; the pattern is exactly the face adjacency
    adj = th.zeros(V, V, dtype=th.bool)
    f = vi.reshape(-1, 3).long()
    for a in range(3):
        for b in range(3):
            adj[f[:, a], f[:, b]] = True
    assert th.equal(dense_pat, adj)


def test_wireframe_restatement_gives_the_hand_derived_known_answers():
    """Wireframe mode has no reference CPU twin and no reference test: the restatement is held to answers worked out on
    paper from the rules of rasterize_kernel.cu:170-400 (tests/wireframe_known_answers.py: diamond rule on axis-aligned,
    45-degree and slope-2 edges incl. corner touches, all 16 nibble values, occluding fill, drawn-beats-fill ties, the
    unwritten canvas border, the culls), and to the same rules evaluated in exact rational arithmetic on random
    quarter-pixel-grid scenes."""
    import wireframe_known_answers as K

    for dt in (th.float32, th.float64):
        K.check(lambda v, vi, H, W: O.rasterize_lines(v, vi, H, W), dt)
    K.check_against_exact_model(lambda v, vi, H, W: O.rasterize_lines(v, vi, H, W), range(16))
    K.check_against_exact_model(lambda v, vi, H, W: O.rasterize_lines(v, vi, H, W), range(16, 20), th.float64)


def test_oracle_reproduces_the_strict_image_of_the_fast_math_fixture():
    """tests/golden/fastmath_owner_changes_100k.npz (oracle/gen_golden_fastmath.py) carries SHA-256s of the reference's
    strict-IEEE index / depth image of one full benchmark view: the oracle restatement must produce exactly that image
    (the GPU suite holds the HIP rasterizer to the same hashes)."""
    import hashlib

    import numpy as np
    import oracle as O
    from conftest import GOLDEN
    from drtk_amd import synthetic as S

    z = np.load(f"{GOLDEN}/fastmath_owner_changes_100k.npz")
    res = int(z["res"])
    nl, no = S.MESH_SIZES["100k"]
    _, vi = S.uv_sphere(nl, no, lobes=0.05)
    v = th.from_numpy(z["v"])[None]
    d, i = O.rasterize(v, vi, res, res, nthreads=0)
    sha = lambda t: hashlib.sha256(np.ascontiguousarray(t.numpy()).tobytes()).hexdigest()  # noqa: E731
    assert sha(i) == str(z["sha256_index_strict"]) and sha(d) == str(z["sha256_depth_strict"])
    px = th.from_numpy(z["pixels"])
    assert th.equal(i.flatten()[px], th.from_numpy(z["index_strict"])) and not th.equal(i.flatten()[px], th.from_numpy(z["index_fast"]))

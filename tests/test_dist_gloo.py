"""CPU, world_size 2 over gloo: the multi-GPU scheme of drtk_amd/dist.py -- views shard across
ranks with no data-path collective, ONE fused all-reduce sums the gradients of the view-shared
tensors -- reproduces the single-process result.  Per-rank compute uses the CPU oracle ops (test
infrastructure); the product's part under test is the sharding + SharedGradReducer logic."""
import os
import socket
import sys

import pytest
import torch as th
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _scene():
    from drtk_amd import synthetic as S

    N, H, W, C = 4, 40, 48, 5
    v_world, vi = S.uv_sphere(12, 14)
    cams = S.ring_cameras(N, W, H)
    attr = S.random_attributes(1, v_world.shape[0], C, seed=3)[:1].contiguous()
    return N, H, W, C, v_world, vi, cams, attr


def _local_step(ops, v_world, attr, vi, cams, views, H, W):
    from drtk_amd.transform import transform

    campos, camrot, focal, princpt = (t[views.start:views.stop] for t in cams)
    n = len(views)
    v_pix = transform(v_world[None].expand(n, -1, -1), campos, camrot, focal, princpt)
    a = attr.expand(n, -1, -1)
    index_img = ops.rasterize(v_pix, vi, H, W)
    depth_img, bary_img = ops.render(v_pix, vi, index_img)
    img = ops.interpolate(a, vi, index_img, bary_img)
    img = th.where((index_img != -1)[:, None], img, 0.0)
    img = ops.edge_grad_estimator(v_pix=v_pix, vi=vi, bary_img=bary_img, img=img, index_img=index_img)
    # sums (not means) so that the loss of a shard is the shard's part of the global loss
    loss = (img * img).sum() + depth_img.sum()
    loss.backward()
    return float(loss.detach())


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    th.set_num_threads(1)
    from backends import OracleBackend, make_ops

    from drtk_amd import dist as ddist

    r, w, _ = ddist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    N, H, W, C, v_world, vi, cams, attr = _scene()
    v_world = v_world.clone().requires_grad_(True)
    attr = attr.clone().requires_grad_(True)
    views = ddist.shard_views(N, rank, world)
    ops = make_ops(OracleBackend(1))
    out = {"views": list(views)}
    # (a) overlapped: hooks launch each segment's all-reduce as its gradient becomes final; (b) everything after the
    # backward pass in one call; (c) overlapped, with `.grad` replaced behind the reducer's back (set_to_none) and a
    # second step on the same reducer
    for tag, overlap in (("", True), ("_late", False)):
        red = ddist.SharedGradReducer([v_world, attr], overlap=overlap)
        assert red.nbytes() == 4 * (v_world.numel() + attr.numel())
        assert v_world.grad.data_ptr() == red.flat.data_ptr()
        order = []
        if overlap:
            launch = red._launch
            red._launch = lambda i, launch=launch, order=order: (order.append(i), launch(i))[1]
        loss = _local_step(ops, v_world, attr, vi, cams, views, H, W)
        if overlap:
            assert order[:2] == [1, 0], order  # attributes first (final after interpolate backward), vertices last
            assert set(red._pending) == {0, 1}  # both collectives were launched from inside the backward pass
        else:
            assert not red._pending
        red.finish()
        out.update({"v" + tag: v_world.grad.clone(), "a" + tag: attr.grad.clone(), "loss": loss})
        for h in red._handles:
            h.remove()
    red = ddist.SharedGradReducer([v_world, attr])
    for step in range(2):
        if step == 0:
            red.zero_grad()
        else:
            v_world.grad, attr.grad = None, None  # as optimizer.zero_grad(set_to_none=True) would
        _local_step(ops, v_world, attr, vi, cams, views, H, W)
        red.all_reduce()
        out.update({f"v_step{step}": v_world.grad.clone(), f"a_step{step}": attr.grad.clone()})
    ddist.barrier_and_sync()
    th.save(out, os.path.join(out_dir, f"r{rank}.pt"))
    th.distributed.destroy_process_group()


def test_shard_views_partition():
    from drtk_amd.dist import shard_views

    for n in (1, 7, 8, 64):
        for w in (1, 2, 3, 8):
            parts = [list(shard_views(n, r, w)) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


@pytest.mark.timeout(300)
def test_two_ranks_match_single_process(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from backends import OracleBackend, make_ops

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    res = [th.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    assert res[0]["views"] == [0, 1] and res[1]["views"] == [2, 3]
    # every rank ends up with the same, summed gradients
    assert th.equal(res[0]["v"], res[1]["v"]) and th.equal(res[0]["a"], res[1]["a"])
    # overlapped == reduced after the backward pass == a later step on a reducer whose .grad views had been dropped
    for r in res:
        for k in ("v", "a"):
            assert th.equal(r[k], r[k + "_late"]) and th.equal(r[k], r[k + "_step0"]) and th.equal(r[k], r[k + "_step1"]), k

    th.set_num_threads(1)
    N, H, W, C, v_world, vi, cams, attr = _scene()
    v_world = v_world.clone().requires_grad_(True)
    attr = attr.clone().requires_grad_(True)
    loss = _local_step(make_ops(OracleBackend(1)), v_world, attr, vi, cams, range(0, N), H, W)
    assert abs(loss - (res[0]["loss"] + res[1]["loss"])) <= 1e-3 * abs(loss)
    for got, want in ((res[0]["v"], v_world.grad), (res[0]["a"], attr.grad)):
        tol = 1e-5 + 1e-5 * float(want.abs().max())
        assert float((got - want).abs().max()) <= tol * 10

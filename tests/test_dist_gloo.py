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
            red._launch = lambda g, launch=launch, order=order: (order.append(g), launch(g))[1]
        loss = _local_step(ops, v_world, attr, vi, cams, views, H, W)
        if overlap:
            assert order == [1, 0], order  # attributes first (final after interpolate backward), vertices last
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
    for h in red._handles:
        h.remove()

    # (d) a second backward pass before finish() must not add local gradients to a sum that was already reduced:
    # it raises; gradient accumulation goes through no_sync() -- two passes, one reduction of their sum
    red = ddist.SharedGradReducer([v_world, attr])
    _local_step(ops, v_world, attr, vi, cams, views, H, W)
    try:
        _local_step(ops, v_world, attr, vi, cams, views, H, W)
        out["second_backward_raised"] = False
    except RuntimeError as e:
        out["second_backward_raised"] = "no_sync" in str(e)
    red.finish()
    red.zero_grad()
    with red.no_sync():
        _local_step(ops, v_world, attr, vi, cams, views, H, W)
        assert not red._pending
    _local_step(ops, v_world, attr, vi, cams, views, H, W)
    red.finish()
    out.update({"v_accum": v_world.grad.clone(), "a_accum": attr.grad.clone()})
    for h in red._handles:
        h.remove()

    # (e) a rank whose shard is EMPTY (one view, two ranks) runs no backward pass at all: no hook fires there, finish()
    # issues the same collectives in the same fixed order as the rank that launched them from its hooks
    red = ddist.SharedGradReducer([v_world, attr])
    one = ddist.shard_views(1, rank, world)
    launched = []
    launch = red._launch
    red._launch = lambda g, launch=launch: (launched.append((g, bool(red._ready))), launch(g))[1]
    if len(one):
        _local_step(ops, v_world, attr, vi, cams, one, H, W)
    red.finish()
    out.update({"v_one": v_world.grad.clone(), "a_one": attr.grad.clone(), "one_views": list(one), "one_launched": launched})
    for h in red._handles:
        h.remove()

    # (f) leaves stored in fp16 (attributes), reduced as ONE group with the vertices' own collective beside it: staged in
    # float32 -- accumulated unrounded through upcast(), summed over the ranks in float32, rounded to fp16 once
    attr16 = attr.detach().half().requires_grad_(True)
    extra16 = th.ones(3, dtype=th.float16, requires_grad=True)  # a second member of the group; no gradient on rank 1
    red = ddist.SharedGradReducer([v_world, [attr16, extra16]], dtype=th.float32)
    assert red.collectives_per_step() == 2 and red.flat.dtype == th.float32 and attr16.grad is None
    v_world.grad = None
    n = len(views)
    from drtk_amd.transform import transform

    campos, camrot, focal, princpt = (t[views.start:views.stop] for t in cams)
    v_pix = transform(v_world[None].expand(n, -1, -1), campos, camrot, focal, princpt)
    a32 = red.upcast(attr16)
    assert a32.dtype == th.float32 and th.equal(a32, attr16.float())
    index_img = ops.rasterize(v_pix, vi, H, W)
    depth_img, bary_img = ops.render(v_pix, vi, index_img)
    img = ops.interpolate(a32.expand(n, -1, -1), vi, index_img, bary_img)
    img = th.where((index_img != -1)[:, None], img, 0.0)
    extra = red.upcast(extra16).sum() * (3.0 if rank == 0 else 0.0) if rank == 0 else 0.0
    ((img * img).sum() + depth_img.sum() + extra).backward()
    red.finish()
    i16 = next(k for k, q in enumerate(red.params) if q is attr16)
    out.update({"a16_f32_sum": red.segments[i16].clone().view_as(attr16), "a16_grad": attr16.grad.clone(),
                "extra16_grad": extra16.grad.clone(), "a16_input": attr16.detach().float()})
    for h in red._handles:
        h.remove()

    # (g) a staged leaf that reaches the loss through upcast() TWICE and directly (`p.float()`) as well: every part of
    # its gradient must arrive in the reduced sum, and the collective is launched once, after the last of them
    w16 = th.full((5,), 0.5, dtype=th.float16, requires_grad=True)
    red = ddist.SharedGradReducer([w16], dtype=th.float32)
    launched = []
    launch = red._launch
    red._launch = lambda g, launch=launch: (launched.append(g), launch(g))[1]
    s = float(rank + 1)
    ((red.upcast(w16) ** 2).sum() * s + (red.upcast(w16) * 2.0).sum() + (w16.float() * 3.0).sum()).backward()
    assert launched == [0] and w16.grad is None  # launched from the leaf's hook, the direct part moved into the segment
    red.finish()
    out.update({"mixed_grad": w16.grad.clone(), "mixed_f32": red.segments[0].clone()})
    for h in red._handles:
        h.remove()

    # (h) upcast() as the leaf's ONLY use: _StagedCast.backward returns no gradient for the leaf, so the leaf's
    # post-accumulate hook has to fire with an undefined gradient for the group to be launched from inside the backward pass
    # (that it does is engine behaviour, not documented API).  If it ever stops firing, results stay correct -- finish()
    # launches what is left -- but the overlap is lost silently: pinned here, beside the reducer's own end-of-backward
    # fallback (dist.py: _StagedCast queues an engine callback that marks such leaves ready)
    c16 = th.full((4,), 0.25, dtype=th.float16, requires_grad=True)
    red = ddist.SharedGradReducer([c16], dtype=th.float32)
    launched = []
    launch = red._launch
    red._launch = lambda g, launch=launch: (launched.append(g), launch(g))[1]
    ((red.upcast(c16) ** 2).sum() * float(rank + 1)).backward()
    assert launched == [0], "the cast-only staged leaf's group was not launched from the backward pass"
    red.finish()
    assert launched == [0]
    out.update({"castonly_grad": c16.grad.clone()})
    for h in red._handles:
        h.remove()

    # (i) the end-of-backward fallback ITSELF, after accumulation passes: the leaf's hook is taken away (the case the fallback
    # exists for), two passes accumulate under no_sync() -- they fill `_staged_seen` for the round -- and the synchronising
    # pass must still queue the callback and launch the group from inside backward() (round 5 queued it only while
    # `_staged_seen` was empty: after no_sync() passes nothing was queued and the overlap was lost silently)
    d16 = th.full((4,), 0.25, dtype=th.float16, requires_grad=True)
    red = ddist.SharedGradReducer([d16], dtype=th.float32)
    for h in red._handles:
        h.remove()  # no leaf hook: only the fallback can mark the leaf ready
    launched = []
    launch = red._launch
    red._launch = lambda g, launch=launch: (launched.append(g), launch(g))[1]
    with red.no_sync():
        for _ in range(2):
            ((red.upcast(d16) ** 2).sum() * float(rank + 1)).backward()
            assert launched == [] and not red._fallback_queued
    ((red.upcast(d16) ** 2).sum() * float(rank + 1)).backward()
    assert launched == [0], "after no_sync() passes the synchronising pass did not launch the group from its end-of-backward fallback"
    red.finish()
    assert launched == [0]
    out.update({"fallback_accum_grad": d16.grad.clone()})
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
    ops = make_ops(OracleBackend(1))
    loss = _local_step(ops, v_world, attr, vi, cams, range(0, N), H, W)
    assert abs(loss - (res[0]["loss"] + res[1]["loss"])) <= 1e-3 * abs(loss)
    for got, want in ((res[0]["v"], v_world.grad), (res[0]["a"], attr.grad)):
        tol = 1e-5 + 1e-5 * float(want.abs().max())
        assert float((got - want).abs().max()) <= tol * 10
    # (d) second backward before finish() raises and names no_sync(); two accumulated passes = twice the gradients
    for r in res:
        assert r["second_backward_raised"] is True
        for k, want in (("v_accum", v_world.grad), ("a_accum", attr.grad)):
            assert float((r[k] - 2 * want).abs().max()) <= 2e-4 * float(want.abs().max()), k
    # (e) one view over two ranks: rank 1 has none, ran no backward pass, launched both collectives from finish() in the
    # order rank 0 launched them from its hooks; both end with the one-view gradients
    assert res[0]["one_views"] == [0] and res[1]["one_views"] == []
    assert [g for g, _ in res[0]["one_launched"]] == [g for g, _ in res[1]["one_launched"]] == [1, 0]
    assert all(ready for _, ready in res[0]["one_launched"]) and not any(ready for _, ready in res[1]["one_launched"])
    v1, a1 = v_world.detach().clone().requires_grad_(True), attr.detach().clone().requires_grad_(True)
    _local_step(ops, v1, a1, vi, cams, range(0, 1), H, W)
    for r in res:
        for got, want in ((r["v_one"], v1.grad), (r["a_one"], a1.grad)):
            assert float((got - want).abs().max()) <= 1e-5 + 1e-5 * float(want.abs().max())
    # (f) fp16 leaves staged in float32: the float32 sum over the ranks equals the single-process float32 gradient (at
    # the fp16-stored attribute values) to 1e-5, the leaf's .grad is that sum rounded ONCE, and the group member that
    # only rank 0 used carries rank 0's gradient on both ranks
    a_in = res[0]["a16_input"].clone().requires_grad_(True)
    v2 = v_world.detach().clone().requires_grad_(True)
    _local_step(ops, v2, a_in, vi, cams, range(0, N), H, W)
    for r in res:
        tol = 1e-5 + 1e-5 * float(a_in.grad.abs().max())
        assert float((r["a16_f32_sum"] - a_in.grad).abs().max()) <= tol
        assert r["a16_grad"].dtype == th.float16 and th.equal(r["a16_grad"], r["a16_f32_sum"].half())
        assert th.equal(r["extra16_grad"], th.full((3,), 3.0, dtype=th.float16))
        # (g) sum over ranks r of  2 * 0.5 * (r + 1)  +  2  +  3   =  (1 + 2) + 2 * 5
        assert th.equal(r["mixed_f32"], th.full((5,), 13.0)) and th.equal(r["mixed_grad"], th.full((5,), 13.0, dtype=th.float16))
        # (i) three passes of d/dx x^2 * (r + 1) at x = 0.25, summed over ranks 0 and 1: 3 * 2 * 0.25 * (1 + 2)
        assert th.equal(r["fallback_accum_grad"], th.full((4,), 4.5, dtype=th.float16))


def _shard_worker(rank, world, port, out_dir, n_views, H, W):
    """One rank of a view-sharded step: BASELINE configs[3]'s partitioning (contiguous blocks of views, shared world-space
    vertices and attributes, two collectives in the fixed order) at reduced resolution."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    th.set_num_threads(1)
    from backends import OracleBackend, make_ops

    from drtk_amd import dist as ddist
    from drtk_amd import synthetic as S

    ddist.init_from_env(backend="gloo")
    v_world, vi = S.uv_sphere(12, 14)
    cams = S.ring_cameras(n_views, W, H)
    attr = S.random_attributes(1, v_world.shape[0], 5, seed=3)[:1].contiguous()
    v_world = v_world.double().requires_grad_(True)  # float64: the comparison below is about the SHARDING, at 1e-6
    attr = attr.double().requires_grad_(True)
    cams = tuple(t.double() for t in cams)
    views = ddist.shard_views(n_views, rank, world)
    red = ddist.SharedGradReducer([v_world, attr])
    launched = []
    launch = red._launch
    red._launch = lambda g, launch=launch: (launched.append(g), launch(g))[1]
    loss = _local_step(make_ops(OracleBackend(1)), v_world, attr, vi, cams, views, H, W) if len(views) else 0.0
    red.finish()
    th.save({"views": list(views), "v": v_world.grad.clone(), "a": attr.grad.clone(), "loss": loss, "launched": launched},
            os.path.join(out_dir, f"r{rank}.pt"))
    ddist.barrier_and_sync()
    th.distributed.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,n_views", [(8, 64), (8, 3), (4, 11), (8, 13)],
                         ids=["configs3_64_views_over_8", "3_views_over_8_five_empty_shards", "11_views_over_4", "13_views_over_8"])
def test_view_sharding_at_world_sizes_4_and_8(tmp_path, world, n_views):
    """BASELINE configs[3]'s sharding (64 views -> 8 per rank) at reduced resolution, a batch smaller than the world
    (five ranks own nothing and run no backward pass) and uneven remainders: every rank ends with the same gradients,
    they equal the single-process gradients to 1e-6, and every rank issued the same two collectives in the same order."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from backends import OracleBackend, make_ops

    from drtk_amd import synthetic as S

    H, W = 24, 32
    mp.spawn(_shard_worker, args=(world, _free_port(), str(tmp_path), n_views, H, W), nprocs=world, join=True)
    res = [th.load(os.path.join(tmp_path, f"r{r}.pt")) for r in range(world)]
    assert sorted(sum((r["views"] for r in res), [])) == list(range(n_views))
    sizes = [len(r["views"]) for r in res]
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    if n_views == 64:
        assert sizes == [8] * 8
    if n_views == 3:
        assert sizes == [1, 1, 1, 0, 0, 0, 0, 0]
    for r in res:
        assert r["launched"] == [1, 0]  # attributes' group first, vertices last -- from hooks or, on an empty shard, finish()
        assert th.equal(r["v"], res[0]["v"]) and th.equal(r["a"], res[0]["a"])
    th.set_num_threads(4)
    v_world, vi = S.uv_sphere(12, 14)
    cams = tuple(t.double() for t in S.ring_cameras(n_views, W, H))
    attr = S.random_attributes(1, v_world.shape[0], 5, seed=3)[:1].contiguous().double().requires_grad_(True)
    v_world = v_world.double().requires_grad_(True)
    loss = _local_step(make_ops(OracleBackend(1)), v_world, attr, vi, cams, range(n_views), H, W)
    assert abs(loss - sum(r["loss"] for r in res)) <= 1e-9 * abs(loss)
    for got, want in ((res[0]["v"], v_world.grad), (res[0]["a"], attr.grad)):
        assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())

#!/usr/bin/env python3
"""Headline benchmark: fwd+bwd Mpixels/s of rasterize -> render -> interpolate -> edge_grad.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the one the metric is quoted on): per GPU 8 camera views of a
100 352-triangle "head" UV sphere at 2048x2048, 16 attribute channels, float32.  One step =
transform -> rasterize -> render -> interpolate(C=16) -> mask -> edge_grad_estimator ->
loss = mean(img^2) + mean(depth) -> backward, producing gradients for the SHARED world-space
vertices [V,3] and SHARED attributes [1,V,C]; with N > 1 ranks every rank renders its own 8 views
(weak scaling) and the shared gradients are summed with ONE fused RCCL all-reduce per step.
All inputs are resident in HBM before the timed region.

Rank 0 prints ONE JSON line.  `value` = all ranks' pixels / max-over-ranks wall time of the K timed
steps.  `roofline` prices the dominant kernel (algorithmic bytes per SURVEY.md §8d / its HIP-event
time, measured on the launch stream right after the timed region on the same tensors);
`cpu_baseline` times the CPU oracle on this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch as th

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--views", type=int, default=8, help="views per GPU")
    ap.add_argument("--mesh", default="100k", choices=["10k", "100k", "250k", "1M"])
    ap.add_argument("--res", type=int, default=2048)
    ap.add_argument("--channels", type=int, default=16)
    ap.add_argument("--cpu-sample-views", type=int, default=2, help="views timed on the CPU oracle (0 = skip)")
    ap.add_argument("--kernel-reps", type=int, default=5)
    return ap.parse_args()


def algorithmic_bytes_per_px(C):
    """bytes that must cross HBM per pixel for each op."""
    return {
        "rasterize": 8,
        "render": 4 + 16,
        "interpolate": 16 + 4 * C,
        "interpolate_vpix": 16 + 12,
        "edge_grad_backward": 4 + 8 * C + 12,
        # fused edge_grad backward + v_pix scatter: reads index, img, grad_out and bary; no per-pixel write
        "edge_grad_backward_fused": 4 + 8 * C + 12,
        "interpolate_backward_vpix": 28,
        "interpolate_backward": 4 * C + 16 + 12,
        "render_backward": 20,
    }


# op -> substrings of the HIP kernels it launches (names as rocprofv3 prints them)
OP_KERNELS = {
    "rasterize": ["bin_count_kernel", "bin_scan_kernel", "bin_fill_kernel", "tile_raster_kernel"],
    "render": ["render_kernel<"],
    "interpolate": ["interpolate_kernel<float, 4, 4>"],
    "edge_grad_backward": ["edge_dots_kernel", "edge_gather"],
    "edge_grad_backward_fused": ["edge_dots_kernel", "edge_scatter_pairs_kernel"],
    "interpolate_backward_vpix": ["interpolate_backward_kernel<float, true, false"],
    "interpolate_backward": ["interpolate_backward_wide_kernel<float>", "interpolate_backward_kernel<float, true, true, 4, 16>"],
    "render_backward": ["render_backward_kernel"],
}


def measured_traffic(op):
    """HBM bytes per launch of `op` from the newest committed PMC summary (profiles/rNN/traffic.json:
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes on the same kernels and workload,
    corrected as MI355X_MICROARCH.md prescribes).  None if no profile covers the op."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")))
    if not files or op not in OP_KERNELS:
        return None
    kernels = json.load(open(files[-1]))["kernels"]
    total, hit = 0, 0
    for pat in OP_KERNELS[op]:
        for name, rec in kernels.items():
            if pat in name:
                total += rec["hbm_bytes"]
                hit += 1
    return int(total) if hit else None


def time_kernels(v_pix, vi, attr, H, W, reps):
    """Per-kernel HIP-event timing through the C ABI on torch's current stream (the stream the
    kernels are launched on).  Returns {name: ms}."""
    from drtk_amd import capi

    def timed(fn):
        fn()
        th.cuda.synchronize()
        ev = [th.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        ev[0].record()
        for r in range(reps):
            fn()
            ev[r + 1].record()
        th.cuda.synchronize()
        return sum(ev[r].elapsed_time(ev[r + 1]) for r in range(reps)) / reps

    out = {}
    depth0, index = capi.rasterize(v_pix, vi, H, W)
    depth, bary = capi.render(v_pix, vi, index)
    img = capi.interpolate(attr, vi, index, bary)
    img = img * (index != -1)[:, None]
    g = th.Generator(device=v_pix.device).manual_seed(0)
    go = th.rand(img.shape, device=v_pix.device, generator=g) * 2 - 1
    gd = th.rand(depth.shape, device=v_pix.device, generator=g)
    gb = th.rand(bary.shape, device=v_pix.device, generator=g)
    ws_r = th.empty(capi.rasterize_workspace_bytes(v_pix.shape[0], vi.shape[0], H, W), dtype=th.uint8, device=v_pix.device)
    ws_e = th.empty(capi.edge_grad_backward_workspace_bytes(v_pix.dtype, v_pix.shape[0], H, W), dtype=th.uint8, device=v_pix.device)
    out["rasterize"] = timed(lambda: capi.rasterize(v_pix, vi, H, W, workspace=ws_r))
    out["render"] = timed(lambda: capi.render(v_pix, vi, index))
    out["interpolate"] = timed(lambda: capi.interpolate(attr, vi, index, bary))
    out["interpolate_vpix"] = timed(lambda: capi.interpolate(v_pix, vi, index, bary))
    out["edge_grad_backward"] = timed(lambda: capi.edge_grad_backward(v_pix, img, index, vi, go, workspace=ws_e))
    out["edge_grad_backward_fused"] = timed(lambda: capi.edge_grad_backward_fused(v_pix, img, index, vi, bary, go))
    eg = capi.edge_grad_backward(v_pix, img, index, vi, go)
    out["interpolate_backward_vpix"] = timed(lambda: capi.interpolate_backward(eg, v_pix, vi, index, bary, True, False))
    out["interpolate_backward"] = timed(lambda: capi.interpolate_backward(go, attr, vi, index, bary, True, True))
    out["render_backward"] = timed(lambda: capi.render_backward(v_pix, vi, index, gd, gb))
    return out


def _cpu_backend():
    """(kind, backend, threads): the reference's own CPU kernels if their prebuilt library travelled
    with the repo (oracle/_ref/libdrtk_ref_fast.so = /root/reference/src/*/*_kernel_cpu.cpp built with
    the reference's `-O3 --fast-math` flags by oracle/ref_build.py), else the CPU oracle port."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        from backends import RefBackend

        b = RefBackend("fast")
        return "reference", b, th.get_num_threads()
    except Exception:
        from backends import OracleBackend

        import oracle as O

        return "port", OracleBackend(nthreads=0), O.max_threads()


def cpu_baseline(v_pix, vi, attr, H, W, n_views, min_seconds=10.0):
    """fwd+bwd of the four ops on the host cores over `n_views` views of the same workload, with the
    reference's own CPU kernels (at::parallel_for over all cores) when available."""
    kind, B, cores = _cpu_backend()
    v = v_pix[:n_views].detach().cpu().contiguous()
    a = attr[:n_views].detach().cpu().contiguous()
    vi_c = vi.cpu()
    g = th.Generator().manual_seed(0)
    secs, passes = 0.0, 0
    while secs < min_seconds and passes < 64:
        t0 = time.perf_counter()
        depth0, index = B.rasterize(v, vi_c, H, W)
        depth, bary = B.render(v, vi_c, index)
        img = B.interpolate(a, vi_c, index, bary)
        vpix_img = B.interpolate(v, vi_c, index, bary)  # edge_grad_estimator's forward
        t_fwd = time.perf_counter()
        img = img * (index != -1)[:, None]
        if passes == 0:
            go = th.rand(img.shape, generator=g) * 2 - 1
            gd = th.rand(depth.shape, generator=g)
            gb = th.rand(bary.shape, generator=g)
        t1 = time.perf_counter()
        eg = B.edge_grad_backward(v, img, index, vi_c, go, 1e4)
        B.interpolate_backward(eg, v, vi_c, index, bary, True, False)
        B.interpolate_backward(go, a, vi_c, index, bary, True, True)
        B.render_backward(v, vi_c, index, gd, gb)
        t2 = time.perf_counter()
        secs += (t_fwd - t0) + (t2 - t1)
        passes += 1
    del vpix_img, depth0
    what = ("the reference's own CPU kernels (oracle/_ref, -O3 --fast-math, at::parallel_for)" if kind == "reference"
            else "the CPU oracle port (OpenMP)")
    return {
        "value": round(passes * n_views * H * W / secs / 1e6, 4),
        "unit": "Mpix/s",
        "cores": cores,
        "kind": kind,
        "sample": f"{n_views} of the {v_pix.shape[0]} views of the same workload, {passes} fwd+bwd passes of the four ops "
                  f"with {what}, {cores} threads, {secs:.2f} s of CPU work",
    }


class _MeanSquare(th.autograd.Function):
    """User-side loss term mean(x^2), written so that forward is ONE reduction pass and backward ONE
    scaling pass over x (plain `(x*x).mean()` or `vector_norm(x)**2/n` cost 2-4 extra passes over the
    2 GB image in eager PyTorch).  Same value, same gradient 2*x/n."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return th.linalg.vector_norm(x).square() / x.numel()

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return x * (g * (2.0 / x.numel()))


def main():
    args = parse()
    from drtk_amd import dist as ddist
    from drtk_amd import synthetic as S
    from drtk_amd.transform import transform

    rank, world, local_rank = ddist.init_from_env()
    assert th.cuda.is_available(), "bench.py needs a GPU (the HIP path is the product; there is no CPU fallback)"
    # DRTK_FORCE_DEVICE: test-only (several ranks on one GPU with DRTK_DIST_BACKEND=gloo)
    dev = th.device("cuda", int(os.environ.get("DRTK_FORCE_DEVICE", local_rank)))
    th.cuda.set_device(dev)
    import drtk_amd

    H = W = args.res
    C = args.channels
    n_local = args.views
    n_total = n_local * world
    nl, no = S.MESH_SIZES[args.mesh]
    v_world, vi = S.uv_sphere(nl, no, lobes=0.05, device=dev)
    campos, camrot, focal, princpt = S.ring_cameras(n_total, W, H, device=dev)
    mine = ddist.shard_views(n_total, rank, world)
    sl = slice(mine.start, mine.stop)
    campos, camrot, focal, princpt = campos[sl], camrot[sl], focal[sl], princpt[sl]
    attr = S.random_attributes(1, v_world.shape[0], C, seed=0, device=dev)[:1].contiguous()

    v_world = v_world.clone().requires_grad_(True)  # shared across views and ranks
    attr = attr.clone().requires_grad_(True)        # shared across views and ranks
    reducer = ddist.SharedGradReducer([v_world, attr])

    def step(fused_mask=False):
        v_pix = transform(v_world[None], campos, camrot, focal, princpt)  # shared [1,V,3] -> [n_local,V,3]
        a = attr.expand(n_local, -1, -1)
        index_img = drtk_amd.rasterize(v_pix, vi, H, W)
        depth_img, bary_img = drtk_amd.render(v_pix, vi, index_img)
        if fused_mask:
            # drtk_amd extension, NOT part of the headline number: interpolate + background mask in one op
            img = drtk_amd.interpolate_masked(a, vi, index_img, bary_img)
        else:
            img = drtk_amd.interpolate(a, vi, index_img, bary_img)
            # user-side shading and loss (plain PyTorch, timed inside the step): mask the background,
            # loss = mean(img^2) + mean(depth), written with the cheapest equivalent torch ops
            img = th.where((index_img != -1)[:, None], img, 0.0)
        img = drtk_amd.edge_grad_estimator(v_pix=v_pix, vi=vi, bary_img=bary_img, img=img, index_img=index_img)
        loss = _MeanSquare.apply(img) + depth_img.mean()
        loss.backward()
        reducer.all_reduce()
        v_world.grad = None
        attr.grad = None
        return loss

    for _ in range(args.warmup):
        step()
    ddist.barrier_and_sync(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    ddist.barrier_and_sync(dev)
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = th.tensor([elapsed], dtype=th.float64, device=dev)
        th.distributed.all_reduce(t, op=th.distributed.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    mpix = n_total * H * W * args.steps / elapsed / 1e6

    # the same step with the drtk_amd.interpolate_masked extension (reported beside, never as `value`)
    step(True)
    ddist.barrier_and_sync(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss_fused = step(True)
    ddist.barrier_and_sync(dev)
    elapsed_fused = time.perf_counter() - t0
    if world > 1:
        t = th.tensor([elapsed_fused], dtype=th.float64, device=dev)
        th.distributed.all_reduce(t, op=th.distributed.ReduceOp.MAX)
        elapsed_fused = float(t.item())

    result = None
    if rank == 0:
        with th.no_grad():
            v_pix = transform(v_world[None].expand(n_local, -1, -1), campos, camrot, focal, princpt).contiguous()
            a_full = attr.detach().expand(n_local, -1, -1).contiguous()
            kt = time_kernels(v_pix, vi, a_full, H, W, args.kernel_reps)
        P = n_local * H * W
        bpp = algorithmic_bytes_per_px(C)
        # ops the step actually launches (drtk_amd.edge_grad_estimator takes the fused route when no
        # hook is registered); the remaining entries time the reference-graph route for comparison
        in_step = ["rasterize", "render", "interpolate", "edge_grad_backward_fused", "interpolate_backward",
                   "render_backward"]
        dom = max(in_step, key=lambda k: kt[k])
        ach = bpp[dom] * P / (kt[dom] * 1e-3) / 1e9
        t_ops = sum(kt[k] for k in in_step)
        unfused_bpp = 164 + 16 * C  # the figure of the unfused operator boundary
        fused_bpp = sum(bpp[k] for k in in_step)
        roofline = {
            "bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4),
            "traffic": measured_traffic(dom) if (args.mesh, H, n_local, C) == ("100k", 2048, 8, 16) else None,
            "algorithmic_bytes": bpp[dom] * P,
            "bytes_per_px": bpp[dom], "ms_per_launch": round(kt[dom], 4),
            # the op is one C-ABI call = these HIP kernels back to back (names as rocprofv3 prints them in
            # profiles/rNN/bench_step_kernel_stats.txt; their average durations add up to ms_per_launch)
            "hip_kernels": OP_KERNELS[dom],
        }
        path = {
            "bytes_per_px": unfused_bpp, "bytes_per_px_fused_route": fused_bpp, "t_ops_ms": round(t_ops, 4),
            "achieved_GBps_ops": round(unfused_bpp * P / (t_ops * 1e-3) / 1e9, 1),
            "frac_ops": round(unfused_bpp * P / (t_ops * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "frac_ops_fused_bytes": round(fused_bpp * P / (t_ops * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "frac_step": round(unfused_bpp * P / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "ops_in_step": in_step,
            "kernels_ms": {k: round(x, 4) for k, x in kt.items()},
            "kernels_GBps": {k: round(bpp[k] * P / (kt[k] * 1e-3) / 1e9, 1) for k in kt},
        }
        cpu = None
        if world == 1 and args.cpu_sample_views > 0:
            cpu = cpu_baseline(v_pix, vi, a_full, H, W, min(args.cpu_sample_views, n_local))
        result = {
            "metric": "Mpixels/sec fwd+bwd, 100k-tri @ 2048x2048, 1/2/4/8 GPU; HBM BW %",
            "value": round(mpix, 2),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{n_local} views/GPU x {world} GPU, {args.mesh}-tri UV-sphere head mesh "
                            f"(F={vi.shape[0]}, V={v_world.shape[0]}), {H}x{W}, C={C} attribute channels, "
                            "transform+rasterize+render+interpolate+mask+edge_grad_estimator+loss fwd+bwd",
                "views_per_gpu": n_local, "triangles": int(vi.shape[0]), "vertices": int(v_world.shape[0]),
                "height": H, "width": W, "channels": C,
                "parallelism": f"views sharded {world}-way, one fused all-reduce of {reducer.nbytes()} B shared grads"
                               if world > 1 else "single GPU",
            },
            "loss": round(float(loss.detach()), 6),
            "extensions": {
                "interpolate_masked": {
                    "note": "same step with drtk_amd.interpolate_masked replacing interpolate + torch.where "
                            "(identical loss and gradients); an opt-in extension, not the reference API, hence "
                            "not the headline value",
                    "value": round(n_total * H * W * args.steps / elapsed_fused / 1e6, 2),
                    "ms_per_step": round(elapsed_fused / args.steps * 1e3, 4),
                    "loss": round(float(loss_fused.detach()), 6),
                },
            },
            "roofline": roofline,
            "path_roofline": path,
            "cpu_baseline": cpu,
        }
        print(json.dumps(result), flush=True)
    if world > 1:
        th.distributed.barrier()
        th.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

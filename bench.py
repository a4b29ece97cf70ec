#!/usr/bin/env python3
"""Headline benchmark: fwd+bwd Mpixels/s of rasterize -> render -> interpolate -> edge_grad.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (default = `--config 3` = BASELINE.json configs[2], the one the metric is quoted on): per GPU 8 camera views
of a 100 352-triangle "head" UV sphere at 2048x2048, 16 attribute channels, float32.  One step =
transform -> rasterize -> render -> interpolate(C=16) -> mask -> edge_grad_estimator ->
loss = mean(img^2) + mean(depth) -> backward, producing gradients for the SHARED world-space
vertices [V,3] and SHARED attributes [1,V,C]; with N > 1 ranks every rank renders its own 8 views
(weak scaling) and the shared gradients are summed over RCCL, each shared tensor's all-reduce launched on a side
stream the moment its gradient is final (drtk_amd/dist.py).  All inputs are resident in HBM before the timed region.

Other configurations (parity-test cases of BASELINE.json, measured for profiles/rNN/other_configs/, never the
headline): `--config 2` 4 views, 10k triangles, 512^2, C=3; `--config 4` 8 views/GPU, 250k triangles, 2048^2, C=16
(BASELINE configs[3]: 64 views over 8 GPUs); `--config 5` = `--workload textured`: 1M triangles, 4096^2, uv
interpolate -> screen_space_uv_derivative -> mipmap_grid_sample -> mask -> edge_grad_estimator with the uv attributes
and the texture pyramid stored in fp16 under autocast (BASELINE configs[4]).  (Numbering as in SURVEY.md 8: config k =
BASELINE.json configs[k-1].)

Rank 0 prints ONE JSON line.  `value` = all ranks' pixels / max-over-ranks wall time of the K timed steps.
`roofline` prices the single HIP kernel with the largest average duration: SURVEY.md 8d's algorithmic bytes of the
tensors that kernel streams / its HIP-event time, measured live through the library's per-kernel timing
(drtk_amd_kernel_timing_*: a pair of HIP events around every launch, on the stream it is launched on) over extra
steps of the same workload right after the timed region; `path_roofline` holds the per-op and whole-path figures;
`cpu_baseline` times the reference's own CPU kernels (oracle/_ref) or the CPU oracle on this host's cores on a
bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this platform needs dmabuf IPC (the image exports it already; kept here so that a bare launcher
# environment works too) -- must be in place before the HIP runtime initialises
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch as th

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable

CONFIGS = {  # SURVEY.md 8 numbering; config k = BASELINE.json configs[k-1]
    2: dict(workload="mesh", mesh="10k", res=512, views=4, channels=3),
    3: dict(workload="mesh", mesh="100k", res=2048, views=8, channels=16),
    4: dict(workload="mesh", mesh="250k", res=2048, views=8, channels=16),
    5: dict(workload="textured", mesh="1M", res=4096, views=2, channels=3),
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS), help="BASELINE.json configs[k-1]; 3 = the headline")
    ap.add_argument("--workload", default=None, choices=["mesh", "textured"])
    ap.add_argument("--views", type=int, default=None, help="views per GPU")
    ap.add_argument("--mesh", default=None, choices=["10k", "100k", "250k", "1M"])
    ap.add_argument("--res", type=int, default=None)
    ap.add_argument("--channels", type=int, default=None)
    ap.add_argument("--tex", type=int, default=4096, help="textured workload: texture size")
    ap.add_argument("--cpu-sample-views", type=int, default=None, help="views timed on the CPU (0 = skip)")
    ap.add_argument("--kernel-steps", type=int, default=5, help="extra steps run under the per-kernel HIP-event timing")
    ap.add_argument("--no-graph", action="store_true", help="skip the captured-graph timing of the same step")
    ap.add_argument("--reduce-after-backward", action="store_true",
                    help="N > 1: ONE all-reduce of all shared gradients after the backward pass instead of one per shared tensor launched from inside it (A/B of the overlap)")
    ap.add_argument("--grad-reset", default="auto", choices=["auto", "set_to_none", "flat_zero"],
                    help="how a step clears the shared gradients: set_to_none (a one-GPU loop's optimizer.zero_grad(); the default without a "
                         "process group) or flat_zero (zero the reducer's flat buffer and accumulate into it; what every rank of a group does). "
                         "Recorded in config.grad_reset; at N = 1 the line also carries the other mode's time (extensions.grad_reset_flat_zero)")
    ap.add_argument("--graph-child", action="store_true", help=argparse.SUPPRESS)  # internal: this process only captures + replays the step
    a = ap.parse_args()
    if a.workload == "textured" and a.config == 3:
        a.config = 5
    for k, v in CONFIGS[a.config].items():
        if getattr(a, k) is None:
            setattr(a, k, v)
    if a.cpu_sample_views is None:
        a.cpu_sample_views = 2 if a.workload == "mesh" else 1
    return a


# ---- algorithmic bytes per pixel (f32), per op and per HIP kernel ----------------------------------------
def op_bytes_per_px(C):
    return {
        "rasterize": 8,
        "render": 4 + 16,
        "interpolate": 16 + 4 * C,
        # fused edge_grad backward + v_pix scatter: reads index, img, grad_out and bary; no per-pixel write
        "edge_grad_backward_fused": 4 + 8 * C + 12,
        "interpolate_backward": 4 * C + 16 + 12,
        "render_backward": 20,
    }


# HIP kernel (name as spelled at its launch site = prefix of what rocprofv3 prints) -> (op it belongs to, the bytes/px
# of SURVEY 8d's per-pixel tensors that THIS kernel streams).  Kernels with 0 move per-triangle / per-vertex tables or
# zero-fill a gradient (counted with the op's write, 8d's accounting rule).
def kernel_table(C):
    return {
        "bin_count_kernel": ("rasterize", 0),
        "bin_scan_kernel": ("rasterize", 0),
        "bin_fill_kernel": ("rasterize", 0),
        "tile_raster_kernel": ("rasterize", 8),                        # index 4 + depth 4 written
        "render_kernel": ("render", 20),                               # index 4 read, depth 4 + bary 12 written
        "interpolate_kernel": ("interpolate", 16 + 4 * C),             # index 4 + bary 12 read, C planes written
        "edge_dots_kernel": ("edge_grad_backward_fused", 4 + 8 * C),   # index 4, img 4C, grad_out 4C read
        "edge_scatter_pairs_kernel": ("edge_grad_backward_fused", 16), # index 4 + bary 12 read
        "interpolate_backward_wide_kernel": ("interpolate_backward", 4 * C + 28),   # grad_out 4C, index, bary read; bary_grad 12 written
        "interpolate_backward_kernel": ("interpolate_backward", 4 * C + 28),
        "interpolate_backward_small_kernel": ("interpolate_backward", 4 * C + 28),   # C <= 4
        "render_backward_kernel": ("render_backward", 20),             # index 4, grad_depth 4, grad_bary 12 read
        "fill_bytes_kernel": ("zero-fill of gradients / workspaces", 0),
    }


TEXTURED_KERNELS = {  # additional kernels of the textured workload: (stage, bytes/px of per-pixel tensors streamed)
    "uv_derivative_kernel": ("screen_space_uv_derivative", 16 + 16),   # index 4 + bary 12 read, Jacobian 16 written
    "mipmap_forward_kernel": ("mipmap_grid_sample", 8 + 16 + 12),      # grid 8 + Jacobian 16 read, RGB 12 written (+ texel gathers)
    "mipmap_forward_lean_kernel": ("mipmap_grid_sample", 8 + 16 + 12),
    "mipmap_backward_tiled2_kernel": ("mipmap_grid_sample backward", 12 + 8 + 16 + 8),
    "mipmap_backward_lean_kernel": ("mipmap_grid_sample backward", 12 + 8 + 16 + 8),
    "mipmap_backward_kernel": ("mipmap_grid_sample backward", 12 + 8 + 16 + 8),
    "mipmap_backward_wave_kernel": ("mipmap_grid_sample backward", 12 + 8 + 16 + 8),  # (C > 4; the textured workload's RGB takes the tiled one)
}


def site_name(site):
    """launch-site spelling '(edge_dots_kernel<T, 4, kStripRows, kDotsWaves>' -> 'edge_dots_kernel'"""
    return site.strip("()").split("<")[0].strip()


def measured_traffic_table():
    """{kernel: HBM bytes per launch} from the newest COMMITTED PMC collection (profiles/rNN/traffic.json: rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE in separate passes over profiles/kernel_bench.py = the same kernels on the headline
    shape, corrected as MI355X_MICROARCH.md prescribes), and the file it came from.  Builder-side evidence replayed into
    the line: bench.py does not run the profiler.  ({}, None) if there is no collection."""
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")))
    if not files:
        return {}, None
    out = {}
    doc = json.load(open(files[-1]))
    for name, rec in doc["kernels"].items():
        out[name.split("(")[0].split("<")[0].split()[-1]] = int(rec["hbm_bytes"])
    measured_traffic_table.sources = doc.get("sources")  # {csrc file: sha256} at collection time (profiles/make_traffic.py)
    return out, os.path.relpath(files[-1], ROOT)


def traffic_staleness(kernel):
    """Has the source of `kernel` changed since the committed PMC collection was taken?  (stale, [files that differ]):
    the collection records the SHA-256 of every file under drtk_amd/csrc; compared are the file that defines the kernel
    and the shared headers.  A collection without that record (rounds 1-4) counts as stale: provenance unknown."""
    import hashlib

    sources = getattr(measured_traffic_table, "sources", None)
    if not sources:
        return True, ["(the collection records no source hashes)"]
    csrc = os.path.join(ROOT, "drtk_amd", "csrc")
    mine = [f for f in sorted(os.listdir(csrc)) if f.endswith(".hpp")]
    for f in sorted(os.listdir(csrc)):
        if f.endswith(".hip") and (" " + kernel + "(") in open(os.path.join(csrc, f)).read():
            mine.append(f)
    changed = [f for f in mine if sources.get(f) != hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest()]
    return bool(changed), changed


def measured_traffic(kernel):
    return measured_traffic_table()[0].get(kernel)


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


def cpu_baseline_textured(v_world, v_pix, vi, vt, vti, tex, cams, H, W, min_seconds=8.0):
    """fwd+bwd of the textured pipeline on ONE view with the CPU oracle (the sampler and the uv Jacobian have no CPU
    implementation in the reference: `kind` is "port" for this workload)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as O

    cores = O.max_threads()
    c = lambda t: t[:1].detach().float().cpu().contiguous()  # noqa: E731
    v, vw, vtc = c(v_pix), v_world.detach().cpu()[None], c(vt)
    vi_c, vti_c = vi.cpu(), vti.cpu()
    texc = [c(t) for t in tex]
    campos, camrot, focal = (c(t) for t in cams[:3])
    g = th.Generator().manual_seed(0)
    secs, passes = 0.0, 0
    while secs < min_seconds and passes < 16:
        t0 = time.perf_counter()
        _, index = O.rasterize(v, vi_c, H, W, nthreads=0)
        depth, bary = O.render(v, vi_c, index, nthreads=0)
        uv = O.interpolate(vtc, vti_c, index, bary, nthreads=0)
        mask = index != -1
        jac = O.screen_space_uv_derivative(vw, vtc, vi_c, vti_c, index, bary, mask, campos, camrot, focal)
        grid = ((uv.permute(0, 2, 3, 1) * 2 - 1) * mask[..., None]).contiguous()
        img = O.mipmap_grid_sampler_2d(texc, grid, jac, 8, 1, 0) * mask[:, None]
        O.interpolate(v, vi_c, index, bary, nthreads=0)  # edge_grad_estimator's forward
        if passes == 0:
            go = th.rand(img.shape, generator=g) * 2 - 1
            gd = th.rand(depth.shape, generator=g)
        eg = O.edge_grad_backward(v, img, index, vi_c, go, 1e4, nthreads=0)
        O.interpolate_backward(eg, v, vi_c, index, bary, True, False, nthreads=0)
        _, gg = O.mipmap_grid_sampler_2d_backward(go * mask[:, None], texc, grid, jac, 8, 1, 0)
        guv = (gg * 2 * mask[..., None]).permute(0, 3, 1, 2).contiguous()
        _, gb = O.interpolate_backward(guv, vtc, vti_c, index, bary, True, True, nthreads=0)
        O.render_backward(v, vi_c, index, gd, gb, nthreads=0)
        secs += time.perf_counter() - t0
        passes += 1
    return {
        "value": round(passes * H * W / secs / 1e6, 4), "unit": "Mpix/s", "cores": cores, "kind": "port",
        "sample": f"1 of the views of the same workload, {passes} fwd+bwd passes of the textured pipeline with the CPU oracle "
                  f"(four ops + restated sampler and uv Jacobian; the latter in PyTorch on the host), {cores} threads, {secs:.2f} s of CPU work",
    }


class _MeanSquare(th.autograd.Function):
    """User-side loss term mean(x^2), written so that forward is ONE reduction pass and backward ONE
    scaling pass over x (plain `(x*x).mean()` costs 2-4 extra passes over the 2 GB image in eager PyTorch).
    Same value, same gradient 2*x/n.  The reduction runs per image row first (`vector_norm(..., dim=1)`) and then over
    the rows' squares: ATen's reduction of a [rows, W] tensor along dim 1 streams at 6.2 TB/s, its all-elements reduction
    of the same tensor at 3.9 (profiles/glue_norm_bench.py: 0.349 vs 0.547 ms for 2.1 GB)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        rows = th.linalg.vector_norm(x.reshape(-1, x.shape[-1]), dim=1)
        return th.linalg.vector_norm(rows).square() / x.numel()

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return x * (g * (2.0 / x.numel()))


_JSON_FD = 1  # main() replaces it by a private duplicate of the real stdout


def emit_line(obj):
    """The one JSON line of this process, on the REAL stdout (see main(): descriptor 1 itself is pointed at stderr)."""
    sys.stdout.flush()
    os.write(_JSON_FD, (json.dumps(obj) + "\n").encode())


def timed_loop(step, steps, dev, world):
    """The contract's timing: barrier + synchronize, EXACTLY `steps` steps, barrier + synchronize, wall clock, MAX over
    ranks.  Beside it every step lies between two HIP events on the stream the step runs on (hipEvent,
    median): the median tells what a step costs when nothing disturbs it, the wall clock is what `value` is made of."""
    from drtk_amd import dist as ddist

    evs = [th.cuda.Event(enable_timing=True) for _ in range(steps + 1)]  # one per step BOUNDARY (an event record is a barrier packet)
    ddist.barrier_and_sync(dev)
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(steps):
        loss = step()
        evs[i + 1].record()
    ddist.barrier_and_sync(dev)
    elapsed = time.perf_counter() - t0
    per_step = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
    median_ms = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    if world > 1:
        t = th.tensor([elapsed, median_ms], dtype=th.float64, device=dev)
        th.distributed.all_reduce(t, op=th.distributed.ReduceOp.MAX)
        elapsed, median_ms = float(t[0].item()), float(t[1].item())
    timed_loop.last_median_ms = median_ms
    return elapsed, loss


def graph_child(step, leaves, steps, pixels):
    """torch's whole-network capture recipe: warm up on a side stream, gradients None so that the captured backward
    allocates them in the graph's pool, capture ONE step, replay it `steps` times."""
    for p in leaves:
        p.grad = None
    s = th.cuda.Stream()
    s.wait_stream(th.cuda.current_stream())
    with th.cuda.stream(s):
        for _ in range(3):
            step(reduce=False)
    th.cuda.current_stream().wait_stream(s)
    th.cuda.synchronize()
    for p in leaves:
        p.grad = None
    g = th.cuda.CUDAGraph()
    with th.cuda.graph(g):
        g_loss = step(reduce=False)
    g.replay()
    th.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    th.cuda.synchronize()
    el = time.perf_counter() - t0
    emit_line({
        "value": round(pixels * steps / el / 1e6, 2), "ms_per_step": round(el / steps * 1e3, 4), "loss": round(float(g_loss.detach()), 6),
        "note": "the identical step captured with torch.cuda.graph and replayed (same kernels, no host launch / autograd bookkeeping "
                "between them), in a child process; reported beside the eager headline, never as `value`"})


def main():
    args = parse()
    # The contract is ONE JSON line on stdout.  Libraries write there too -- RCCL 2.26 prints a five-line version banner to
    # file descriptor 1 when its communicator is created, on every rank -- so descriptor 1 is pointed at stderr for the
    # whole run and the line goes out through a private duplicate of the real stdout.
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    from drtk_amd import dist as ddist
    from drtk_amd import synthetic as S
    from drtk_amd.transform import transform

    # DRTK_SINGLE_RANK_GROUP: test-only -- a ONE-rank process group, so that the multi-rank code path (reducers, the max over
    # ranks, the all_reduce block of the line) runs against RCCL on a one-GPU box (tests/test_gpu_bench_contract.py)
    single_rank_group = os.environ.get("DRTK_SINGLE_RANK_GROUP") == "1"
    rank, world, local_rank = ddist.init_from_env(single_rank_group=single_rank_group)
    grouped = world > 1 or single_rank_group
    if args.gpus != world:
        # Never re-exec or spawn from here: say how to launch instead (a mislabelled n_gpus would poison a scaling curve).
        raise SystemExit(
            f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s).  One process per GPU: launch with\n"
            f"  python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port 29500 "
            f"bench.py --gpus {args.gpus} --steps {args.steps} --warmup {args.warmup}")
    assert th.cuda.is_available(), "bench.py needs a GPU (the HIP path is the product; there is no CPU fallback)"
    # DRTK_FORCE_DEVICE: test-only (several ranks on one GPU with DRTK_DIST_BACKEND=gloo)
    dev = th.device("cuda", int(os.environ.get("DRTK_FORCE_DEVICE", local_rank)))
    th.cuda.set_device(dev)
    import drtk_amd
    from drtk_amd import capi

    textured = args.workload == "textured"
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
    v_world = v_world.clone().requires_grad_(True)  # shared across views and ranks

    if textured:
        vt, vti = S.uv_sphere_atlas(nl, no, device=dev)
        vt = vt[None].half().requires_grad_(True)  # uv attributes stored in fp16, shared across views and ranks
        tex = [t.half().requires_grad_(True) for t in S.texture_pyramid(1, 3, args.tex, device=dev)]  # shared fp16 texture pyramid
        # two collectives per step: the fp16 leaves (uv attributes + every mip level) as ONE group staged in float32 --
        # their gradients are accumulated, summed over the ranks and only then rounded to fp16 -- and the vertices
        reducers = [ddist.SharedGradReducer([v_world, [vt] + tex], dtype=th.float32, overlap=not args.reduce_after_backward)]
    else:
        attr = S.random_attributes(1, v_world.shape[0], C, seed=0, device=dev)[:1].contiguous()
        attr = attr.clone().requires_grad_(True)        # shared across views and ranks
        reducers = [ddist.SharedGradReducer([v_world, attr], overlap=not args.reduce_after_backward)]

    for r in reducers:
        r.run_single_rank = single_rank_group
    leaves = [p for r in reducers for p in r.params]

    grad_reset = args.grad_reset if args.grad_reset != "auto" else ("flat_zero" if grouped else "set_to_none")

    def step(fused_mask=False, reduce=True, flat_zero=None):
        for r in reducers:
            r.enabled = reduce  # off: this rank steps alone (no collective may be entered, not even from a gradient hook)
        if flat_zero is None:
            flat_zero = (reduce and grouped) or (grad_reset == "flat_zero" and reduce)
        if flat_zero:
            for r in reducers:
                r.zero_grad()  # gradients accumulate straight into the reducer's flat buffer
        else:
            # one process, no group: what a single-GPU loop does (optimizer.zero_grad(set_to_none=True), PyTorch's default);
            # also the graph-capture recipe: gradients must be allocated by the captured backward pass itself
            for p in leaves:
                p.grad = None
        v_pix = transform(v_world[None], campos, camrot, focal, princpt)  # shared [1,V,3] -> [n_local,V,3]
        if textured:
            with th.autocast("cuda", dtype=th.float16):
                up = reducers[0].upcast  # the fp16 leaves enter the pipeline as float32 (what autocast would do inside the ops)
                out = S.textured_shading(drtk_amd, v_world[None].expand(n_local, -1, -1), v_pix, vi, up(vt).expand(n_local, -1, -1), vti,
                                         [up(t).expand(n_local, -1, -1, -1) for t in tex], campos, camrot, focal, H, W)
            img, depth_img = out["img"], out["depth_img"]
        else:
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
        if reduce:
            for r in reducers:
                r.finish()
        return loss

    if args.graph_child:
        return graph_child(step, leaves, args.steps, n_total * H * W)
    for _ in range(args.warmup):
        step()
    elapsed, loss = timed_loop(step, args.steps, dev, 2 if grouped else 1)
    median_step_ms = timed_loop.last_median_ms
    ms_per_step = elapsed / args.steps * 1e3
    mpix = n_total * H * W * args.steps / elapsed / 1e6
    comm = None
    if grouped:
        # the collectives' own times come from ONE extra step with the reducers' event bracketing on (every rank runs it);
        # the timed region above runs without those events
        for r in reducers:
            r.record_timings = True
        step()
        for r in reducers:
            r.record_timings = False
        per = [r.timings_ms() for r in reducers]
        if all(t is not None for t in per):
            comm = (sum(t[0] for t in per), sum(t[1] for t in per))
    loss_value = float(loss.detach())

    # N = 1 without a group clears gradients the one-GPU way (set to None); the ranks of a group zero the reducer's flat
    # buffer and accumulate into it.  So that a scaling curve can be read like for like, the one-GPU line also carries the
    # SAME step timed the group's way (no collective: there is no group)
    flat_zero_alt = None
    if not grouped and grad_reset == "set_to_none":
        step(flat_zero=True)
        el_fz, _ = timed_loop(lambda: step(flat_zero=True), args.steps, dev, world)
        flat_zero_alt = {"note": "the same step with the shared gradients zeroed in the reducer's flat buffer and accumulated into it -- what "
                                 "every rank of an N > 1 run does (config.grad_reset there) -- for a like-for-like N = 1 point of the scaling curve",
                         "value": round(n_total * H * W * args.steps / el_fz / 1e6, 2), "ms_per_step": round(el_fz / args.steps * 1e3, 4)}
        step()  # leave the leaves in the headline's mode

    # the same step with the drtk_amd.interpolate_masked extension (reported beside, never as `value`)
    ext = None
    if not textured:
        step(True)
        elapsed_fused, loss_fused = timed_loop(lambda: step(True), args.steps, dev, world)
        ext = {"interpolate_masked": {
            "note": "same step with drtk_amd.interpolate_masked replacing interpolate + torch.where (identical loss and "
                    "gradients); an opt-in extension, not the reference API, hence not the headline value",
            "value": round(n_total * H * W * args.steps / elapsed_fused / 1e6, 2),
            "ms_per_step": round(elapsed_fused / args.steps * 1e3, 4), "loss": round(float(loss_fused.detach()), 6)}}

    # the operators alone (reference API, wall clock): the same forward, and a backward pass seeded with RESIDENT upstream
    # gradients instead of the user-side mask and loss -- what the step costs without the PyTorch glue around the path
    if not textured and world == 1:
        gsrc = th.Generator(device=dev).manual_seed(1)
        with th.no_grad():
            v_pix0 = transform(v_world[None], campos, camrot, focal, princpt)
            fg = (drtk_amd.rasterize(v_pix0, vi, H, W) != -1)[:, None]
        g_img = (th.rand(n_local, C, H, W, device=dev, generator=gsrc) * 2 - 1) * fg  # zero on the background, like a masked loss's
        g_depth = th.full((n_local, H, W), 1.0 / (n_local * H * W), device=dev)
        del v_pix0, fg

        def ops_step():
            for r in reducers:
                r.enabled = False
            for p in leaves:
                p.grad = None
            v_pix = transform(v_world[None], campos, camrot, focal, princpt)
            a = attr.expand(n_local, -1, -1)
            index_img = drtk_amd.rasterize(v_pix, vi, H, W)
            depth_img, bary_img = drtk_amd.render(v_pix, vi, index_img)
            img = drtk_amd.interpolate(a, vi, index_img, bary_img)
            img = drtk_amd.edge_grad_estimator(v_pix=v_pix, vi=vi, bary_img=bary_img, img=img, index_img=index_img)
            th.autograd.backward([img, depth_img], [g_img, g_depth])
            return depth_img

        ops_step()
        elapsed_ops, _ = timed_loop(ops_step, args.steps, dev, world)
        ext["operators_only"] = {
            "note": "transform + rasterize + render + interpolate + edge_grad_estimator forward, and their backward pass seeded with "
                    "resident upstream gradients (random on the foreground, zero on the background): the step without the user-side "
                    "torch.where mask and loss, wall clock; reported beside, never as `value`",
            "value": round(n_total * H * W * args.steps / elapsed_ops / 1e6, 2),
            "ms_per_step": round(elapsed_ops / args.steps * 1e3, 4)}
        del g_img, g_depth

    # the same step (reference API) captured once as a HIP graph and replayed: what the launch / autograd overhead costs.
    # Measured in a fresh CHILD process (started here, never exec'ed over this one) while this process idles: a capture
    # that goes wrong takes down the process it runs in, and the headline must not depend on it.
    graph = None
    if world == 1 and not grouped and not args.no_graph:
        import subprocess

        th.cuda.synchronize()
        cmd = [sys.executable, os.path.abspath(__file__), "--graph-child", "--steps", str(args.steps), "--config", str(args.config),
               "--workload", args.workload, "--mesh", args.mesh, "--res", str(args.res), "--views", str(args.views),
               "--channels", str(args.channels), "--tex", str(args.tex)]
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            graph = json.loads(lines[-1]) if r.returncode == 0 and lines else {"error": f"child exited {r.returncode}: {r.stderr[-300:]}"}
        except Exception as e:
            graph = {"error": f"{type(e).__name__}: {e}"[:300]}

    result = None
    if rank == 0:
        # ---- per-kernel HIP-event timing of the real step (library kernels only; PyTorch glue is t_step - t_ops).
        # Rank 0 alone runs these steps, so with several ranks they must not enter a collective.
        alone = world == 1
        step(reduce=alone)
        th.cuda.synchronize()
        capi.kernel_timing_begin()
        for _ in range(args.kernel_steps):
            step(reduce=alone)
        th.cuda.synchronize()
        sites = capi.kernel_timing_report()
        table = dict(kernel_table(C), **(TEXTURED_KERNELS if textured else {}))
        P = n_local * H * W
        kernels = {}
        unknown = [site_name(x) for x in sites if site_name(x) not in table and not site_name(x).startswith("transform")]
        assert not unknown, f"bench.py's kernel table does not know {unknown}: their time would be missing from t_ops"
        for site, (count, total_ms) in sites.items():
            k = site_name(site)
            rec = kernels.setdefault(k, {"launches_per_step": 0.0, "ms_per_step": 0.0})
            rec["launches_per_step"] += count / args.kernel_steps
            rec["ms_per_step"] += total_ms / args.kernel_steps
        # Bytes that MOVE beside SURVEY 8d's algorithmic figure: two kernels skip the background (edge_dots: a lane whose
        # pixels and halo are all background neither loads img / grad_out nor stores; interpolate_backward_wide: a wave
        # whose 64 x 4 pixels are all background only zeroes its bary_grad), so on 8d's bytes they can come out above the
        # HBM peak -- a figure > peak says "not streaming what it is priced with", not "fast".  `bytes_per_px_moved`
        # weights the skippable tensors with the share of the image those kernels really touch (measured on this
        # step's own index_img), `GBps_moved` prices the kernel with it, `GBps_traffic` with the PMC collection.
        moved = {}
        if not textured:
            with th.no_grad():
                idx = drtk_amd.rasterize(transform(v_world[None], campos, camrot, focal, princpt), vi, H, W)
                fg = (idx != -1)[:, None].float()
                share_pix = float(fg.mean())
                tiles = th.nn.functional.max_pool2d(fg, (4, 64), ceil_mode=True)        # a wave's 64 x 4 pixels
                share_wave_tiles = float(tiles.mean())
                near = th.nn.functional.max_pool2d(th.nn.functional.pad(fg, (1, 4, 0, 1)), (2, 9), stride=(1, 4))  # a lane's 4 pixels, their left/right neighbours, the row below
                share_dots_lanes = float(near.mean())
                del idx, fg, tiles, near
            moved = {
                "edge_dots_kernel": 4 + 8 * C * share_dots_lanes + 8 * share_dots_lanes,   # index everywhere; img, grad_out and the two pair planes where a lane has foreground near it
                "interpolate_backward_wide_kernel": 4 + 12 + (4 * C + 12) * share_wave_tiles,  # index read + bary_grad written everywhere; grad_out + bary where the wave has foreground
            }
        traffic_table, traffic_file = measured_traffic_table()
        headline_shape = args.config == 3 and (args.mesh, H, n_local, C) == ("100k", 2048, 8, 16)
        if not headline_shape:
            traffic_table, traffic_file = {}, None
        for k, rec in kernels.items():
            op, bpp = table.get(k, ("outside the four ops", 0))
            rec["op"] = op
            rec["ms_per_launch"] = rec["ms_per_step"] / max(rec["launches_per_step"], 1e-9)
            rec["bytes_per_px"] = bpp
            if bpp:
                per_s = P * rec["launches_per_step"] / (rec["ms_per_step"] * 1e-3) / 1e9
                rec["GBps"] = round(bpp * per_s, 1)
                rec["bytes_per_px_moved"] = round(moved.get(k, bpp), 2)
                rec["GBps_moved"] = round(moved.get(k, bpp) * per_s, 1)
            if k in traffic_table:
                rec["traffic_bytes_per_launch"] = traffic_table[k]
                rec["GBps_traffic"] = round(traffic_table[k] / (rec["ms_per_launch"] * 1e-3) / 1e9, 1)
            rec["ms_per_step"], rec["ms_per_launch"] = round(rec["ms_per_step"], 4), round(rec["ms_per_launch"], 4)
            rec["launches_per_step"] = round(rec["launches_per_step"], 2)
        priced = {k: r for k, r in kernels.items() if r["bytes_per_px"]}
        # the dominant kernel: the largest time per launch; among kernels within 3 % of it (interpolate backward and edge_dots
        # tie on the headline shape and swap places from run to run) the one FURTHEST below the roofline -- the conservative
        # figure, and the same kernel in every run
        t_max = max(r["ms_per_launch"] for r in priced.values())
        dom = min((k for k, r in priced.items() if r["ms_per_launch"] >= 0.97 * t_max), key=lambda k: priced[k]["GBps"])
        d = priced[dom]
        alg = d["bytes_per_px"] * P
        ach = alg / (d["ms_per_launch"] * 1e-3) / 1e9
        traffic = traffic_table.get(dom)
        stale, stale_files = traffic_staleness(dom) if traffic else (None, [])
        roofline = {
            "bound": "hbm", "kernel": dom, "op": d["op"], "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
            "frac_traffic": round(traffic / (d["ms_per_launch"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
            # the kernel's source (or a shared header) differs from what the PMC collection was taken on: `traffic` describes
            # an older kernel -- re-collect (profiles/scripts/collect_r05.sh) before reading it
            "traffic_stale": stale, "traffic_sources_changed": stale_files,
            "achieved_moved": d["GBps_moved"], "frac_moved": round(d["GBps_moved"] / HBM_PEAK_GBS, 4),
            "algorithmic_bytes": alg, "bytes_per_px": d["bytes_per_px"], "bytes_per_px_moved": d["bytes_per_px_moved"],
            "ms_per_launch": d["ms_per_launch"], "launches_per_step": d["launches_per_step"],
            "how": "the kernel with the largest time per launch (of those within 3 % of it, the one furthest below the roofline); "
                   f"HIP events around every launch of the kernel on its launch stream, {args.kernel_steps} steps of the same workload "
                   "right after the timed region (drtk_amd_kernel_timing_*); `achieved` = SURVEY 8d's per-pixel tensors this kernel streams / "
                   "that time; `achieved_moved` = the same with the tensors it skips on the background weighted by the share it touches; "
                   + (f"`traffic` = HBM bytes per launch of this kernel from the committed PMC collection {traffic_file} (rocprofv3 --pmc "
                      "FETCH_SIZE / WRITE_SIZE over profiles/kernel_bench.py on this shape; NOT collected by this run), priced with "
                      "this run's time" if traffic else "`traffic` = null: no PMC collection covers this shape"),
        }
        ops_ms = {}
        for k, r in kernels.items():
            if r["op"] not in ("outside the four ops",):
                ops_ms[r["op"]] = round(ops_ms.get(r["op"], 0.0) + r["ms_per_step"], 4)
        path_kernels = [k for k, r in kernels.items() if k in table]
        t_ops = sum(kernels[k]["ms_per_step"] for k in path_kernels)
        if textured:
            # rasterize 8 + render 20 + interpolate(uv, C=2) 24 + uv Jacobian 32 + sampler 36 + edge fwd (interpolate C=3) 28 +
            # edge bwd (C=3) 4+24+12 + interpolate bwd C=3 28 + sampler bwd 44 + interpolate bwd (uv) 36 + render bwd 20
            unfused_bpp, fused_bpp = 8 + 20 + 24 + 32 + 36 + 28 + 40 + 28 + 44 + 36 + 20, 8 + 20 + 24 + 32 + 36 + (4 + 24 + 12) + 44 + 36 + 20
        else:
            unfused_bpp = 164 + 16 * C  # the figure of the unfused operator boundary
            fused_bpp = sum(op_bytes_per_px(C).values())
        moved_bytes = sum(kernels[k].get("bytes_per_px_moved", 0) * kernels[k]["launches_per_step"] for k in path_kernels)
        covered = [k for k in path_kernels if k in traffic_table]
        traffic_bytes = sum(traffic_table[k] * kernels[k]["launches_per_step"] for k in covered)
        traffic_ms = sum(kernels[k]["ms_per_step"] for k in covered)
        path = {
            "bytes_per_px": unfused_bpp, "bytes_per_px_fused_route": fused_bpp, "t_ops_ms": round(t_ops, 4),
            "achieved_GBps_ops": round(unfused_bpp * P / (t_ops * 1e-3) / 1e9, 1),
            "frac_ops": round(unfused_bpp * P / (t_ops * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "frac_ops_fused_bytes": round(fused_bpp * P / (t_ops * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "frac_ops_moved": round(moved_bytes * P / (t_ops * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "frac_ops_traffic": round(traffic_bytes / (t_ops * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if covered else None,
            "frac_step": round(unfused_bpp * P / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "note": "frac_ops: SURVEY 8d's 164 + 16 C bytes per pixel (every per-pixel tensor of the UNFUSED operator boundary) / t_ops / "
                    "8 TB/s -- the figure the target is stated in; frac_ops_fused_bytes: the tensors the fused edge route really needs; "
                    "frac_ops_moved: additionally without the background the kernels skip; frac_ops_traffic: HBM bytes the PMC collection "
                    + (f"{traffic_file} counted for these kernels (covering {round(traffic_ms / max(t_ops, 1e-9) * 100)} % of t_ops)" if covered else "(none for this shape)")
                    + " over the same t_ops",
            "ops_ms": ops_ms,
            "kernels": kernels,
        }
        cpu = None
        if world == 1 and args.cpu_sample_views > 0:
            with th.no_grad():
                v_pix = transform(v_world[None].expand(n_local, -1, -1), campos, camrot, focal, princpt).contiguous()
            if textured:
                cpu = cpu_baseline_textured(v_world, v_pix, vi, vt, vti, tex, (campos, camrot, focal), H, W)
            else:
                a_full = attr.detach().expand(n_local, -1, -1).contiguous()
                cpu = cpu_baseline(v_pix, vi, a_full, H, W, min(args.cpu_sample_views, n_local))
        if textured:
            what = (f"{n_local} views/GPU x {world} GPU, {args.mesh}-tri UV-sphere scene (F={vi.shape[0]}, V={v_world.shape[0]}), {H}x{W}, "
                    f"RGB texture {args.tex}^2 + {len(tex) - 1} mip levels and uv attributes stored in fp16 under autocast, "
                    "transform+rasterize+render+interpolate(uv)+screen_space_uv_derivative+mipmap_grid_sample(aniso 8)+mask+"
                    "edge_grad_estimator+loss fwd+bwd")
            metric = "Mpixels/sec fwd+bwd, 1M-tri @ 4096x4096 textured (edge_grad + mipmap_grid_sampler, fp16 attributes); HBM BW %"
        else:
            what = (f"{n_local} views/GPU x {world} GPU, {args.mesh}-tri UV-sphere head mesh (F={vi.shape[0]}, V={v_world.shape[0]}), "
                    f"{H}x{W}, C={C} attribute channels, transform+rasterize+render+interpolate+mask+edge_grad_estimator+loss fwd+bwd")
            metric = "Mpixels/sec fwd+bwd, 100k-tri @ 2048x2048, 1/2/4/8 GPU; HBM BW %"
        nbytes = sum(r.nbytes() for r in reducers)
        result = {
            "metric": metric,
            "value": round(mpix, 2),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "ms_per_step_median_hipevent": round(median_step_ms, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": what, "baseline_config": f"BASELINE.json configs[{args.config - 1}]",
                "views_per_gpu": n_local, "triangles": int(vi.shape[0]), "vertices": int(v_world.shape[0]),
                "height": H, "width": W, "channels": C, "grad_reset": grad_reset,
                "parallelism": (f"views sharded {world}-way, no data-path collective; {nbytes} B of shared gradients all-reduced over RCCL "
                                "per step, each shared tensor on a side stream as soon as its gradient is final") if world > 1 else "single GPU",
            },
            "loss": round(loss_value, 6),
            "all_reduce": None if comm is None else {
                "bytes": nbytes, "collectives_per_step": sum(r.collectives_per_step() for r in reducers),
                "staging_dtype": str(reducers[0].flat.dtype).replace("torch.", ""),
                "ms_launch_to_done": round(comm[0], 4), "ms_exposed_on_main_stream": round(comm[1], 4),
                "note": "one extra step after the timed region, rank 0: side-stream time from each collective's launch to its "
                        "completion (sum), and how long the main stream then waited for them"},
            "graph_step": graph,
            "extensions": dict(ext or {}, **({"grad_reset_flat_zero": flat_zero_alt} if flat_zero_alt else {})) or None,
            "roofline": roofline,
            "path_roofline": path,
            "cpu_baseline": cpu,
        }
        emit_line(result)
    if grouped:
        th.distributed.barrier()
        th.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

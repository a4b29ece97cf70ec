"""Multi-GPU for the hot path: views shard, one all-reduce on the shared vertex gradients.

The reference has no distributed code.  Every kernel on the path treats the view
index as an outer, independent dimension, so a batch of camera views shards across the GPUs of a
node with NO data-path collective: each rank rasterizes/renders/interpolates its own views and all
per-pixel tensors stay local.  The only exchange is on tensors that are *shared across views* --
world-space vertices `[V,3]` fed through `transform`, attributes/textures broadcast as `[1,V,C]` --
whose gradients autograd has already summed over the local views: one fused `all_reduce(SUM)` per
step over RCCL (`backend="nccl"` is RCCL on ROCm).  At 12*V (+4*V*C) bytes the collective is
latency-bound on xGMI, so it is a single call on a single flat buffer, issued on a side stream.

One process per GPU; launch with `python -m torch.distributed.run --nproc-per-node N ...`.
"""
import os
from typing import Iterable, List, Optional, Tuple

import torch as th
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (as set by
    torch.distributed.run).  Returns (rank, world_size, local_rank); a no-op for world size 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            # DRTK_DIST_BACKEND=gloo lets the multi-rank code path be exercised on a single-GPU box
            backend = os.environ.get("DRTK_DIST_BACKEND") or ("nccl" if th.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            th.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, device_id=th.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    return rank, world, local_rank


def shard_views(n_views: int, rank: int, world: int) -> range:
    """Contiguous block partition of `n_views` over `world` ranks (ranks < n_views % world get
    one extra view).  Rank r owns views `shard_views(n, r, world)`."""
    base, rem = divmod(n_views, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


class SharedGradReducer:
    """Fuses the gradients of view-shared leaf tensors into one flat buffer and all-reduces it.

    Usage per step:  loss.backward();  reducer.all_reduce();  optimizer.step()
    The flat buffer and the side stream are created once; `all_reduce()` packs, launches the
    collective on the side stream (so a caller may overlap it with further work on the current
    stream), waits, and scatters the sums back into each `.grad`.
    """

    def __init__(self, params: Iterable[th.Tensor], average: bool = False):
        self.params: List[th.Tensor] = list(params)
        assert self.params, "no shared tensors given"
        dev, dt = self.params[0].device, self.params[0].dtype
        assert all(p.device == dev and p.dtype == dt for p in self.params)
        self.numel = sum(p.numel() for p in self.params)
        self.flat = th.zeros(self.numel, dtype=dt, device=dev)
        self.average = average
        self.stream = th.cuda.Stream(device=dev) if dev.type == "cuda" else None

    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()

    def all_reduce(self) -> None:
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        if self.stream is not None:
            self.stream.wait_stream(th.cuda.current_stream(self.flat.device))
            with th.cuda.stream(self.stream):
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            th.cuda.current_stream(self.flat.device).wait_stream(self.stream)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
        if self.average:
            self.flat.div_(dist.get_world_size())
        off = 0
        for p in self.params:
            n = p.numel()
            g = self.flat[off:off + n].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n


def barrier_and_sync(device=None) -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if th.cuda.is_available():
        th.cuda.synchronize(device)

"""Multi-GPU for the hot path: views shard, one all-reduce on the shared vertex gradients.

The reference has no distributed code.  Every kernel on the path treats the view
index as an outer, independent dimension, so a batch of camera views shards across the GPUs of a
node with NO data-path collective: each rank rasterizes/renders/interpolates its own views and all
per-pixel tensors stay local.  The only exchange is on tensors that are *shared across views* --
world-space vertices `[V,3]` fed through `transform`, attributes/textures broadcast as `[1,V,C]` --
whose gradients autograd has already summed over the local views: `all_reduce(SUM)` over RCCL
(`backend="nccl"` is RCCL on ROCm) of one flat buffer, 12*V (+4*V*C) bytes, latency-bound on xGMI.  Each
shared tensor's segment is reduced on a side stream as soon as autograd has finished its gradient
(`SharedGradReducer`), so the attributes' collective runs under the remaining backward kernels.

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
    """All-reduces the gradients of view-shared leaf tensors, overlapped with the rest of the backward pass.

    The gradients live in ONE flat buffer (`.grad` of every parameter is a view of its segment, so nothing is packed
    or unpacked).  A post-accumulate-grad hook on each parameter launches the all-reduce of its segment on a side
    stream the moment autograd has finished that gradient: on the hot path `attr.grad` (4*V*C bytes, most of the
    buffer) is final right after interpolate backward, and its collective then runs under render backward, the edge
    route and transform backward; the vertices' gradient is final last and is reduced at the end.  xGMI is
    point-to-point and these messages are latency-bound (megabytes), so each segment is one call.

    Usage per step:   reducer.zero_grad();  loss.backward();  reducer.finish();  optimizer.step()
    `finish()` reduces whatever no hook has reduced yet (a parameter without gradient this step still takes part, with
    zeros: every rank issues the same collectives in the same order), waits for the side stream and leaves the summed
    (or averaged) gradients in `.grad`.  `all_reduce()` is the old name of `finish()`; with `overlap=False` no hooks
    are installed and `finish()` does everything after the backward pass, in one call on the whole buffer.
    """

    def __init__(self, params: Iterable[th.Tensor], average: bool = False, overlap: bool = True):
        self.params: List[th.Tensor] = list(params)
        assert self.params, "no shared tensors given"
        dev, dt = self.params[0].device, self.params[0].dtype
        assert all(p.device == dev and p.dtype == dt and p.is_leaf for p in self.params)
        self.numel = sum(p.numel() for p in self.params)
        self.flat = th.zeros(self.numel, dtype=dt, device=dev)
        self.average = average
        self.overlap = overlap
        self.cuda = dev.type == "cuda"
        self.stream = th.cuda.Stream(device=dev) if self.cuda else None
        self.segments = []
        off = 0
        for p in self.params:
            self.segments.append(self.flat[off:off + p.numel()])
            off += p.numel()
        self._pending = {}   # param index -> work handle of its in-flight all-reduce
        self._reduced = set()
        self._events = []    # (start, end) HIP events on the side stream, one pair per collective of the step
        self._handles = []
        self._finished = False
        self._wait_events = None
        self.enabled = True  # False: hooks and finish() do nothing (a rank stepping on its own, e.g. for profiling)
        if overlap:
            for i, p in enumerate(self.params):
                self._handles.append(p.register_post_accumulate_grad_hook(lambda _p, i=i: self._launch(i)))
        self.zero_grad()

    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()

    def _active(self) -> bool:
        return self.enabled and dist.is_initialized() and dist.get_world_size() > 1

    def zero_grad(self) -> None:
        """Zero the flat buffer and (re-)attach every `.grad` as a view of its segment."""
        self.flat.zero_()
        for p, seg in zip(self.params, self.segments):
            p.grad = seg.view_as(p)
        self._pending.clear()
        self._reduced.clear()
        self._events.clear()

    def _new_step(self) -> None:
        if self._finished:  # first collective after a finish(): a new step's bookkeeping
            self._events.clear()
            self._finished = False

    def _launch(self, i: int) -> None:
        if not self._active() or i in self._reduced:
            return
        self._new_step()
        p, seg = self.params[i], self.segments[i]
        if p.grad is None:
            seg.zero_()
        elif p.grad.data_ptr() != seg.data_ptr():  # someone replaced .grad (e.g. zero_grad(set_to_none=True)): pack it
            seg.copy_(p.grad.reshape(-1))
        self._reduced.add(i)
        if self.cuda:
            cur = th.cuda.current_stream(self.flat.device)
            self.stream.wait_stream(cur)  # the gradient is written on the stream autograd runs on
            with th.cuda.stream(self.stream):
                ev0 = th.cuda.Event(enable_timing=True)
                ev0.record()
                self._pending[i] = dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True)
                self._events.append([ev0, None])
        else:
            self._pending[i] = dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True)

    def finish(self) -> None:
        if not self._active():
            return
        if not self.overlap and not self._reduced:
            # one call on the whole buffer (gradients that are not views of it are packed first)
            self._new_step()
            for i, (p, seg) in enumerate(zip(self.params, self.segments)):
                if p.grad is None:
                    seg.zero_()
                elif p.grad.data_ptr() != seg.data_ptr():
                    seg.copy_(p.grad.reshape(-1))
                self._reduced.add(i)
            if self.cuda:
                self.stream.wait_stream(th.cuda.current_stream(self.flat.device))
                with th.cuda.stream(self.stream):
                    ev0 = th.cuda.Event(enable_timing=True)
                    ev0.record()
                    self._pending[-1] = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)
                    self._events.append([ev0, None])
            else:
                self._pending[-1] = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=True)
        else:
            for i in range(len(self.params)):  # parameters whose hook did not fire, in a fixed order
                self._launch(i)
        if self.cuda:
            cur = th.cuda.current_stream(self.flat.device)
            with th.cuda.stream(self.stream):
                for k, work in enumerate(self._pending.values()):
                    work.wait()
                    ev1 = th.cuda.Event(enable_timing=True)
                    ev1.record()
                    if k < len(self._events):
                        self._events[k][1] = ev1
            w0, w1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
            w0.record(cur)
            cur.wait_stream(self.stream)
            w1.record(cur)
            self._wait_events = (w0, w1)
        else:
            for work in self._pending.values():
                work.wait()
        self._pending.clear()
        self._reduced.clear()  # the next backward pass starts a new round, whoever zeroes the gradients
        self._finished = True
        if self.average:
            self.flat.div_(dist.get_world_size())
        for p, seg in zip(self.params, self.segments):
            if p.grad is None or p.grad.data_ptr() != seg.data_ptr():
                p.grad = seg.view_as(p)

    all_reduce = finish

    def timings_ms(self):
        """(sum over this step's collectives of launch -> completion on the side stream, time the main stream spent
        waiting for them in finish()) in milliseconds; synchronises.  None on CPU / single rank."""
        if not (self.cuda and self._events and all(e[1] is not None for e in self._events)):
            return None
        th.cuda.synchronize(self.flat.device)
        total = sum(e0.elapsed_time(e1) for e0, e1 in self._events)
        return total, self._wait_events[0].elapsed_time(self._wait_events[1])


def barrier_and_sync(device=None) -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if th.cuda.is_available():
        th.cuda.synchronize(device)

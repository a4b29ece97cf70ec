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

# RCCL between processes needs dmabuf IPC on this platform (hipIpcGetMemHandle fails otherwise); only effective if nothing has
# initialised the HIP runtime yet, which is the case when a launcher script imports this module first
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch as th
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None, single_rank_group: bool = False) -> Tuple[int, int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (as set by
    torch.distributed.run).  Returns (rank, world_size, local_rank); a no-op for world size 1 unless
    `single_rank_group` asks for a one-rank process group (tests/test_gpu_rccl_single_rank.py: RCCL on a one-GPU box)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or single_rank_group) and not dist.is_initialized():
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


class _StagedCast(th.autograd.Function):
    """`p.to(staging dtype)` whose backward hands the gradient to the reducer IN THE STAGING DTYPE (see
    SharedGradReducer.upcast)."""

    @staticmethod
    def forward(ctx, p, reducer, index):
        ctx.reducer, ctx.index, ctx.dtype = reducer, index, p.dtype
        return p.to(reducer.flat.dtype)

    @staticmethod
    def backward(ctx, g):
        r = ctx.reducer
        if r._active():
            # the gradient of THIS cast tensor (autograd has summed its uses): accumulate it unrounded.  The leaf is not
            # marked ready here -- it may be cast a second time, or reach the loss directly as well (`p.float()`); its own
            # post-accumulate hook fires once every path into it has run (with an undefined gradient when upcast() was the
            # only one), and that is where the segment is completed and the group launched (_on_leaf_grad)
            r._check_not_in_flight(ctx.index)
            r._begin_round()
            r.segments[ctx.index].add_(g.reshape(-1))
            if not r._fallback_queued:
                # Fallback for the hook above not firing on an undefined gradient (engine behaviour, not API): at the END of
                # this backward pass every staged leaf that was seen and is not marked ready yet is marked then -- later than
                # the hook would have (no overlap with the rest of the pass), still from inside backward(), never wrong.
                # Queued once per BACKWARD PASS (the flag is cleared by the callback itself): `_staged_seen` lives for the whole
                # round, so after accumulation passes under no_sync() had filled it the synchronising pass queued nothing.
                r._fallback_queued = True
                th.autograd.Variable._execution_engine.queue_callback(r._staged_leaves_fallback)
            r._staged_seen.add(ctx.index)
            return None, None, None  # the leaf's .grad is written once, by finish(), from the reduced sum
        return g.to(ctx.dtype), None, None


class SharedGradReducer:
    """All-reduces the gradients of view-shared leaf tensors, overlapped with the rest of the backward pass.

    `params` lists tensors and/or LISTS of tensors; a list is a group that is reduced by ONE collective (many small
    leaves -- the levels of a mip pyramid -- must not cost one latency-bound call each).  All gradients live in ONE flat
    buffer of `dtype` (default: the parameters' common dtype, float32 if they differ):

    * a parameter of that dtype has its `.grad` set to a VIEW of its segment, so nothing is packed or unpacked;
    * a parameter of another dtype (fp16 attributes / textures) is STAGED: its segment holds the gradient in the buffer's
      dtype, the sum over the ranks is formed there, and `.grad` receives it rounded once, after the reduction.  Feed
      such a leaf to the pipeline through `reducer.upcast(p)` and its gradient is also ACCUMULATED unrounded (the
      gradient of the cast tensor goes straight into the segment); without `upcast` the leaf's own, already rounded
      `.grad` is copied into the segment when it is final.

    Order of the collectives.  Every rank must issue the same collectives in the same order, whatever its autograd
    graph looked like this step (a rank with an empty shard runs no backward pass at all; a parameter may be unused on
    one rank).  So the order is FIXED: groups are reduced last-listed first -- list the tensors in the order the
    forward pass uses them, their gradients then become final in the order the collectives are issued.  A gradient
    hook only marks its parameter ready; the next group in the fixed order is launched (on a side stream) as soon as
    all its members are ready, and `finish()` launches the rest in the same order, ready or not (a parameter that got
    no gradient takes part with zeros).  On the hot path `attr.grad` (4*V*C bytes, most of the buffer) is final right
    after interpolate backward: listed after the vertices, its collective runs under render backward, the edge route
    and transform backward; the vertices' gradient is final last and is reduced at the end.

    One backward pass per `finish()`: a group that was already reduced this round would silently receive
    un-reduced local gradients from a second pass, so that raises.  For gradient accumulation run the earlier passes
    under `with reducer.no_sync():` -- they only accumulate locally, the last pass (outside the context) reduces the sum.

    Usage per step:   reducer.zero_grad();  loss.backward();  reducer.finish();  optimizer.step()
    `all_reduce()` is the old name of `finish()`; with `overlap=False` nothing is launched from inside the backward
    pass and `finish()` reduces the whole buffer in one call.
    """

    def __init__(self, params: Iterable, average: bool = False, overlap: bool = True, dtype: Optional[th.dtype] = None):
        groups = [list(g) if isinstance(g, (list, tuple)) else [g] for g in params]
        self.params: List[th.Tensor] = [p for g in groups for p in g]
        assert self.params, "no shared tensors given"
        dev = self.params[0].device
        assert all(p.device == dev and p.is_leaf and p.is_floating_point() for p in self.params)
        dtypes = {p.dtype for p in self.params}
        if dtype is None:
            dtype = dtypes.pop() if len(dtypes) == 1 else th.float32
        self.numel = sum(p.numel() for p in self.params)
        self.flat = th.zeros(self.numel, dtype=dtype, device=dev)
        self.average = average
        self.overlap = overlap
        self.cuda = dev.type == "cuda"
        self.stream = th.cuda.Stream(device=dev) if self.cuda else None
        self.segments, self.direct, self.group_of = [], [], []
        self.group_members: List[List[int]] = []
        self.group_slices = []
        off = 0
        for gi, g in enumerate(groups):
            g_off, members = off, []
            for p in g:
                members.append(len(self.segments))
                self.segments.append(self.flat[off:off + p.numel()])
                self.direct.append(p.dtype == dtype)
                self.group_of.append(gi)
                off += p.numel()
            self.group_members.append(members)
            self.group_slices.append(self.flat[g_off:off])
        self.order = list(range(len(groups) - 1, -1, -1))  # launch order of the groups: fixed, the same on every rank
        self._next = 0          # position in `order` of the next group to launch
        self._ready = set()     # parameters whose gradient of this round is final
        self._staged_seen = set()  # staged parameters whose segment was filled through upcast() this round
        self._fallback_queued = False  # the end-of-backward callback is queued for the backward pass that is running
        self._pending = {}      # group index -> work handle of its in-flight all-reduce
        self._events = []       # (start, end) HIP events on the side stream, one pair per collective of the step
        self._handles = []
        self._finished = False
        self._defer = False
        self._staged_dirty = False  # staged segments still hold the last round's reduced sums (readable until the next round)
        self._wait_events = None
        self.enabled = True  # False: hooks and finish() do nothing (a rank stepping on its own, e.g. for profiling)
        # True: a process group of ONE rank still issues its collectives (the sum over one rank is the identity) -- how
        # tests/test_gpu_rccl_single_rank.py runs the real call sequence against RCCL on a one-GPU box
        self.run_single_rank = False
        # True: bracket every collective (and finish()'s wait) with HIP events for timings_ms().  Off by default: an event
        # record is a barrier packet on its stream, six of them per step are measurable on a 5 ms step (bench.py switches
        # it on for one extra step after its timed region)
        self.record_timings = False
        for i, p in enumerate(self.params):
            self._handles.append(p.register_post_accumulate_grad_hook(lambda _p, i=i: self._on_leaf_grad(i)))
        self.zero_grad()

    def nbytes(self) -> int:
        return self.flat.numel() * self.flat.element_size()

    def collectives_per_step(self) -> int:
        return len(self.group_members) if self.overlap else 1

    def _active(self) -> bool:
        return self.enabled and dist.is_initialized() and (dist.get_world_size() > 1 or self.run_single_rank)

    def upcast(self, p: th.Tensor) -> th.Tensor:
        """`p` in the buffer's dtype for use in the forward pass.  For a staged parameter (stored in another dtype) the
        gradient of the returned tensor is accumulated into the flat buffer unrounded and reduced there; `p.grad` is
        written once by `finish()`.  With the reducer inactive (one rank) this is `p.to(dtype)` with the usual gradient.
        May be called several times per step for the same leaf, and the leaf may ALSO be used directly (`p.float()`):
        every cast's gradient is added to the segment, the direct part (rounded to the leaf's dtype by autograd) is added
        when the leaf's own gradient hook fires, and only then does the leaf count as ready."""
        i = next((k for k, q in enumerate(self.params) if q is p), None)
        assert i is not None, "upcast(): not one of this reducer's tensors"
        return p if self.direct[i] else _StagedCast.apply(p, self, i)

    def no_sync(self):
        """Context manager: backward passes inside it only accumulate locally (no collective is launched); the first
        pass outside reduces the accumulated sum."""
        reducer = self

        class _NoSync:
            def __enter__(self):
                reducer._defer = True

            def __exit__(self, *exc):
                reducer._defer = False
                return False

        return _NoSync()

    def zero_grad(self) -> None:
        """Zero the flat buffer, (re-)attach every direct `.grad` as a view of its segment, drop the staged ones."""
        self.flat.zero_()
        for p, seg, direct in zip(self.params, self.segments, self.direct):
            p.grad = seg.view_as(p) if direct else None
        self._reset_round()
        self._staged_dirty = False
        self._events.clear()

    def _begin_round(self) -> None:
        """Staged segments ACCUMULATE, so a new round must find them zero: the sums finish() left there (in the buffer's
        dtype, for whoever wants them unrounded) are cleared when the next round's first gradient arrives."""
        if self._staged_dirty:
            for seg, direct in zip(self.segments, self.direct):
                if not direct:
                    seg.zero_()
            self._staged_dirty = False

    def _reset_round(self) -> None:
        self._pending.clear()
        self._ready.clear()
        self._staged_seen.clear()
        self._next = 0

    def _new_step(self) -> None:
        if self._finished:  # first collective after a finish(): a new step's bookkeeping
            self._events.clear()
            self._finished = False

    def _on_leaf_grad(self, i: int) -> None:
        # Runs after EVERY path into the leaf has been differentiated.  A leaf fed through upcast() arrives here with an
        # undefined gradient when the casts were its only uses; if it ALSO reached the loss directly, that part sits in
        # `.grad` (rounded to the leaf's dtype by autograd) and is moved into the segment beside the unrounded part.
        if i in self._staged_seen and self._active():
            self._absorb_direct_use(i)
        self._on_grad_ready(i)

    def _staged_leaves_fallback(self) -> None:
        self._fallback_queued = False  # (the engine runs this at the end of the pass that queued it)
        if not self._active() or self._defer:
            return
        for i in sorted(self._staged_seen):
            if i not in self._ready and self.group_of[i] not in self._pending:
                self._on_leaf_grad(i)

    def _absorb_direct_use(self, i: int) -> None:
        p = self.params[i]
        if p.grad is not None:
            self._check_not_in_flight(i)
            self.segments[i].add_(p.grad.reshape(-1).to(self.flat.dtype))
            p.grad = None

    def _check_not_in_flight(self, i: int) -> None:
        if self.group_of[i] in self._pending:
            raise RuntimeError(
                "SharedGradReducer: a second backward pass reached a gradient whose all-reduce was already launched this "
                "round -- it would add un-reduced local gradients to the reduced sum.  Call finish() after every backward "
                "pass, or run the passes to be accumulated under `with reducer.no_sync():`")

    def _on_grad_ready(self, i: int) -> None:
        if not self._active() or self._defer:
            return
        self._check_not_in_flight(i)
        self._begin_round()
        self._ready.add(i)
        if self.overlap:
            while self._next < len(self.order) and all(m in self._ready for m in self.group_members[self.order[self._next]]):
                self._launch(self.order[self._next])
                self._next += 1

    def _fill_segment(self, i: int) -> None:
        """Bring parameter i's segment up to date with its gradient, whichever way that gradient was delivered."""
        p, seg = self.params[i], self.segments[i]
        if i in self._staged_seen:
            self._absorb_direct_use(i)  # (normally done by the leaf's hook already; under no_sync() it is done here)
            return  # accumulated unrounded by upcast()'s backward
        if p.grad is None:
            if self.direct[i]:
                seg.zero_()  # (a staged segment without gradient is still zero from zero_grad() / the last finish())
        elif not (self.direct[i] and p.grad.data_ptr() == seg.data_ptr()):
            seg.copy_(p.grad.reshape(-1))  # replaced `.grad` (zero_grad(set_to_none=True)) or a staged leaf's own gradient

    def _all_reduce(self, key: int, buf: th.Tensor) -> None:
        if self.cuda:
            cur = th.cuda.current_stream(self.flat.device)
            self.stream.wait_stream(cur)  # the gradient is written on the stream autograd runs on
            with th.cuda.stream(self.stream):
                if self.record_timings:
                    ev0 = th.cuda.Event(enable_timing=True)
                    ev0.record()
                    self._events.append([ev0, None])
                self._pending[key] = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)
        else:
            self._pending[key] = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)

    def _launch(self, g: int) -> None:
        self._new_step()
        self._begin_round()
        for i in self.group_members[g]:
            self._fill_segment(i)
        self._all_reduce(g, self.group_slices[g])

    def finish(self) -> None:
        if not self._active():
            return
        if not self.overlap:
            # one call on the whole buffer
            self._new_step()
            self._begin_round()
            for i in range(len(self.params)):
                self._fill_segment(i)
            self._all_reduce(-1, self.flat)
        else:
            while self._next < len(self.order):  # groups no hook completed, in the same fixed order
                self._launch(self.order[self._next])
                self._next += 1
        if self.cuda:
            cur = th.cuda.current_stream(self.flat.device)
            with th.cuda.stream(self.stream):
                for k, work in enumerate(self._pending.values()):
                    work.wait()
                    if self.record_timings and k < len(self._events):
                        ev1 = th.cuda.Event(enable_timing=True)
                        ev1.record()
                        self._events[k][1] = ev1
            if self.record_timings:
                w0, w1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
                w0.record(cur)
                cur.wait_stream(self.stream)
                w1.record(cur)
                self._wait_events = (w0, w1)
            else:
                cur.wait_stream(self.stream)
        else:
            for work in self._pending.values():
                work.wait()
        self._reset_round()  # the next backward pass starts a new round, whoever zeroes the gradients
        self._finished = True
        if self.average:
            self.flat.div_(dist.get_world_size())
        for p, seg, direct in zip(self.params, self.segments, self.direct):
            if direct:
                if p.grad is None or p.grad.data_ptr() != seg.data_ptr():
                    p.grad = seg.view_as(p)
            else:
                p.grad = seg.view_as(p).to(p.dtype)  # the ONE rounding of a staged gradient
                self._staged_dirty = True

    all_reduce = finish

    def timings_ms(self):
        """(sum over this step's collectives of launch -> completion on the side stream, time the main stream spent
        waiting for them in finish()) in milliseconds; synchronises.  None on CPU / single rank, or when the step ran
        without `record_timings`."""
        if not (self.cuda and self._events and self._wait_events and all(e[1] is not None for e in self._events)):
            return None
        th.cuda.synchronize(self.flat.device)
        total = sum(e0.elapsed_time(e1) for e0, e1 in self._events)
        return total, self._wait_events[0].elapsed_time(self._wait_events[1])


def barrier_and_sync(device=None) -> None:
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if th.cuda.is_available():
        th.cuda.synchronize(device)

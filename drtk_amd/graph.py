"""`capture_step` -- one training step as a HIP graph.

Every kernel of the library is capture-safe by construction (no synchronisation, no host read of device data, grids
from static sizes, zero-fills done by kernels -- DESIGN.md section 1, profiles/NOTES.md "Graph capture"), so a whole step -- transform ->
rasterize -> render -> interpolate -> shading -> edge_grad_estimator -> loss -> backward -- can be recorded once with
torch's whole-network capture recipe and replayed.  That pays where the step is bound by launches rather than by the
GPU: 4 views of a 10k-triangle mesh at 512 x 512 take 0.48 ms eagerly (25 launches, ~0.25 ms of kernels) and 0.27 ms
replayed; at the 2048 x 2048 benchmark shape the GPU is busy either way.
"""
from typing import Callable, Iterable, Optional

import torch as th


class CapturedStep:
    """Result of `capture_step`: call it to replay.  `outputs` is what `step()` returned during capture (static tensors,
    refreshed by every replay); the leaves' `.grad` tensors are static too and are overwritten by every replay."""

    def __init__(self, graph, outputs, leaves):
        self.graph, self.outputs, self.leaves = graph, outputs, list(leaves)
        self.grads = [p.grad for p in self.leaves]

    def __call__(self):
        for p, g in zip(self.leaves, self.grads):
            p.grad = g  # whoever set it to None / replaced it between replays: the graph writes into these tensors
        self.graph.replay()
        return self.outputs


def capture_step(step: Callable[[], object], leaves: Iterable[th.Tensor], warmup: int = 3,
                 stream: Optional["th.cuda.Stream"] = None, reducers: Iterable = ()) -> CapturedStep:
    """Record `step()` -- forward, loss AND `backward()` -- into a HIP graph.

    step     a function without arguments that runs one whole step on static input tensors (update them IN PLACE
             between replays: `v_world.copy_(...)`, an optimizer's in-place update) and returns whatever should stay
             readable (the loss); it must call `.backward()` itself and must not synchronise or read device data.
    leaves   the tensors whose `.grad` the step produces; their gradients are set to None before the capture so that
             the captured backward pass allocates them from the graph's private pool (torch's capture recipe), and stay
             attached afterwards.
    warmup   eager runs on a side stream before the capture (allocator and autograd warm-up, as torch prescribes).
    stream   the side stream of the warm-up runs AND of the capture (default: a new one).
    reducers `SharedGradReducer`s attached to the leaves.  Their gradient hooks would launch all-reduces on the reducer's
             own stream from inside the warm-up and the captured backward pass -- work the capture cannot join, so it
             fails -- and are therefore switched off (`enabled = False`) for the duration of this call and restored after.

    Multi-GPU: capture the LOCAL step only and run `SharedGradReducer.finish()` after each replay; a collective inside
    the captured region is not supported here.  A reducer that is not passed in `reducers` must be disabled by the caller.
    """
    leaves = list(leaves)
    assert leaves and all(p.is_cuda and p.is_leaf for p in leaves), "capture_step(): leaves must be CUDA leaf tensors"
    reducers = list(reducers)
    was_enabled = [r.enabled for r in reducers]
    for r in reducers:
        r.enabled = False
    try:
        side = stream if stream is not None else th.cuda.Stream(device=leaves[0].device)
        side.wait_stream(th.cuda.current_stream(leaves[0].device))
        with th.cuda.stream(side):
            for _ in range(max(warmup, 1)):
                for p in leaves:
                    p.grad = None
                step()
        th.cuda.current_stream(leaves[0].device).wait_stream(side)
        th.cuda.synchronize(leaves[0].device)
        for p in leaves:
            p.grad = None
        graph = th.cuda.CUDAGraph()
        with th.cuda.graph(graph, stream=side):
            outputs = step()
    finally:
        for r, e in zip(reducers, was_enabled):
            r.enabled = e
    return CapturedStep(graph, outputs, leaves)

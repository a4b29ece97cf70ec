#!/usr/bin/env python3
"""Times alternative PyTorch formulations of the user-side glue of the benchmark step
(mask multiply + loss, forward and backward) on the bench shapes."""
import torch as th

dev = "cuda:0"
N, C, H, W = 8, 16, 2048, 2048
img0 = th.rand(N, C, H, W, device=dev)
index = (th.rand(N, H, W, device=dev) < 0.57).int() - 1
depth = th.rand(N, H, W, device=dev, requires_grad=True)


def timeit(name, fn, reps=5):
    fn()
    th.cuda.synchronize()
    e0, e1 = th.cuda.Event(enable_timing=True), th.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    th.cuda.synchronize()
    print(f"{name:60s} {e0.elapsed_time(e1) / reps:8.3f} ms")


def variant(mask_fn, loss_fn):
    def run():
        img = img0.clone().requires_grad_(True)  # stands for interpolate's output (+1 copy, same in all variants)
        m = mask_fn(img)
        loss = loss_fn(m) + depth.mean()
        loss.backward()
    return run


masks = {
    "img * bool[:,None]": lambda img: img * (index != -1)[:, None],
    "img * float[:,None]": lambda img: img * (index != -1).to(img.dtype)[:, None],
    "where(bool[:,None], img, 0)": lambda img: th.where((index != -1)[:, None], img, 0.0),
    "masked_fill(~bool[:,None], 0)": lambda img: img.masked_fill((index == -1)[:, None], 0.0),
    "img * float.expand.contiguous": lambda img: img * (index != -1).to(img.dtype)[:, None].expand(-1, C, -1, -1).contiguous(),
}
losses = {
    "(x*x).mean()": lambda x: (x * x).mean(),
    "x.square().mean()": lambda x: x.square().mean(),
    "vector_norm(x)^2/numel": lambda x: th.linalg.vector_norm(x).square() / x.numel(),
    "x.pow(2).sum()/numel": lambda x: x.pow(2).sum() / x.numel(),
}
timeit("clone only", lambda: img0.clone())
for mn, mf in masks.items():
    timeit(f"{mn} + (x*x).mean()", variant(mf, losses["(x*x).mean()"]))
for ln, lf in losses.items():
    timeit(f"img * bool[:,None] + {ln}", variant(masks["img * bool[:,None]"], lf))

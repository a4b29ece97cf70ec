"""TEST INFRASTRUCTURE ONLY -- CPU "drtk-like" op sets built from a kernel backend.

`make_ops(backend)` wires forward/backward CPU kernels into torch.autograd Functions with the same
contracts as the reference's C++ autograd nodes (rasterize_module.cpp:31-52, render_module.cpp:27-72,
interpolate_module.cpp:378-433, edge_grad_module.cpp:114-170) and returns an object exposing
rasterize / rasterize_with_depth / render / interpolate / edge_grad_estimator with the drtk.*
signatures.  Backends:

  OracleBackend   -- oracle/libdrtk_oracle.so (our plain-C restatement)
  RefBackend      -- oracle/_ref/libdrtk_ref_<variant>.so (the reference's own CPU kernels)

Used by the golden-vector generator, the CPU parity tests and the world_size-2 gloo test.  The
product (drtk_amd/) never imports this.
"""
import os
import sys
import types

import torch as th

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.join(_ROOT, "oracle") not in sys.path:
    sys.path.insert(0, os.path.join(_ROOT, "oracle"))


class OracleBackend:
    name = "oracle"

    def __init__(self, nthreads=1):
        import oracle as O

        self.O = O
        self.nt = nthreads

    def rasterize(self, v, vi, h, w):
        return self.O.rasterize(v, vi, h, w, nthreads=self.nt)

    def render(self, v, vi, index_img):
        return self.O.render(v, vi, index_img, nthreads=self.nt)

    def render_backward(self, v, vi, index_img, gd, gb):
        return self.O.render_backward(v, vi, index_img, gd, gb, nthreads=self.nt)

    def interpolate(self, a, vi, index_img, bary_img):
        return self.O.interpolate(a, vi, index_img, bary_img, nthreads=self.nt)

    def interpolate_backward(self, go, a, vi, index_img, bary_img, vert_rg, bary_rg):
        return self.O.interpolate_backward(go, a, vi, index_img, bary_img, vert_rg, bary_rg, nthreads=self.nt)

    def edge_grad_backward(self, v_pix, img, index_img, vi, go, max_dp_dr):
        return self.O.edge_grad_backward(v_pix, img, index_img, vi, go, max_dp_dr, nthreads=self.nt)


class RefBackend:
    """The reference's CPU kernels (single torch thread => deterministic accumulation order)."""

    def __init__(self, variant="strict"):
        import ref_build

        self.ns = ref_build.load(variant)
        self.name = f"ref_{variant}"

    @staticmethod
    def _vib(vi, n):
        return vi[None].expand(n, -1, -1) if vi.ndim == 2 else vi

    def rasterize(self, v, vi, h, w):
        d, i = self.ns.rasterize(v, self._vib(vi, v.shape[0]), h, w)
        return d, i

    def render(self, v, vi, index_img):
        d, b = self.ns.render(v, self._vib(vi, v.shape[0]), index_img)
        return d, b

    def render_backward(self, v, vi, index_img, gd, gb):
        return self.ns.render_backward(v, self._vib(vi, v.shape[0]), index_img, gd, gb)

    def interpolate(self, a, vi, index_img, bary_img):
        return self.ns.interpolate(a, self._vib(vi, a.shape[0]), index_img, bary_img)

    def interpolate_backward(self, go, a, vi, index_img, bary_img, vert_rg, bary_rg):
        vg, bg = self.ns.interpolate_backward(go, a, self._vib(vi, a.shape[0]), index_img, bary_img, vert_rg, bary_rg)
        return (vg if vert_rg else None), (bg if bary_rg else None)

    def edge_grad_backward(self, v_pix, img, index_img, vi, go, max_dp_dr):
        return self.ns.edge_grad_backward(v_pix, img, index_img, self._vib(vi, v_pix.shape[0]), go, max_dp_dr)


def make_ops(backend):
    B = backend

    class _Render(th.autograd.Function):
        @staticmethod
        def forward(ctx, v, vi, index_img):
            ctx.save_for_backward(v, vi, index_img)
            ctx.rg = v.requires_grad
            d, b = B.render(v.detach(), vi, index_img)
            return d, b

        @staticmethod
        def backward(ctx, gd, gb):
            if not ctx.rg:
                return None, None, None
            v, vi, index_img = ctx.saved_tensors
            if gd is None:
                gd = th.zeros(index_img.shape, dtype=v.dtype)
            if gb is None:
                gb = th.zeros(index_img.shape[0], 3, *index_img.shape[1:], dtype=v.dtype)
            return B.render_backward(v.detach(), vi, index_img, gd.contiguous(), gb.contiguous()), None, None

    class _Interpolate(th.autograd.Function):
        @staticmethod
        def forward(ctx, a, vi, index_img, bary_img):
            ctx.save_for_backward(a, vi, index_img, bary_img)
            ctx.rg = (a.requires_grad, bary_img.requires_grad)
            return B.interpolate(a.detach(), vi, index_img, bary_img.detach())

        @staticmethod
        def backward(ctx, go):
            a, vi, index_img, bary_img = ctx.saved_tensors
            vert_rg, bary_rg = ctx.rg
            if (not vert_rg and not bary_rg) or go is None:
                return None, None, None, None
            vg, bg = B.interpolate_backward(go.contiguous(), a.detach(), vi, index_img, bary_img.detach(), vert_rg, bary_rg)
            return vg, None, None, bg

    class _EdgeGrad(th.autograd.Function):
        @staticmethod
        def forward(ctx, v_pix, v_pix_img, vi, img, index_img, max_dp_dr):
            ctx.save_for_backward(v_pix, img, index_img, vi)
            ctx.rg = v_pix_img.requires_grad
            ctx.max_dp_dr = max_dp_dr
            return img.view_as(img)

        @staticmethod
        def backward(ctx, go):
            if not ctx.rg or go is None:
                return None, None, None, go, None, None
            v_pix, img, index_img, vi = ctx.saved_tensors
            g = B.edge_grad_backward(v_pix.detach(), img.detach(), index_img, vi, go.contiguous(), ctx.max_dp_dr)
            return None, g, None, go, None, None

    ops = types.SimpleNamespace()
    ops.backend = B

    def rasterize_with_depth(v, vi, height, width, wireframe=False):
        assert not wireframe
        d, i = B.rasterize(v.detach(), vi, height, width)
        return d, i

    def rasterize(v, vi, height, width, wireframe=False):
        return rasterize_with_depth(v, vi, height, width, wireframe)[1]

    def render(v, vi, index_img):
        return _Render.apply(v, vi, index_img)

    def interpolate(vert_attributes, vi, index_img, bary_img):
        return _Interpolate.apply(vert_attributes, vi, index_img, bary_img)

    def edge_grad_estimator(v_pix, vi, bary_img, img, index_img, v_pix_img_hook=None, max_dp_dr=1e4):
        v_pix_img = interpolate(v_pix, vi, index_img, bary_img.detach())
        out = _EdgeGrad.apply(v_pix, v_pix_img, vi, img, index_img, max_dp_dr)
        if v_pix_img_hook is not None:
            v_pix_img.register_hook(v_pix_img_hook)
        return out

    ops.rasterize = rasterize
    ops.rasterize_with_depth = rasterize_with_depth
    ops.render = render
    ops.interpolate = interpolate
    ops.edge_grad_estimator = edge_grad_estimator
    return ops

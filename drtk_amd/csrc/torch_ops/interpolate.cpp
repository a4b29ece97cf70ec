// interpolate_ext::interpolate -- src/interpolate/interpolate_module.cpp:376-433,584-669 -- and the drtk_amd extension
// interpolate_masked (background zeroed in the same pass), over drtk_amd_interpolate / drtk_amd_interpolate_backward.
#include "common.hpp"

namespace {
using namespace drtk_amd_torch;

// ---------------------------------------------------------------------------------------------
// interpolate
// ---------------------------------------------------------------------------------------------
void interpolate_checks(const Tensor& a, const Tensor& vi, const Tensor& index_img, const Tensor& bary_img) {
  // interpolate_kernel.cu:459-526
  TORCH_CHECK(
      a.defined() && vi.defined() && index_img.defined() && bary_img.defined(),
      "interpolate(): expected all inputs to be defined");
  TORCH_CHECK(
      (a.device() == vi.device()) && (a.device() == index_img.device()) &&
          (a.device() == bary_img.device()) && a.is_cuda(),
      "interpolate(): expected all inputs to be on same cuda device");
  TORCH_CHECK(
      a.dtype() == bary_img.dtype(),
      "interpolate(): expected vert_attributes and bary_img to have same dtype, but vert_attributes has ",
      a.dtype(), " and bary_img has ", bary_img.dtype());
  TORCH_CHECK(
      a.is_floating_point(),
      "interpolate(): expected vert_attributes to have floating point type, but vert_attributes has ", a.dtype());
  TORCH_CHECK(vi.dtype() == at::kInt, "interpolate(): expected vi to have int32 type, but vi has ", vi.dtype());
  TORCH_CHECK(
      index_img.dtype() == at::kInt,
      "interpolate(): expected index_img to have int32 type, but index_img has ", index_img.dtype());
  TORCH_CHECK(
      a.layout() == at::kStrided && vi.layout() == at::kStrided && index_img.layout() == at::kStrided &&
          bary_img.layout() == at::kStrided,
      "interpolate(): expected all inputs to have torch.strided layout");
  TORCH_CHECK(
      (a.dim() == 3) && (vi.dim() == 3) && (index_img.dim() == 3) && (bary_img.dim() == 4),
      "interpolate(): expected vert_attributes.ndim == 3, vi.ndim == 3, index_img.ndim == 3, bary_img.ndim == 4, "
      "but got vert_attributes with sizes ", a.sizes(), " and vi with sizes ", vi.sizes(),
      " and index_img with sizes ", index_img.sizes(), " and bary_img with sizes ", bary_img.sizes());
  TORCH_CHECK(
      a.size(0) == index_img.size(0) && a.size(0) == bary_img.size(0),
      "interpolate(): expected vert_attributes, index_img and bary_img to have same batch size, "
      "but got vert_attributes with sizes ", a.sizes(), ", index_img with sizes ", index_img.sizes(),
      " and bary_img with sizes ", bary_img.sizes());
  TORCH_CHECK(
      vi.size(2) == 3 && bary_img.size(1) == 3,
      "interpolate(): expected last dim of vi to be 3 and second dim of bary_img to be 3, but got ",
      vi.size(2), " in the last dim of vi, and ", bary_img.size(1), " in the second dim of bary_img");
  TORCH_CHECK(
      vi.size(0) == a.size(0),
      "interpolate(): expected first dim of vi to match first dim of vert_attributes but got ", a.size(0),
      " in first dim of vert_attributes, and ", vi.size(0), " in the first dim of vi");
  TORCH_CHECK(
      index_img.size(1) == bary_img.size(2) && index_img.size(2) == bary_img.size(3),
      "interpolate(): expected H and W dims of index_img and bary_img to match");
}

Tensor interpolate_launch(const Tensor& a, const Tensor& vi, const Tensor& index_img, const Tensor& bary_img, bool masked) {
  interpolate_checks(a, vi, index_img, bary_img);
  const drtk_dtype_t dt = dtype_of(a, "interpolate");
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(a.device());
  const auto a_c = a.contiguous();
  const auto idx_c = index_img.contiguous();
  const auto bary_c = bary_img.contiguous();
  const ViArg via = prep_vi(vi);
  const int64_t N = a.size(0), V = a.size(1), C = a.size(2), F = vi.size(1), H = bary_img.size(2), W = bary_img.size(3);
  auto out = out_empty({N, C, H, W}, a.options());
  check_status(
      (masked ? drtk_amd_interpolate_masked : drtk_amd_interpolate)(
          dt, a_c.data_ptr(), via.ptr, idx_c.data_ptr<int32_t>(), bary_c.data_ptr(), N, V, C, F, via.sN,
          H, W, out.data_ptr(), current_stream(a)),
      "interpolate");
  return out;
}

Tensor interpolate_hip(const Tensor& a, const Tensor& vi, const Tensor& index_img, const Tensor& bary_img) {
  return interpolate_launch(a, vi, index_img, bary_img, false);
}
// extension: background written as 0 (= interpolate(...) * (index_img != -1)[:, None] in one pass)
Tensor interpolate_masked_hip(const Tensor& a, const Tensor& vi, const Tensor& index_img, const Tensor& bary_img) {
  return interpolate_launch(a, vi, index_img, bary_img, true);
}

std::tuple<Tensor, Tensor> interpolate_backward_hip(
    const Tensor& grad_out, const Tensor& a, const Tensor& vi, const Tensor& index_img,
    const Tensor& bary_img, bool vert_requires_grad, bool bary_requires_grad) {
  const drtk_dtype_t dt = dtype_of(a, "interpolate_backward");
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(a.device());
  const auto a_c = a.contiguous();
  const auto idx_c = index_img.contiguous();
  const auto bary_c = bary_img.contiguous();
  const auto go_c = grad_out.to(a.scalar_type()).contiguous();
  const ViArg via = prep_vi(vi);
  const int64_t N = a.size(0), V = a.size(1), C = a.size(2), F = vi.size(1), H = bary_img.size(2), W = bary_img.size(3);
  // interpolate_kernel.cu:657-663
  Tensor vert_grad = vert_requires_grad ? out_empty({N, V, C}, a.options()) : Tensor();
  Tensor bary_grad = bary_requires_grad ? out_empty({N, 3, H, W}, bary_img.options()) : Tensor();
  // the optional scratch of the padded-row route (attribute rows that are not whole 64-byte segments: include/drtk_amd.h)
  size_t ws_bytes = 0;
  if (vert_requires_grad) check_status(drtk_amd_interpolate_backward_workspace_bytes(dt, N, V, C, &ws_bytes), "interpolate_backward");
  Tensor ws = ws_bytes ? alloc_workspace(ws_bytes, a) : Tensor();
  check_status(
      drtk_amd_interpolate_backward_ws(
          dt, go_c.data_ptr(), a_c.data_ptr(), via.ptr, idx_c.data_ptr<int32_t>(), bary_c.data_ptr(), N, V,
          C, F, via.sN, H, W, vert_requires_grad ? vert_grad.data_ptr() : nullptr,
          bary_requires_grad ? bary_grad.data_ptr() : nullptr, ws_bytes ? ws.data_ptr() : nullptr, ws_bytes, current_stream(a)),
      "interpolate_backward");
  return {vert_grad, bary_grad};
}

Tensor interpolate_cpu(const Tensor&, const Tensor&, const Tensor&, const Tensor&) {
  no_cpu("interpolate");
}

Tensor interpolate_op(const Tensor& a, const Tensor& vi, const Tensor& index_img, const Tensor& bary_img) {
  static auto op = c10::Dispatcher::singleton()
                       .findSchemaOrThrow("interpolate_ext::interpolate", "")
                       .typed<decltype(interpolate_op)>();
  return op.call(a, vi, index_img, bary_img);
}

class InterpolateFunction : public torch::autograd::Function<InterpolateFunction> {
 public:
  static tensor_list forward(
      AutogradContext* ctx, const Tensor& a, const Tensor& vi, const Tensor& index_img, const Tensor& bary_img) {
    ctx->set_materialize_grads(false);
    ctx->save_for_backward({a, vi, index_img, bary_img});
    at::AutoDispatchBelowADInplaceOrView g;
    return {interpolate_op(a, vi, index_img, bary_img)};
  }
  static tensor_list backward(AutogradContext* ctx, tensor_list grad_outputs) {
    const auto saved = ctx->get_saved_variables();
    const Tensor& a = saved[0];
    const Tensor& bary_img = saved[3];
    const bool bary_rg = bary_img.requires_grad(), vert_rg = a.requires_grad(); // interpolate_module.cpp:407-408
    if ((!bary_rg && !vert_rg) || !grad_outputs[0].defined()) return {Tensor(), Tensor(), Tensor(), Tensor()};
    auto g = interpolate_backward_hip(grad_outputs[0], a, saved[1], saved[2], bary_img, vert_rg, bary_rg);
    return {std::get<0>(g), Tensor(), Tensor(), std::get<1>(g)};
  }
};

Tensor interpolate_autograd(const Tensor& a, const Tensor& vi, const Tensor& index_img, const Tensor& bary_img) {
  return InterpolateFunction::apply(a, vi, index_img, bary_img)[0];
}

Tensor interpolate_masked_op(const Tensor& a, const Tensor& vi, const Tensor& index_img, const Tensor& bary_img) {
  static auto op = c10::Dispatcher::singleton()
                       .findSchemaOrThrow("drtk_amd_ext::interpolate_masked", "")
                       .typed<decltype(interpolate_masked_op)>();
  return op.call(a, vi, index_img, bary_img);
}
// Same VJP as interpolate: the backward ignores the upstream gradient of background pixels, which is
// exactly what multiplying the output by the mask would do to it.
class InterpolateMaskedFunction : public torch::autograd::Function<InterpolateMaskedFunction> {
 public:
  static tensor_list forward(
      AutogradContext* ctx, const Tensor& a, const Tensor& vi, const Tensor& index_img, const Tensor& bary_img) {
    ctx->set_materialize_grads(false);
    ctx->save_for_backward({a, vi, index_img, bary_img});
    at::AutoDispatchBelowADInplaceOrView g;
    return {interpolate_masked_op(a, vi, index_img, bary_img)};
  }
  static tensor_list backward(AutogradContext* ctx, tensor_list grad_outputs) {
    return InterpolateFunction::backward(ctx, grad_outputs);
  }
};
Tensor interpolate_masked_autograd(const Tensor& a, const Tensor& vi, const Tensor& index_img, const Tensor& bary_img) {
  return InterpolateMaskedFunction::apply(a, vi, index_img, bary_img)[0];
}
Tensor interpolate_masked_cpu(const Tensor&, const Tensor&, const Tensor&, const Tensor&) {
  no_cpu("interpolate_masked");
}

Tensor interpolate_autocast(const Tensor& a, const Tensor& vi, const Tensor& index_img, const Tensor& bary_img) {
  c10::impl::ExcludeDispatchKeyGuard no_autocast(c10::DispatchKey::Autocast);
  return interpolate_op(
      at::autocast::cached_cast(at::kFloat, a), vi, index_img, at::autocast::cached_cast(at::kFloat, bary_img));
}

} // namespace

// schema: verbatim from the reference (the sparse operators of the same namespace: interp_matrix.cpp)
TORCH_LIBRARY(interpolate_ext, m) {
  m.def("interpolate(Tensor vert_attributes, Tensor vi, Tensor index_img, Tensor bary_img) -> Tensor");
}
TORCH_LIBRARY_IMPL(interpolate_ext, Autograd, m) {
  m.impl("interpolate", &interpolate_autograd);
}
TORCH_LIBRARY_IMPL(interpolate_ext, Autocast, m) {
  m.impl("interpolate", interpolate_autocast);
}
TORCH_LIBRARY_IMPL(interpolate_ext, CUDA, m) {
  m.impl("interpolate", &interpolate_hip);
}
TORCH_LIBRARY_IMPL(interpolate_ext, CPU, m) {
  m.impl("interpolate", &interpolate_cpu);
}

// drtk_amd's own namespace (extensions with no reference counterpart)
TORCH_LIBRARY_FRAGMENT(drtk_amd_ext, m) {
  m.def("interpolate_masked(Tensor vert_attributes, Tensor vi, Tensor index_img, Tensor bary_img) -> Tensor");
}
TORCH_LIBRARY_IMPL(drtk_amd_ext, Autograd, m) {
  m.impl("interpolate_masked", &interpolate_masked_autograd);
}
TORCH_LIBRARY_IMPL(drtk_amd_ext, CUDA, m) {
  m.impl("interpolate_masked", &interpolate_masked_hip);
}
TORCH_LIBRARY_IMPL(drtk_amd_ext, CPU, m) {
  m.impl("interpolate_masked", &interpolate_masked_cpu);
}

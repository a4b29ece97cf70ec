// edge_grad_ext::edge_grad_estimator -- src/edge_grad/edge_grad_module.cpp:18-224 -- and the fused variant that takes
// bary_img directly (drtk_amd extension), over drtk_amd_edge_grad_backward[_fused].
#include "common.hpp"

namespace {
using namespace drtk_amd_torch;

// ---------------------------------------------------------------------------------------------
// edge_grad_estimator
// ---------------------------------------------------------------------------------------------
Tensor edge_grad_fwd_hip(
    const Tensor& v_pix, const Tensor& v_pix_img, const Tensor& vi, const Tensor& img,
    const Tensor& index_img, double /*max_dp_dr*/) {
  // edge_grad_module.cpp:30-112 : validation only, returns img itself
  TORCH_CHECK(
      v_pix.defined() && v_pix_img.defined() && vi.defined() && img.defined() && index_img.defined(),
      "edge_grad_estimator(): expected all inputs to be defined");
  TORCH_CHECK(
      (v_pix.device() == v_pix_img.device()) && (v_pix.device() == vi.device()) &&
          (v_pix.device() == img.device()) && (v_pix.device() == index_img.device()) && v_pix.is_cuda(),
      "edge_grad_estimator(): expected all inputs to be on same cuda device");
  TORCH_CHECK(
      v_pix.is_floating_point() && v_pix_img.is_floating_point() && img.is_floating_point(),
      "edge_grad_estimator(): expected v_pix, v_pix_img, and img to have floating point type, but v_pix has ",
      v_pix.dtype(), " v_pix has ", v_pix_img.dtype(), " img has ", img.dtype());
  TORCH_CHECK(vi.dtype() == at::kInt, "edge_grad_estimator(): expected vi to have int32 type, but vi has ", vi.dtype());
  TORCH_CHECK(
      index_img.dtype() == at::kInt,
      "edge_grad_estimator(): expected index_img to have int32 type, but index_img has ", index_img.dtype());
  TORCH_CHECK(
      v_pix.layout() == at::kStrided && v_pix_img.layout() == at::kStrided && vi.layout() == at::kStrided &&
          img.layout() == at::kStrided && index_img.layout() == at::kStrided,
      "edge_grad_estimator(): expected all inputs to have torch.strided layout");
  TORCH_CHECK(
      (v_pix.dim() == 3) && (v_pix_img.dim() == 4) && (vi.dim() == 3) && (img.dim() == 4) && (index_img.dim() == 3),
      "edge_grad_estimator(): expected v_pix.ndim == 3, v_pix_img.ndim == 4, vi.ndim == 3, img.ndim == 4, index_img.ndim == 3, "
      "but got v_pix with sizes ", v_pix.sizes(), " and v_pix_img with sizes ", v_pix_img.sizes(),
      " and vi with sizes ", vi.sizes(), " and img with sizes ", img.sizes(), " and index_img with sizes ",
      index_img.sizes());
  TORCH_CHECK(
      v_pix.size(0) == v_pix_img.size(0) && v_pix.size(0) == img.size(0) && v_pix.size(0) == index_img.size(0),
      "edge_grad_estimator(): expected v and index_img to have same batch size, but got v_pix with sizes ",
      v_pix.sizes(), ", v_pix_img with sizes ", v_pix_img.sizes(), ", img with sizes ", img.sizes(),
      " and index_img with sizes ", index_img.sizes());
  TORCH_CHECK(
      v_pix.size(2) == 3 && v_pix_img.size(1) == 3 && vi.size(2) == 3,
      "edge_grad_estimator(): expected third dim of v_pix to be of size 3, and third dim of vi to be of size 3, but got ",
      v_pix.size(2), " in the third dim of v_pix, and ", v_pix_img.size(1), " in the second dim of v_pix_img, and ",
      vi.size(2), " in the third dim of vi");
  TORCH_CHECK(
      v_pix_img.size(3) == img.size(3) && v_pix_img.size(3) == index_img.size(2) &&
          v_pix_img.size(2) == img.size(2) && v_pix_img.size(2) == index_img.size(1),
      "edge_grad_estimator(): expected width and height of v_pix_img, img, and index_img to match, but got size of v_pix_img: ",
      v_pix_img.sizes(), ", size of img: ", img.sizes(), ", size of index_img: ", index_img.sizes());
  return img;
}

Tensor edge_grad_backward_hip(
    const Tensor& v_pix, const Tensor& img, const Tensor& index_img, const Tensor& vi,
    const Tensor& grad_outputs, double max_dp_dr) {
  const drtk_dtype_t dt = dtype_of(v_pix, "edge_grad_estimator_backward");
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(v_pix.device());
  const auto v_c = v_pix.contiguous();
  const auto img_c = img.to(v_pix.scalar_type()).contiguous();
  const auto idx_c = index_img.contiguous();
  const auto go_c = grad_outputs.to(v_pix.scalar_type()).contiguous();
  const ViArg via = prep_vi(vi);
  const int64_t N = img.size(0), C = img.size(1), H = img.size(2), W = img.size(3), V = v_pix.size(1), F = vi.size(1);
  auto grad_v_pix_img = out_empty({N, 3, H, W}, v_pix.options()); // fully written by the call
  size_t ws_bytes = 0;
  check_status(drtk_amd_edge_grad_backward_workspace_bytes(dt, N, H, W, &ws_bytes), "edge_grad_estimator");
  auto ws = alloc_workspace(ws_bytes, v_pix);
  check_status(
      drtk_amd_edge_grad_backward(
          dt, v_c.data_ptr(), img_c.data_ptr(), idx_c.data_ptr<int32_t>(), via.ptr, go_c.data_ptr(), N, V, C,
          F, via.sN, H, W, max_dp_dr, grad_v_pix_img.data_ptr(), ws.data_ptr(), ws_bytes, current_stream(v_pix)),
      "edge_grad_estimator");
  return grad_v_pix_img;
}

Tensor edge_grad_cpu(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, double) {
  no_cpu("edge_grad_estimator");
}

Tensor edge_grad_op(
    const Tensor& v_pix, const Tensor& v_pix_img, const Tensor& vi, const Tensor& img,
    const Tensor& index_img, double max_dp_dr) {
  static auto op = c10::Dispatcher::singleton()
                       .findSchemaOrThrow("edge_grad_ext::edge_grad_estimator", "")
                       .typed<decltype(edge_grad_op)>();
  return op.call(v_pix, v_pix_img, vi, img, index_img, max_dp_dr);
}

class EdgeGradEstimatorFunction : public torch::autograd::Function<EdgeGradEstimatorFunction> {
 public:
  static tensor_list forward(
      AutogradContext* ctx, const Tensor& v_pix, const Tensor& v_pix_img, const Tensor& vi,
      const Tensor& img, const Tensor& index_img, double max_dp_dr) {
    if (v_pix.is_cuda()) {
      edge_grad_fwd_hip(v_pix, v_pix_img, vi, img, index_img, max_dp_dr);
    } else {
      no_cpu("edge_grad_estimator");
    }
    ctx->set_materialize_grads(false);
    ctx->save_for_backward({v_pix, img, index_img, vi});
    ctx->saved_data["v_pix_img_requires_grad"] = v_pix_img.requires_grad();
    ctx->saved_data["max_dp_dr"] = max_dp_dr;
    return {img}; // edge_grad_module.cpp:136
  }
  static tensor_list backward(AutogradContext* ctx, tensor_list grad_outputs) {
    // edge_grad_module.cpp:143-151 : passthrough when v_pix_img needs no gradient
    if (!ctx->saved_data["v_pix_img_requires_grad"].toBool() || !grad_outputs[0].defined()) {
      return {Tensor(), Tensor(), Tensor(), grad_outputs[0], Tensor(), Tensor()};
    }
    const auto saved = ctx->get_saved_variables();
    const double max_dp_dr = ctx->saved_data["max_dp_dr"].toDouble();
    auto g = edge_grad_backward_hip(saved[0], saved[1], saved[2], saved[3], grad_outputs[0], max_dp_dr);
    return {Tensor(), g, Tensor(), grad_outputs[0], Tensor(), Tensor()};
  }
};

Tensor edge_grad_autograd(
    const Tensor& v_pix, const Tensor& v_pix_img, const Tensor& vi, const Tensor& img,
    const Tensor& index_img, double max_dp_dr) {
  return EdgeGradEstimatorFunction::apply(v_pix, v_pix_img, vi, img, index_img, max_dp_dr)[0];
}

Tensor edge_grad_autocast(
    const Tensor& v_pix, const Tensor& v_pix_img, const Tensor& vi, const Tensor& img,
    const Tensor& index_img, double max_dp_dr) {
  c10::impl::ExcludeDispatchKeyGuard no_autocast(c10::DispatchKey::Autocast);
  return edge_grad_op(
      at::autocast::cached_cast(at::kFloat, v_pix), at::autocast::cached_cast(at::kFloat, v_pix_img), vi,
      at::autocast::cached_cast(at::kFloat, img), index_img, max_dp_dr);
}

// ---------------------------------------------------------------------------------------------
// edge_grad_estimator_fused -- drtk_amd extension (NOT a reference op): the default route of
// drtk_amd.edge_grad_estimator when no v_pix_img hook is registered.  Same forward value (img), same
// gradient to v_pix as [interpolate(v_pix) -> edge_grad_estimator] in the reference graph
// (drtk/edge_grad_estimator.py:168-176), without the unused C=3 interpolate forward and without
// materialising grad_v_pix_img.
// ---------------------------------------------------------------------------------------------
Tensor edge_grad_fused_fwd_hip(
    const Tensor& v_pix, const Tensor& vi, const Tensor& bary_img, const Tensor& img,
    const Tensor& index_img, double max_dp_dr) {
  TORCH_CHECK(bary_img.defined() && bary_img.dim() == 4 && bary_img.size(1) == 3,
              "edge_grad_estimator(): expected bary_img of shape [N, 3, H, W]");
  TORCH_CHECK(bary_img.device() == v_pix.device() && bary_img.dtype() == v_pix.dtype(),
              "edge_grad_estimator(): expected bary_img on the device and of the dtype of v_pix");
  // bary_img has the shape of the v_pix_img the reference validates against
  return edge_grad_fwd_hip(v_pix, bary_img, vi, img, index_img, max_dp_dr);
}

Tensor edge_grad_fused_backward_hip(
    const Tensor& v_pix, const Tensor& img, const Tensor& index_img, const Tensor& vi,
    const Tensor& bary_img, const Tensor& grad_outputs, double max_dp_dr) {
  const drtk_dtype_t dt = dtype_of(v_pix, "edge_grad_estimator_backward");
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(v_pix.device());
  const auto v_c = v_pix.contiguous();
  const auto img_c = img.to(v_pix.scalar_type()).contiguous();
  const auto idx_c = index_img.contiguous();
  const auto bary_c = bary_img.contiguous();
  const auto go_c = grad_outputs.to(v_pix.scalar_type()).contiguous();
  const ViArg via = prep_vi(vi);
  const int64_t N = img.size(0), C = img.size(1), H = img.size(2), W = img.size(3), V = v_pix.size(1), F = vi.size(1);
  auto grad_v_pix = out_empty({N, V, 3}, v_pix.options()); // zero-filled by the call
  size_t ws_bytes = 0;
  check_status(drtk_amd_edge_grad_backward_fused_workspace_bytes(dt, N, H, W, &ws_bytes), "edge_grad_estimator");
  auto ws = alloc_workspace(ws_bytes, v_pix);
  check_status(
      drtk_amd_edge_grad_backward_fused(
          dt, v_c.data_ptr(), img_c.data_ptr(), idx_c.data_ptr<int32_t>(), via.ptr, bary_c.data_ptr(),
          go_c.data_ptr(), N, V, C, F, via.sN, H, W, max_dp_dr, grad_v_pix.data_ptr(), ws.data_ptr(), ws_bytes,
          current_stream(v_pix)),
      "edge_grad_estimator");
  return grad_v_pix;
}

Tensor edge_grad_fused_cpu(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&, double) {
  no_cpu("edge_grad_estimator");
}

Tensor edge_grad_fused_op(
    const Tensor& v_pix, const Tensor& vi, const Tensor& bary_img, const Tensor& img,
    const Tensor& index_img, double max_dp_dr) {
  static auto op = c10::Dispatcher::singleton()
                       .findSchemaOrThrow("edge_grad_ext::edge_grad_estimator_fused", "")
                       .typed<decltype(edge_grad_fused_op)>();
  return op.call(v_pix, vi, bary_img, img, index_img, max_dp_dr);
}

class EdgeGradEstimatorFusedFunction : public torch::autograd::Function<EdgeGradEstimatorFusedFunction> {
 public:
  static tensor_list forward(
      AutogradContext* ctx, const Tensor& v_pix, const Tensor& vi, const Tensor& bary_img,
      const Tensor& img, const Tensor& index_img, double max_dp_dr) {
    if (v_pix.is_cuda()) {
      edge_grad_fused_fwd_hip(v_pix, vi, bary_img, img, index_img, max_dp_dr);
    } else {
      no_cpu("edge_grad_estimator");
    }
    ctx->set_materialize_grads(false);
    ctx->save_for_backward({v_pix, img, index_img, vi, bary_img});
    ctx->saved_data["v_pix_requires_grad"] = v_pix.requires_grad();
    ctx->saved_data["max_dp_dr"] = max_dp_dr;
    return {img};
  }
  static tensor_list backward(AutogradContext* ctx, tensor_list grad_outputs) {
    if (!ctx->saved_data["v_pix_requires_grad"].toBool() || !grad_outputs[0].defined()) {
      return {Tensor(), Tensor(), Tensor(), grad_outputs[0], Tensor(), Tensor()};
    }
    const auto saved = ctx->get_saved_variables();
    const double max_dp_dr = ctx->saved_data["max_dp_dr"].toDouble();
    auto g = edge_grad_fused_backward_hip(saved[0], saved[1], saved[2], saved[3], saved[4], grad_outputs[0], max_dp_dr);
    return {g, Tensor(), Tensor(), grad_outputs[0], Tensor(), Tensor()};
  }
};

Tensor edge_grad_fused_autograd(
    const Tensor& v_pix, const Tensor& vi, const Tensor& bary_img, const Tensor& img,
    const Tensor& index_img, double max_dp_dr) {
  return EdgeGradEstimatorFusedFunction::apply(v_pix, vi, bary_img, img, index_img, max_dp_dr)[0];
}

Tensor edge_grad_fused_autocast(
    const Tensor& v_pix, const Tensor& vi, const Tensor& bary_img, const Tensor& img,
    const Tensor& index_img, double max_dp_dr) {
  c10::impl::ExcludeDispatchKeyGuard no_autocast(c10::DispatchKey::Autocast);
  return edge_grad_fused_op(
      at::autocast::cached_cast(at::kFloat, v_pix), vi, at::autocast::cached_cast(at::kFloat, bary_img),
      at::autocast::cached_cast(at::kFloat, img), index_img, max_dp_dr);
}

} // namespace

// schema: verbatim from the reference
TORCH_LIBRARY(edge_grad_ext, m) {
  m.def(
      "edge_grad_estimator(Tensor v_pix, Tensor v_pix_img, Tensor vi, Tensor img, Tensor index_img, float max_dp_dr=1e4) -> Tensor");
  // drtk_amd extension, see EdgeGradEstimatorFusedFunction
  m.def(
      "edge_grad_estimator_fused(Tensor v_pix, Tensor vi, Tensor bary_img, Tensor img, Tensor index_img, float max_dp_dr=1e4) -> Tensor");
}
TORCH_LIBRARY_IMPL(edge_grad_ext, Autograd, m) {
  m.impl("edge_grad_estimator", &edge_grad_autograd);
  m.impl("edge_grad_estimator_fused", &edge_grad_fused_autograd);
}
TORCH_LIBRARY_IMPL(edge_grad_ext, Autocast, m) {
  m.impl("edge_grad_estimator", edge_grad_autocast);
  m.impl("edge_grad_estimator_fused", edge_grad_fused_autocast);
}
TORCH_LIBRARY_IMPL(edge_grad_ext, CUDA, m) {
  m.impl("edge_grad_estimator", &edge_grad_fwd_hip);
  m.impl("edge_grad_estimator_fused", &edge_grad_fused_fwd_hip);
}
TORCH_LIBRARY_IMPL(edge_grad_ext, CPU, m) {
  m.impl("edge_grad_estimator", &edge_grad_cpu);
  m.impl("edge_grad_estimator_fused", &edge_grad_fused_cpu);
}

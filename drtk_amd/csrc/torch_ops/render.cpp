// render_ext::render -- src/render/render_module.cpp:16-107 over drtk_amd_render / drtk_amd_render_backward.
#include "common.hpp"

namespace {
using namespace drtk_amd_torch;

// ---------------------------------------------------------------------------------------------
// render
// ---------------------------------------------------------------------------------------------
void render_checks(const Tensor& v, const Tensor& vi, const Tensor& index_img) {
  // render_kernel.cu:285-336
  TORCH_CHECK(v.defined() && vi.defined() && index_img.defined(), "render(): expected all inputs to be defined");
  TORCH_CHECK(
      (v.device() == vi.device()) && (v.device() == index_img.device()) && v.is_cuda(),
      "render(): expected all inputs to be on same cuda device");
  TORCH_CHECK(v.is_floating_point(), "render(): expected v to have floating point type, but v has ", v.dtype());
  TORCH_CHECK(vi.dtype() == at::kInt, "render(): expected vi to have int32 type, but vi has ", vi.dtype());
  TORCH_CHECK(
      index_img.dtype() == at::kInt,
      "render(): expected index_img to have int32 type, but index_img has ", index_img.dtype());
  TORCH_CHECK(
      v.layout() == at::kStrided && vi.layout() == at::kStrided && index_img.layout() == at::kStrided,
      "render(): expected all inputs to have torch.strided layout");
  TORCH_CHECK(
      (v.dim() == 3) && (vi.dim() == 3) && (index_img.dim() == 3),
      "render(): expected v.ndim == 3, vi.ndim == 3, index_img.ndim == 3, but got v with sizes ", v.sizes(),
      " and vi with sizes ", vi.sizes(), " and index_img with sizes ", index_img.sizes());
  TORCH_CHECK(
      v.size(0) == index_img.size(0),
      "render(): expected v and index_img to have same batch size, but got v with sizes ", v.sizes(),
      " and index_img with sizes ", index_img.sizes());
  TORCH_CHECK(
      vi.size(0) == v.size(0),
      "render(): expected first dim of vi to match first dim of v but got ", v.size(0),
      " in first dim of v, and ", vi.size(0), " in the first dim of vi");
  TORCH_CHECK(
      v.size(2) == 3 && vi.size(2) == 3,
      "render(): expected third dim of v and vi to be 3, but got ", v.size(2), " and ", vi.size(2));
}

std::vector<Tensor> render_hip(const Tensor& v, const Tensor& vi, const Tensor& index_img) {
  render_checks(v, vi, index_img);
  const drtk_dtype_t dt = dtype_of(v, "render");
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(v.device());
  const auto v_c = v.contiguous();
  const auto idx_c = index_img.contiguous();
  const ViArg via = prep_vi(vi);
  const int64_t N = v.size(0), V = v.size(1), F = vi.size(1), H = index_img.size(1), W = index_img.size(2);
  auto depth_img = out_empty({N, H, W}, v.options());
  auto bary_img = out_empty({N, 3, H, W}, v.options());
  check_status(
      drtk_amd_render(
          dt, v_c.data_ptr(), via.ptr, idx_c.data_ptr<int32_t>(), N, V, F, via.sN, H, W,
          depth_img.data_ptr(), bary_img.data_ptr(), current_stream(v)),
      "render");
  return {depth_img, bary_img};
}

Tensor render_backward_hip(
    const Tensor& v, const Tensor& vi, const Tensor& index_img, const Tensor& grad_depth_img,
    const Tensor& grad_bary_img) {
  const drtk_dtype_t dt = dtype_of(v, "render_backward");
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(v.device());
  const auto v_c = v.contiguous();
  const auto idx_c = index_img.contiguous();
  const ViArg via = prep_vi(vi);
  const int64_t N = v.size(0), V = v.size(1), F = vi.size(1), H = index_img.size(1), W = index_img.size(2);
  const auto gd = grad_depth_img.to(v.scalar_type()).contiguous();
  const auto gb = grad_bary_img.to(v.scalar_type()).contiguous();
  auto grad_v = out_empty({N, V, 3}, v.options()); // zero-filled by the call
  check_status(
      drtk_amd_render_backward(
          dt, v_c.data_ptr(), via.ptr, idx_c.data_ptr<int32_t>(), gd.data_ptr(), gb.data_ptr(), N, V,
          F, via.sN, H, W, grad_v.data_ptr(), current_stream(v)),
      "render_backward");
  return grad_v;
}

std::vector<Tensor> render_cpu(const Tensor&, const Tensor&, const Tensor&) {
  no_cpu("render");
}

tensor_list render_op(const Tensor& v, const Tensor& vi, const Tensor& index_img) {
  static auto op = c10::Dispatcher::singleton()
                       .findSchemaOrThrow("render_ext::render", "")
                       .typed<decltype(render_op)>();
  return op.call(v, vi, index_img);
}

class RenderFunction : public torch::autograd::Function<RenderFunction> {
 public:
  static tensor_list forward(AutogradContext* ctx, const Tensor& v, const Tensor& vi, const Tensor& index_img) {
    // grads stay materialised: an unused depth/bary output arrives as zeros (render_module.cpp:34)
    ctx->save_for_backward({v, vi, index_img});
    ctx->saved_data["requires_grad"] = v.requires_grad(); // render_module.cpp:41
    at::AutoDispatchBelowADInplaceOrView g;
    return render_op(v, vi, index_img);
  }
  static tensor_list backward(AutogradContext* ctx, tensor_list grad_outputs) {
    if (!ctx->saved_data["requires_grad"].toBool()) return {Tensor(), Tensor(), Tensor()};
    const auto saved = ctx->get_saved_variables();
    auto grad_v = render_backward_hip(saved[0], saved[1], saved[2], grad_outputs[0], grad_outputs[1]);
    return {grad_v, Tensor(), Tensor()};
  }
};

tensor_list render_autograd(const Tensor& v, const Tensor& vi, const Tensor& index_img) {
  return RenderFunction::apply(v, vi, index_img);
}

tensor_list render_autocast(const Tensor& v, const Tensor& vi, const Tensor& index_img) {
  c10::impl::ExcludeDispatchKeyGuard no_autocast(c10::DispatchKey::Autocast);
  return render_op(at::autocast::cached_cast(at::kFloat, v), vi, index_img);
}

} // namespace

// schema: verbatim from the reference
TORCH_LIBRARY(render_ext, m) {
  m.def("render(Tensor v, Tensor vi, Tensor index_img) -> Tensor[]");
}
TORCH_LIBRARY_IMPL(render_ext, Autograd, m) {
  m.impl("render", &render_autograd);
}
TORCH_LIBRARY_IMPL(render_ext, Autocast, m) {
  m.impl("render", render_autocast);
}
TORCH_LIBRARY_IMPL(render_ext, CUDA, m) {
  m.impl("render", &render_hip);
}
TORCH_LIBRARY_IMPL(render_ext, CPU, m) {
  m.impl("render", &render_cpu);
}

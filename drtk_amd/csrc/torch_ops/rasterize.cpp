// rasterize_ext::rasterize -- schema, dispatch keys and autograd contract of src/rasterize/rasterize_module.cpp:16-95,
// over drtk_amd_rasterize (include/drtk_amd.h).  Host-only C++; no CPU compute path (the CPU key raises).
#include "common.hpp"

namespace {
using namespace drtk_amd_torch;

// ---------------------------------------------------------------------------------------------
// rasterize
// ---------------------------------------------------------------------------------------------
std::vector<Tensor> rasterize_hip(
    const Tensor& v, const Tensor& vi, int64_t height, int64_t width, bool wireframe) {
  // checks and messages follow rasterize_kernel.cu:423-468
  TORCH_CHECK(v.defined() && vi.defined(), "rasterize(): expected all inputs to be defined");
  TORCH_CHECK(
      (v.device() == vi.device()) && v.is_cuda(),
      "rasterize(): expected all inputs to be on same cuda device");
  TORCH_CHECK(v.is_floating_point(), "rasterize(): expected v to have floating point type, but v has ", v.dtype());
  TORCH_CHECK(vi.dtype() == at::kInt, "rasterize(): expected vi to have int32 type, but vi has ", vi.dtype());
  TORCH_CHECK(
      v.layout() == at::kStrided && vi.layout() == at::kStrided,
      "rasterize(): expected all inputs to have torch.strided layout");
  TORCH_CHECK(
      (v.dim() == 3) && (vi.dim() == 3),
      "rasterize(): expected v.ndim == 3, vi.ndim == 3, but got v with sizes ", v.sizes(),
      " and vi with sizes ", vi.sizes());
  TORCH_CHECK(
      v.size(2) == 3 && vi.size(2) == 3,
      "rasterize(): expected third dim of v and last dim of vi to be 3, but got ", v.size(2), " and ", vi.size(2));
  TORCH_CHECK(
      vi.size(0) == v.size(0),
      "rasterize(): expected first dim of vi to match first dim of v, but got ", v.size(0), " and ", vi.size(0));
  TORCH_CHECK(
      v.size(1) < 0x10000000LL,
      "rasterize(): expected second dim of v to be less than 268435456, but got ", v.size(1));
  TORCH_CHECK(
      height > 0 && width > 0,
      "rasterize(): both height and width must be > 0, but got height: ", height, ", width: ", width);
  const drtk_dtype_t dt = dtype_of(v, "rasterize");

  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(v.device());
  const auto v_c = v.contiguous();
  const ViArg via = prep_vi(vi);
  const int64_t N = v.size(0), V = v.size(1), F = vi.size(1);
  auto depth_img = out_empty({N, height, width}, v.options().dtype(at::kFloat));
  auto index_img = out_empty({N, height, width}, v.options().dtype(at::kInt));
  size_t ws_bytes = 0;
  check_status(
      wireframe ? drtk_amd_rasterize_lines_workspace_bytes(N, height, width, &ws_bytes)
                : drtk_amd_rasterize_workspace_bytes(N, F, height, width, &ws_bytes),
      "rasterize");
  auto ws = alloc_workspace(ws_bytes, v);
  check_status(
      drtk_amd_rasterize(
          dt, v_c.data_ptr(), via.ptr, N, V, F, via.sN, height, width, wireframe ? 1 : 0, depth_img.data_ptr<float>(),
          index_img.data_ptr<int32_t>(), ws.data_ptr(), ws_bytes, current_stream(v)),
      "rasterize");
  return {depth_img, index_img};
}

std::vector<Tensor> rasterize_cpu(const Tensor&, const Tensor&, int64_t, int64_t, bool) {
  no_cpu("rasterize");
}

tensor_list rasterize_op(const Tensor& v, const Tensor& vi, int64_t height, int64_t width, bool wireframe) {
  static auto op = c10::Dispatcher::singleton()
                       .findSchemaOrThrow("rasterize_ext::rasterize", "")
                       .typed<decltype(rasterize_op)>();
  return op.call(v, vi, height, width, wireframe);
}

// Autograd key: rasterize has no gradient (rasterize_module.cpp:43 marks both outputs non-differentiable).  The reference
// says so through an autograd::Function whose backward returns nothing; here no node is built at all -- the outputs
// come back with requires_grad == False and no grad_fn, which is what mark_non_differentiable leaves behind, without
// the node's bookkeeping on every call (a few microseconds of a small scene's launch-bound step).
tensor_list rasterize_autograd(const Tensor& v, const Tensor& vi, int64_t height, int64_t width, bool wireframe) {
  at::AutoDispatchBelowADInplaceOrView g;
  return rasterize_op(v, vi, height, width, wireframe);
}

tensor_list rasterize_autocast(const Tensor& v, const Tensor& vi, int64_t height, int64_t width, bool wireframe) {
  c10::impl::ExcludeDispatchKeyGuard no_autocast(c10::DispatchKey::Autocast);
  return rasterize_op(at::autocast::cached_cast(at::kFloat, v), vi, height, width, wireframe);
}

} // namespace

// schema: verbatim from the reference
TORCH_LIBRARY(rasterize_ext, m) {
  m.def("rasterize(Tensor v, Tensor vi, int height, int width, bool wireframe) -> Tensor[]");
}
TORCH_LIBRARY_IMPL(rasterize_ext, Autograd, m) {
  m.impl("rasterize", &rasterize_autograd);
}
TORCH_LIBRARY_IMPL(rasterize_ext, Autocast, m) {
  m.impl("rasterize", rasterize_autocast);
}
TORCH_LIBRARY_IMPL(rasterize_ext, CUDA, m) {
  m.impl("rasterize", &rasterize_hip);
}
TORCH_LIBRARY_IMPL(rasterize_ext, CPU, m) {
  m.impl("rasterize", &rasterize_cpu);
}

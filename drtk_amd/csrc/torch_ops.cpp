// Torch-operator shim over the drtk_amd C ABI (include/drtk_amd.h).
//
// Registers the reference's four operator schemas VERBATIM, under the same namespaces, with the
// same dispatch keys and the same autograd contracts, so that `drtk.*` Python code and any caller
// of `torch.ops.<name>_ext.<op>` runs unchanged under PyTorch-ROCm:
//
//   rasterize_ext::rasterize          src/rasterize/rasterize_module.cpp:16-95
//   render_ext::render                src/render/render_module.cpp:16-107
//   interpolate_ext::interpolate      src/interpolate/interpolate_module.cpp:376-433,584-669
//   interpolate_ext::interpolation_matrix / interpolation_normal_matrix[_values]   :28-310,435-669
//   edge_grad_ext::edge_grad_estimator   src/edge_grad/edge_grad_module.cpp:18-224
//   mipmap_grid_sampler_ext::mipmap_grid_sampler_2d   src/mipmap_grid_sampler/mipmap_grid_sampler_module.cpp:16-266
//
// This file is host-only C++ (no device code); the kernels live in libdrtk_amd.so.  There is NO
// CPU compute path: the CPU key is registered only to fail with a clear message.
#include <cstdlib>
#include <limits>
#include <ATen/ATen.h>
#include <ATen/autocast_mode.h>
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <c10/hip/HIPStream.h>
#include <torch/csrc/autograd/custom_function.h>
#include <torch/library.h>

#include <array>
#include <limits>
#include <list>
#include <mutex>
#include <unordered_map>

#include "drtk_amd.h"

namespace {

using at::Tensor;
using torch::autograd::AutogradContext;
using torch::autograd::tensor_list;

drtk_dtype_t dtype_of(const Tensor& t, const char* op) {
  switch (t.scalar_type()) {
    case at::kFloat:
      return DRTK_F32;
    case at::kDouble:
      return DRTK_F64;
    default:
      // src/include/kernel_utils.h:35-57 : float and double only
      TORCH_CHECK(false, "\"", op, "\" not implemented for '", toString(t.scalar_type()), "'");
  }
}

drtk_stream_t current_stream(const Tensor& t) {
  return static_cast<drtk_stream_t>(c10::hip::getCurrentHIPStream(t.device().index()).stream());
}

void check_status(int status, const char* op) {
  TORCH_CHECK(status == DRTK_OK, op, "(): ", drtk_amd_status_string(status), " [drtk_amd status ", status, "]");
}

// vi arrives as [N,F,3]; the Python wrappers build it with a stride-0 expand from [F,3]
// (drtk/rasterize.py:61-62).  Keep that broadcast instead of materialising N copies.
struct ViArg {
  Tensor holder;
  const int32_t* ptr;
  int64_t sN;
};
ViArg prep_vi(const Tensor& vi) {
  ViArg a;
  if (vi.size(0) > 1 && vi.stride(0) == 0) {
    a.holder = vi.select(0, 0).contiguous();
    a.sN = 0;
  } else {
    a.holder = vi.contiguous();
    a.sN = vi.size(1) * 3;
  }
  a.ptr = a.holder.data_ptr<int32_t>();
  return a;
}

// Output allocation of the shim: uninitialised memory, or -- with DRTK_CAPI_POISON=1 in the environment, which the test
// suite and the fuzzers set -- memory pre-filled with NaN / a large negative integer / 0xA5 bytes, so that an element a
// kernel forgot to write cannot pass for a value (freshly allocated device memory reads as zeros, a plausible image:
// DESIGN.md 3.1, round 3).  The product never pays for it: one getenv at load time.
const bool g_poison_outputs = [] {
  const char* e = std::getenv("DRTK_CAPI_POISON");
  return e && e[0] && !(e[0] == '0' && !e[1]);
}();
Tensor out_empty(at::IntArrayRef sizes, const at::TensorOptions& opts) {
  Tensor t = at::empty(sizes, opts);
  if (g_poison_outputs && t.numel() > 0) {
    if (at::isFloatingType(t.scalar_type())) t.fill_(std::numeric_limits<double>::quiet_NaN());
    else if (t.scalar_type() == at::kByte) t.fill_(0xA5);
    else t.fill_(-(1 << 30) - 7);
  }
  return t;
}

Tensor alloc_workspace(size_t bytes, const Tensor& like) {
  return out_empty({static_cast<int64_t>(bytes)}, like.options().dtype(at::kByte));
}

[[noreturn]] void no_cpu(const char* op) {
  TORCH_CHECK(false, op, "(): drtk_amd implements the MI355X (HIP) path only; got CPU tensors");
}

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

class RasterizeFunction : public torch::autograd::Function<RasterizeFunction> {
 public:
  static tensor_list forward(
      AutogradContext* ctx, const Tensor& v, const Tensor& vi, int64_t height, int64_t width, bool wireframe) {
    ctx->set_materialize_grads(false);
    at::AutoDispatchBelowADInplaceOrView g;
    auto outputs = rasterize_op(v, vi, height, width, wireframe);
    ctx->mark_non_differentiable(outputs); // rasterize_module.cpp:43
    return outputs;
  }
  static tensor_list backward(AutogradContext*, const tensor_list&) {
    return {Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

tensor_list rasterize_autograd(const Tensor& v, const Tensor& vi, int64_t height, int64_t width, bool wireframe) {
  return RasterizeFunction::apply(v, vi, height, width, wireframe);
}

tensor_list rasterize_autocast(const Tensor& v, const Tensor& vi, int64_t height, int64_t width, bool wireframe) {
  c10::impl::ExcludeDispatchKeyGuard no_autocast(c10::DispatchKey::Autocast);
  return rasterize_op(at::autocast::cached_cast(at::kFloat, v), vi, height, width, wireframe);
}

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
  check_status(
      drtk_amd_interpolate_backward(
          dt, go_c.data_ptr(), a_c.data_ptr(), via.ptr, idx_c.data_ptr<int32_t>(), bary_c.data_ptr(), N, V,
          C, F, via.sN, H, W, vert_requires_grad ? vert_grad.data_ptr() : nullptr,
          bary_requires_grad ? bary_grad.data_ptr() : nullptr, current_stream(a)),
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

// ---------------------------------------------------------------------------------------------
// sparse interpolation operators (interpolate_module.cpp:28-310,435-583 ; interpolate_kernel.cu:699-905)
// ---------------------------------------------------------------------------------------------
using Tensor3 = std::tuple<Tensor, Tensor, Tensor>;
using Tensor4 = std::tuple<Tensor, Tensor, Tensor, Tensor>;

// pair_indices arrives as [N,F,9]; a stride-0 batch (shared topology) is kept as one copy.
ViArg prep_pairs(const Tensor& pair_indices) {
  ViArg a;
  if (pair_indices.size(0) > 1 && pair_indices.stride(0) == 0) {
    a.holder = pair_indices.select(0, 0).contiguous();
    a.sN = 0;
  } else {
    a.holder = pair_indices.contiguous();
    a.sN = pair_indices.size(1) * 9;
  }
  a.ptr = a.holder.data_ptr<int32_t>();
  return a;
}

Tensor4 interpolation_matrix_hip(const Tensor& vi, const Tensor& index_img, const Tensor& bary_img) {
  // interpolate_kernel.cu:703-729
  TORCH_CHECK(
      vi.defined() && index_img.defined() && bary_img.defined(),
      "interpolation_matrix(): expected all inputs to be defined");
  TORCH_CHECK(
      vi.device() == index_img.device() && vi.device() == bary_img.device(),
      "interpolation_matrix(): expected all inputs to be on same device");
  TORCH_CHECK(vi.dtype() == at::kInt, "interpolation_matrix(): expected vi to have int32 type, but vi has ", vi.dtype());
  TORCH_CHECK(
      index_img.dtype() == at::kInt,
      "interpolation_matrix(): expected index_img to have int32 type, but index_img has ", index_img.dtype());
  TORCH_CHECK(
      bary_img.is_floating_point(),
      "interpolation_matrix(): expected bary_img to have floating point type, but has ", bary_img.dtype());
  TORCH_CHECK(
      vi.dim() == 3 && index_img.dim() == 3 && bary_img.dim() == 4,
      "interpolation_matrix(): expected vi.ndim == 3, index_img.ndim == 3, bary_img.ndim == 4");
  TORCH_CHECK(
      vi.size(0) == index_img.size(0) && vi.size(0) == bary_img.size(0) && vi.size(2) == 3 &&
          bary_img.size(1) == 3 && index_img.size(1) == bary_img.size(2) && index_img.size(2) == bary_img.size(3),
      "interpolation_matrix(): expected vi, index_img and bary_img shapes to agree");
  const drtk_dtype_t dt = dtype_of(bary_img, "interpolation_matrix");
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(bary_img.device());
  const auto idx_c = index_img.contiguous();
  const auto bary_c = bary_img.contiguous();
  const ViArg via = prep_vi(vi);
  // The row count is data dependent (one row per foreground pixel), so this op synchronises
  // exactly like the reference's (at::nonzero, interpolate_kernel.cu:736-737).
  auto row_pixels = at::nonzero(idx_c.reshape({-1}).ne(-1)).reshape({-1});
  const int64_t R = row_pixels.numel();
  const auto long_opts = index_img.options().dtype(at::kLong);
  auto crow = at::arange(0, R * 3 + 1, 3, long_opts);
  auto col = out_empty({R * 3}, long_opts);
  auto values = out_empty({R * 3}, bary_img.options());
  check_status(
      drtk_amd_interpolation_matrix(
          dt, via.ptr, idx_c.data_ptr<int32_t>(), bary_c.data_ptr(), row_pixels.data_ptr<int64_t>(), R,
          index_img.size(0), vi.size(1), via.sN, index_img.size(1), index_img.size(2), col.data_ptr<int64_t>(),
          values.data_ptr(), current_stream(bary_img)),
      "interpolation_matrix");
  return {crow, col, values, row_pixels};
}

Tensor interpolation_matrix_backward_hip(
    const Tensor& grad_values, const Tensor& vi, const Tensor& index_img, const Tensor& bary_img,
    const Tensor& row_pixels) {
  const drtk_dtype_t dt = dtype_of(bary_img, "interpolation_matrix_backward");
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(bary_img.device());
  const auto idx_c = index_img.contiguous();
  const auto g_c = grad_values.to(bary_img.scalar_type()).contiguous();
  const auto rp_c = row_pixels.contiguous();
  const ViArg via = prep_vi(vi);
  const int64_t N = index_img.size(0), H = index_img.size(1), W = index_img.size(2);
  auto bary_grad = out_empty({N, 3, H, W}, bary_img.options()); // zero-filled by the call
  check_status(
      drtk_amd_interpolation_matrix_backward(
          dt, g_c.data_ptr(), via.ptr, idx_c.data_ptr<int32_t>(), rp_c.data_ptr<int64_t>(), rp_c.numel(), N,
          vi.size(1), via.sN, H, W, bary_grad.data_ptr(), current_stream(bary_img)),
      "interpolation_matrix_backward");
  return bary_grad;
}

Tensor normal_matrix_values_hip(
    const Tensor& pair_indices, const Tensor& index_img, const Tensor& bary_img, int64_t nnz) {
  // interpolate_kernel.cu:813-836
  TORCH_CHECK(
      pair_indices.defined() && index_img.defined() && bary_img.defined(),
      "interpolation_normal_matrix_values(): expected all inputs to be defined");
  TORCH_CHECK(
      pair_indices.device() == index_img.device() && pair_indices.device() == bary_img.device(),
      "interpolation_normal_matrix_values(): expected all inputs to be on same device");
  TORCH_CHECK(
      pair_indices.dtype() == at::kInt, "interpolation_normal_matrix_values(): expected pair_indices to have int32 type");
  TORCH_CHECK(
      index_img.dtype() == at::kInt, "interpolation_normal_matrix_values(): expected index_img to have int32 type");
  TORCH_CHECK(
      bary_img.is_floating_point(),
      "interpolation_normal_matrix_values(): expected bary_img to have floating point type");
  TORCH_CHECK(
      pair_indices.dim() == 3 && pair_indices.size(2) == 9 && index_img.dim() == 3 && bary_img.dim() == 4 &&
          bary_img.size(1) == 3,
      "interpolation_normal_matrix_values(): expected pair_indices [N,F,9], index_img [N,H,W], bary_img [N,3,H,W]");
  TORCH_CHECK(
      pair_indices.size(0) == index_img.size(0) && pair_indices.size(0) == bary_img.size(0) &&
          index_img.size(1) == bary_img.size(2) && index_img.size(2) == bary_img.size(3),
      "interpolation_normal_matrix_values(): expected pair_indices, index_img and bary_img shapes to agree");
  TORCH_CHECK(nnz >= 0, "interpolation_normal_matrix_values(): expected nnz to be non-negative");
  const drtk_dtype_t dt = dtype_of(bary_img, "interpolation_normal_matrix_values");
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(bary_img.device());
  const auto idx_c = index_img.contiguous();
  const auto bary_c = bary_img.contiguous();
  const ViArg pa = prep_pairs(pair_indices);
  auto values = out_empty({nnz}, bary_img.options()); // zero-filled by the call
  check_status(
      drtk_amd_interpolation_normal_matrix_values(
          dt, pa.ptr, idx_c.data_ptr<int32_t>(), bary_c.data_ptr(), index_img.size(0), pair_indices.size(1), pa.sN,
          index_img.size(1), index_img.size(2), nnz, values.data_ptr(), current_stream(bary_img)),
      "interpolation_normal_matrix_values");
  return values;
}

Tensor normal_matrix_values_backward_hip(
    const Tensor& grad_values, const Tensor& pair_indices, const Tensor& index_img, const Tensor& bary_img) {
  const drtk_dtype_t dt = dtype_of(bary_img, "interpolation_normal_matrix_values_backward");
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(bary_img.device());
  const auto idx_c = index_img.contiguous();
  const auto bary_c = bary_img.contiguous();
  const auto g_c = grad_values.to(bary_img.scalar_type()).contiguous();
  const ViArg pa = prep_pairs(pair_indices);
  const int64_t N = index_img.size(0), H = index_img.size(1), W = index_img.size(2);
  auto bary_grad = N * H * W > 0 ? out_empty({N, 3, H, W}, bary_img.options()) : at::zeros({N, 3, H, W}, bary_img.options());
  check_status(
      drtk_amd_interpolation_normal_matrix_values_backward(
          dt, g_c.data_ptr(), pa.ptr, idx_c.data_ptr<int32_t>(), bary_c.data_ptr(), N, pair_indices.size(1), pa.sN,
          H, W, bary_grad.data_ptr(), current_stream(bary_img)),
      "interpolation_normal_matrix_values_backward");
  return bary_grad;
}

// ---- A^T A sparsity pattern: topology-only, built once per face-index tensor and cached ---------
//
// The reference sorts 9*N*F keys on the host after copying vi off the device
// (interpolate_module.cpp:128-241).  Here the pattern is built WHERE vi LIVES with device-wide
// sort/unique/searchsorted (rocPRIM under ATen on a HIP tensor; the same code runs on a CPU tensor,
// which is how the host-side tests exercise it):
//   key(n,f,i,j) = vi[n,f,i] * V + vi[n,f,j]
//   (uniq, inverse) = unique(keys)         -> col = uniq mod V ; pair_indices = inverse
//   crow            = lower_bound(uniq, row * V), row = 0..V
// A stride-0 (shared-topology) batch is analysed once and pair_indices is returned as a stride-0
// expand, so the value kernels read ONE [F,9] table that stays in L2.
struct NormalMatrixPattern {
  Tensor crow_indices; // int64 [V+1]
  Tensor col_indices; // int64 [nnz]
  Tensor pair_indices; // int32 [N,F,9] (stride-0 batch when vi's is)
};

NormalMatrixPattern build_normal_matrix_pattern(const Tensor& vi, int64_t num_vertices) {
  // interpolate_module.cpp:132-137,150-153,181-183,191-193
  TORCH_CHECK(num_vertices >= 0, "interpolation_normal_matrix(): expected num_vertices to be non-negative");
  TORCH_CHECK(
      num_vertices <= std::numeric_limits<int32_t>::max(),
      "interpolation_normal_matrix(): expected num_vertices to fit in int32");
  const int64_t N = vi.size(0), F = vi.size(1);
  TORCH_CHECK(
      num_vertices > 0 || N * F == 0,
      "interpolation_normal_matrix(): expected num_vertices to be positive when faces are present");
  const bool shared = N > 1 && vi.stride(0) == 0;
  const auto faces = (shared ? vi.detach().narrow(0, 0, 1) : vi.detach()).to(at::kLong); // [n,F,3]
  const auto long_opts = vi.options().dtype(at::kLong);
  NormalMatrixPattern out;
  if (faces.numel() == 0) {
    out.crow_indices = at::zeros({num_vertices + 1}, long_opts);
    out.col_indices = out_empty({0}, long_opts);
    out.pair_indices = out_empty({N, F, 9}, vi.options());
    return out;
  }
  TORCH_CHECK(
      !faces.lt(0).logical_or(faces.ge(num_vertices)).any().item<bool>(),
      "interpolation_normal_matrix(): vi contains a vertex index outside [0, num_vertices)");
  const auto keys = (faces.unsqueeze(3) * num_vertices + faces.unsqueeze(2)).reshape({-1}); // [n*F*9]
  const auto uq = at::_unique2(keys, /*sorted=*/true, /*return_inverse=*/true, /*return_counts=*/false);
  const Tensor& uniq = std::get<0>(uq);
  TORCH_CHECK(
      uniq.numel() <= std::numeric_limits<int32_t>::max(),
      "interpolation_normal_matrix(): normal matrix has too many nonzeros for int32 value indices");
  out.col_indices = at::remainder(uniq, num_vertices);
  out.crow_indices = at::searchsorted(uniq, at::arange(num_vertices + 1, long_opts) * num_vertices);
  auto pairs = std::get<1>(uq).to(at::kInt).reshape({faces.size(0), F, 9});
  out.pair_indices = shared ? pairs.expand({N, F, 9}) : pairs;
  return out;
}

// Cache: identity + version of the face-index tensor, as interpolate_module.cpp:36-113 keys it --
// no content hashing (that would synchronise); an in-place edit bumps the version counter and
// misses.  Each entry pins its vi so a recycled allocation cannot alias a stale entry; 128 entries, LRU.
struct TopologyKey {
  std::array<int64_t, 14> f;
  bool operator==(const TopologyKey& o) const {
    return f == o.f;
  }
};
struct TopologyKeyHash {
  size_t operator()(const TopologyKey& k) const {
    uint64_t h = 1469598103934665603ull; // FNV-1a over the fields
    for (int64_t x : k.f) {
      h ^= static_cast<uint64_t>(x);
      h *= 1099511628211ull;
    }
    return static_cast<size_t>(h);
  }
};
TopologyKey topology_key(const Tensor& vi, int64_t num_vertices) {
  TopologyKey k;
  k.f = {static_cast<int64_t>(vi.device().type()),
         static_cast<int64_t>(vi.device().index()),
         static_cast<int64_t>(reinterpret_cast<uintptr_t>(vi.storage().unsafeGetStorageImpl())),
         static_cast<int64_t>(reinterpret_cast<uintptr_t>(vi.data_ptr())),
         vi.size(0),
         vi.size(1),
         vi.size(2),
         vi.stride(0),
         vi.stride(1),
         vi.stride(2),
         vi.storage_offset(),
         static_cast<int64_t>(vi.scalar_type()),
         num_vertices,
         static_cast<int64_t>(vi.unsafeGetTensorImpl()->version_counter().current_version())};
  return k;
}

class NormalMatrixPatternCache {
 public:
  static constexpr size_t kCapacity = 128; // interpolate_module.cpp:62
  static NormalMatrixPatternCache& instance() {
    static NormalMatrixPatternCache c;
    return c;
  }
  NormalMatrixPattern get(const Tensor& vi, int64_t num_vertices) {
    const TopologyKey key = topology_key(vi, num_vertices);
    {
      std::lock_guard<std::mutex> lock(mu_);
      if (const NormalMatrixPattern* hit = touch(key)) {
        ++hits_;
        return *hit;
      }
    }
    // built outside the lock; a racing identical miss is resolved by the second lookup
    NormalMatrixPattern built = build_normal_matrix_pattern(vi, num_vertices);
    std::lock_guard<std::mutex> lock(mu_);
    if (const NormalMatrixPattern* hit = touch(key)) return *hit;
    ++misses_;
    while (lru_.size() >= kCapacity) {
      index_.erase(lru_.back().key);
      lru_.pop_back();
    }
    lru_.push_front(Entry{key, vi, built});
    index_.emplace(key, lru_.begin());
    return built;
  }
  std::vector<int64_t> stats() {
    std::lock_guard<std::mutex> lock(mu_);
    return {hits_, misses_, static_cast<int64_t>(lru_.size())};
  }
  void clear() {
    std::lock_guard<std::mutex> lock(mu_);
    lru_.clear();
    index_.clear();
    hits_ = misses_ = 0;
  }

 private:
  struct Entry {
    TopologyKey key;
    Tensor pinned_vi;
    NormalMatrixPattern pattern;
  };
  const NormalMatrixPattern* touch(const TopologyKey& key) {
    const auto it = index_.find(key);
    if (it == index_.end()) return nullptr;
    lru_.splice(lru_.begin(), lru_, it->second);
    return &it->second->pattern;
  }
  std::mutex mu_;
  std::list<Entry> lru_;
  std::unordered_map<TopologyKey, std::list<Entry>::iterator, TopologyKeyHash> index_;
  int64_t hits_ = 0, misses_ = 0;
};

void normal_matrix_checks(const Tensor& vi, const Tensor& index_img, const Tensor& bary_img) {
  // interpolate_module.cpp:272-301
  TORCH_CHECK(
      vi.defined() && index_img.defined() && bary_img.defined(),
      "interpolation_normal_matrix(): expected all inputs to be defined");
  TORCH_CHECK(
      vi.device() == index_img.device() && vi.device() == bary_img.device(),
      "interpolation_normal_matrix(): expected all inputs to be on same device");
  TORCH_CHECK(
      vi.dtype() == at::kInt, "interpolation_normal_matrix(): expected vi to have int32 type, but vi has ", vi.dtype());
  TORCH_CHECK(
      index_img.dtype() == at::kInt,
      "interpolation_normal_matrix(): expected index_img to have int32 type, but index_img has ", index_img.dtype());
  TORCH_CHECK(
      bary_img.is_floating_point(), "interpolation_normal_matrix(): expected bary_img to have floating point type");
  TORCH_CHECK(
      vi.layout() == at::kStrided && index_img.layout() == at::kStrided && bary_img.layout() == at::kStrided,
      "interpolation_normal_matrix(): expected all inputs to have torch.strided layout");
  TORCH_CHECK(
      vi.dim() == 3 && index_img.dim() == 3 && bary_img.dim() == 4 && vi.size(2) == 3 && bary_img.size(1) == 3,
      "interpolation_normal_matrix(): expected vi [N,F,3], index_img [N,H,W], bary_img [N,3,H,W]");
  TORCH_CHECK(
      vi.size(0) == index_img.size(0) && vi.size(0) == bary_img.size(0) && index_img.size(1) == bary_img.size(2) &&
          index_img.size(2) == bary_img.size(3),
      "interpolation_normal_matrix(): expected vi, index_img and bary_img shapes to agree");
}

Tensor4 normal_matrix_forward_with_pairs(
    const Tensor& vi, const Tensor& index_img, const Tensor& bary_img, int64_t num_vertices) {
  normal_matrix_checks(vi, index_img, bary_img);
  TORCH_CHECK(bary_img.is_cuda(), "interpolation_normal_matrix(): drtk_amd implements the MI355X (HIP) path only; got CPU tensors");
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(bary_img.device());
  const NormalMatrixPattern p = NormalMatrixPatternCache::instance().get(vi, num_vertices);
  auto values = normal_matrix_values_hip(p.pair_indices, index_img, bary_img, p.col_indices.numel());
  return {p.crow_indices, p.col_indices, values, p.pair_indices};
}

Tensor3 interpolation_normal_matrix_hip(
    const Tensor& vi, const Tensor& index_img, const Tensor& bary_img, int64_t num_vertices) {
  const auto out = normal_matrix_forward_with_pairs(vi, index_img, bary_img, num_vertices);
  return {std::get<0>(out), std::get<1>(out), std::get<2>(out)};
}

// Topology-only entry (extension): the cached pattern itself, on whatever device vi lives.
Tensor3 normal_matrix_structure_any(const Tensor& vi, int64_t num_vertices) {
  TORCH_CHECK(
      vi.defined() && vi.dtype() == at::kInt && vi.layout() == at::kStrided && vi.dim() == 3 && vi.size(2) == 3,
      "normal_matrix_structure(): expected vi to be a strided int32 tensor of shape [N,F,3]");
  const NormalMatrixPattern p = NormalMatrixPatternCache::instance().get(vi, num_vertices);
  return {p.crow_indices, p.col_indices, p.pair_indices};
}
std::vector<int64_t> normal_matrix_cache_stats() {
  return NormalMatrixPatternCache::instance().stats();
}
void normal_matrix_cache_clear() {
  NormalMatrixPatternCache::instance().clear();
}

Tensor4 interpolation_matrix_cpu(const Tensor&, const Tensor&, const Tensor&) {
  no_cpu("interpolation_matrix");
}
Tensor3 interpolation_normal_matrix_cpu(const Tensor&, const Tensor&, const Tensor&, int64_t) {
  no_cpu("interpolation_normal_matrix");
}
Tensor normal_matrix_values_cpu(const Tensor&, const Tensor&, const Tensor&, int64_t) {
  no_cpu("interpolation_normal_matrix_values");
}

Tensor4 interpolation_matrix_op(const Tensor& vi, const Tensor& index_img, const Tensor& bary_img) {
  static auto op = c10::Dispatcher::singleton()
                       .findSchemaOrThrow("interpolate_ext::interpolation_matrix", "")
                       .typed<decltype(interpolation_matrix_op)>();
  return op.call(vi, index_img, bary_img);
}
Tensor3 interpolation_normal_matrix_op(
    const Tensor& vi, const Tensor& index_img, const Tensor& bary_img, int64_t num_vertices) {
  static auto op = c10::Dispatcher::singleton()
                       .findSchemaOrThrow("interpolate_ext::interpolation_normal_matrix", "")
                       .typed<decltype(interpolation_normal_matrix_op)>();
  return op.call(vi, index_img, bary_img, num_vertices);
}
Tensor normal_matrix_values_op(
    const Tensor& pair_indices, const Tensor& index_img, const Tensor& bary_img, int64_t nnz) {
  static auto op = c10::Dispatcher::singleton()
                       .findSchemaOrThrow("interpolate_ext::interpolation_normal_matrix_values", "")
                       .typed<decltype(normal_matrix_values_op)>();
  return op.call(pair_indices, index_img, bary_img, nnz);
}

// interpolate_module.cpp:435-481
class InterpolationMatrixFunction : public torch::autograd::Function<InterpolationMatrixFunction> {
 public:
  static tensor_list forward(AutogradContext* ctx, const Tensor& vi, const Tensor& index_img, const Tensor& bary_img) {
    ctx->set_materialize_grads(false);
    Tensor4 fwd;
    {
      at::AutoDispatchBelowADInplaceOrView g;
      fwd = interpolation_matrix_op(vi, index_img, bary_img);
    }
    const Tensor &crow = std::get<0>(fwd), &col = std::get<1>(fwd), &values = std::get<2>(fwd), &rows = std::get<3>(fwd);
    ctx->save_for_backward({vi, index_img, bary_img, rows});
    ctx->mark_non_differentiable({crow, col, rows});
    return {crow, col, values, rows};
  }
  static tensor_list backward(AutogradContext* ctx, tensor_list grad_outputs) {
    const auto saved = ctx->get_saved_variables();
    const Tensor& bary_img = saved[2];
    tensor_list out(3);
    if (!bary_img.requires_grad() || grad_outputs.size() < 3 || !grad_outputs[2].defined()) return out;
    out[2] = interpolation_matrix_backward_hip(grad_outputs[2], saved[0], saved[1], bary_img, saved[3]);
    return out;
  }
};
Tensor4 interpolation_matrix_autograd(const Tensor& vi, const Tensor& index_img, const Tensor& bary_img) {
  auto out = InterpolationMatrixFunction::apply(vi, index_img, bary_img);
  return {out[0], out[1], out[2], out[3]};
}

// interpolate_module.cpp:483-535
class InterpolationNormalMatrixFunction : public torch::autograd::Function<InterpolationNormalMatrixFunction> {
 public:
  static tensor_list forward(
      AutogradContext* ctx, const Tensor& vi, const Tensor& index_img, const Tensor& bary_img, int64_t num_vertices) {
    ctx->set_materialize_grads(false);
    Tensor4 fwd;
    {
      at::AutoDispatchBelowADInplaceOrView g;
      fwd = normal_matrix_forward_with_pairs(vi, index_img, bary_img, num_vertices);
    }
    const Tensor &crow = std::get<0>(fwd), &col = std::get<1>(fwd), &values = std::get<2>(fwd);
    ctx->save_for_backward({std::get<3>(fwd), index_img, bary_img});
    ctx->mark_non_differentiable({crow, col});
    return {crow, col, values};
  }
  static tensor_list backward(AutogradContext* ctx, tensor_list grad_outputs) {
    const auto saved = ctx->get_saved_variables();
    const Tensor& bary_img = saved[2];
    tensor_list out(4);
    if (!bary_img.requires_grad() || grad_outputs.size() < 3 || !grad_outputs[2].defined()) return out;
    out[2] = normal_matrix_values_backward_hip(grad_outputs[2], saved[0], saved[1], bary_img);
    return out;
  }
};
Tensor3 interpolation_normal_matrix_autograd(
    const Tensor& vi, const Tensor& index_img, const Tensor& bary_img, int64_t num_vertices) {
  auto out = InterpolationNormalMatrixFunction::apply(vi, index_img, bary_img, num_vertices);
  return {out[0], out[1], out[2]};
}

// interpolate_module.cpp:537-582
class NormalMatrixValuesFunction : public torch::autograd::Function<NormalMatrixValuesFunction> {
 public:
  static tensor_list forward(
      AutogradContext* ctx, const Tensor& pair_indices, const Tensor& index_img, const Tensor& bary_img, int64_t nnz) {
    ctx->set_materialize_grads(false);
    ctx->save_for_backward({pair_indices, index_img, bary_img});
    at::AutoDispatchBelowADInplaceOrView g;
    return {normal_matrix_values_op(pair_indices, index_img, bary_img, nnz)};
  }
  static tensor_list backward(AutogradContext* ctx, tensor_list grad_outputs) {
    const auto saved = ctx->get_saved_variables();
    const Tensor& bary_img = saved[2];
    tensor_list out(4);
    if (!bary_img.requires_grad() || grad_outputs.empty() || !grad_outputs[0].defined()) return out;
    out[2] = normal_matrix_values_backward_hip(grad_outputs[0], saved[0], saved[1], bary_img);
    return out;
  }
};
Tensor normal_matrix_values_autograd(
    const Tensor& pair_indices, const Tensor& index_img, const Tensor& bary_img, int64_t nnz) {
  return NormalMatrixValuesFunction::apply(pair_indices, index_img, bary_img, nnz)[0];
}

// interpolate_module.cpp:596-625 : fp32 under autocast
Tensor4 interpolation_matrix_autocast(const Tensor& vi, const Tensor& index_img, const Tensor& bary_img) {
  c10::impl::ExcludeDispatchKeyGuard no_autocast(c10::DispatchKey::Autocast);
  return interpolation_matrix_op(vi, index_img, at::autocast::cached_cast(at::kFloat, bary_img));
}
Tensor3 interpolation_normal_matrix_autocast(
    const Tensor& vi, const Tensor& index_img, const Tensor& bary_img, int64_t num_vertices) {
  c10::impl::ExcludeDispatchKeyGuard no_autocast(c10::DispatchKey::Autocast);
  return interpolation_normal_matrix_op(vi, index_img, at::autocast::cached_cast(at::kFloat, bary_img), num_vertices);
}
Tensor normal_matrix_values_autocast(
    const Tensor& pair_indices, const Tensor& index_img, const Tensor& bary_img, int64_t nnz) {
  c10::impl::ExcludeDispatchKeyGuard no_autocast(c10::DispatchKey::Autocast);
  return normal_matrix_values_op(pair_indices, index_img, at::autocast::cached_cast(at::kFloat, bary_img), nnz);
}

// ---------------------------------------------------------------------------------------------
// mipmap_grid_sampler_2d (mipmap_grid_sampler_module.cpp:16-266 ; mipmap_grid_sampler_kernel.cu:899-1249)
// ---------------------------------------------------------------------------------------------
struct LevelArgs {
  std::vector<Tensor> holders;
  std::vector<const void*> ptrs;
  std::vector<int64_t> h, w, sn;
};
// A level whose views are contiguous [C,h,w] blocks goes to the kernels as it is, whatever its batch stride: the common
// case of ONE texture shared by all camera views ([1,C,h,w].expand(N, ...), stride 0) is not materialised N times per
// call (the reference indexes through the strides, mipmap_grid_sampler_kernel.cu:40,65).  Anything else is copied.
LevelArgs prep_levels(at::TensorList input) {
  LevelArgs a;
  for (const Tensor& t : input) {
    const int64_t view = t.size(1) * t.size(2) * t.size(3);
    const bool views_contiguous = t.stride(3) == 1 && t.stride(2) == t.size(3) && t.stride(1) == t.size(2) * t.size(3);
    const bool as_is = t.size(0) <= 1 ? t.is_contiguous() : (views_contiguous && (t.stride(0) == 0 || t.stride(0) >= view));
    a.holders.push_back(as_is ? t : t.contiguous());
    const Tensor& u = a.holders.back();
    a.ptrs.push_back(u.data_ptr());
    a.h.push_back(t.size(2));
    a.w.push_back(t.size(3));
    a.sn.push_back(u.size(0) > 1 ? u.stride(0) : view);
  }
  return a;
}

// A uv field [N,H,W,2] whose pixels are evenly spaced in memory is read in place: contiguous, or the channel-first image
// `interpolate` produces seen through permute(0, 2, 3, 1) (the reference reads grid through its strides,
// mipmap_grid_sampler_kernel.cu:430-445).  Anything else is made contiguous.  layout = {sN, sP, sC} for the C ABI.
struct GridArg {
  Tensor t;
  int64_t layout[3];
};
GridArg prep_grid(const Tensor& grid) {
  const int64_t N = grid.size(0), H = grid.size(1), W = grid.size(2), P = H * W;
  const int64_t sN = grid.stride(0), sH = grid.stride(1), sW = grid.stride(2), sC = grid.stride(3);
  const bool rows_ok = H <= 1 || sH == W * sW;
  const bool pixel_major = sC == 1 && sW == 2 && rows_ok && (N <= 1 || sN >= 2 * P);
  const bool channel_major = sW == 1 && sC >= P && rows_ok && (N <= 1 || sN >= sC + P);
  GridArg a;
  if (grid.size(3) == 2 && P > 0 && (pixel_major || channel_major)) {
    a.t = grid;
    a.layout[0] = N > 1 ? sN : 2 * P, a.layout[1] = sW, a.layout[2] = sC;
  } else {
    a.t = grid.contiguous();
    a.layout[0] = 2 * P, a.layout[1] = 2, a.layout[2] = 1;
  }
  return a;
}

Tensor mipmap_grid_sampler_2d_hip(
    at::TensorList input, const Tensor& grid, const Tensor& vt_dxdy_img, int64_t max_aniso, int64_t padding_mode,
    int64_t interpolation_mode, bool align_corners, bool force_max_ansio, bool clip_grad) {
  // mipmap_grid_sampler_kernel.cu:909-1000
  const int64_t mipmaps = static_cast<int64_t>(input.size());
  TORCH_CHECK(mipmaps >= 1, "mipmap_aniso_grid_sampler_2d(): expected input to have at least one mipmap level");
  TORCH_CHECK(mipmaps <= 11, "mipmap_aniso_grid_sampler_2d(): at most 11 mipmap levels are supported");
  TORCH_CHECK(
      input[0].defined() && grid.defined(),
      "mipmap_aniso_grid_sampler_2d(): expected input and grid to not be undefined, but input is ", input,
      " and grid is ", grid);
  const auto input_opt = input[0].options();
  const auto grid_opt = grid.options();
  TORCH_CHECK(
      input_opt.device() == grid_opt.device(),
      "mipmap_aniso_grid_sampler_2d(): expected input and grid to be on same device, but input is on ",
      input_opt.device(), " and grid is on ", grid_opt.device());
  TORCH_CHECK(
      input_opt.dtype() == grid_opt.dtype(),
      "mipmap_aniso_grid_sampler_2d(): expected input and grid to have same dtype, but input has ", input_opt.dtype(),
      " and grid has ", grid_opt.dtype());
  TORCH_CHECK(
      input_opt.layout() == at::kStrided && grid_opt.layout() == at::kStrided,
      "mipmap_aniso_grid_sampler_2d(): expected input and grid to have torch.strided layout, but input has ",
      input_opt.layout(), " and grid has ", grid_opt.layout());
  TORCH_CHECK(
      (input[0].dim() == 4) && input[0].dim() == grid.dim() && input[0].dim() + 1 == vt_dxdy_img.dim(),
      "mipmap_aniso_grid_sampler_2d(): expected 4D input and grid with same number of dimensions and 5D vt_dxdy_img, "
      "but got input with sizes ", input[0].sizes(), " and grid with sizes ", grid.sizes(),
      " and vt_dxdy_img with sizes ", vt_dxdy_img.sizes());
  TORCH_CHECK(
      input[0].size(0) == grid.size(0) && input[0].size(0) == vt_dxdy_img.size(0),
      "mipmap_aniso_grid_sampler_2d(): expected grid, vt_dxdy_img and input to have same batch size, but got input "
      "with sizes ", input[0].sizes(), " and grid with sizes ", grid.sizes(), " and vt_dxdy_img with sizes ",
      vt_dxdy_img.sizes());
  TORCH_CHECK(
      grid.size(-1) == input[0].dim() - 2, "mipmap_aniso_grid_sampler_2d(): expected grid to have size ",
      input[0].dim() - 2, " in last dimension, but got grid with sizes ", grid.sizes());
  TORCH_CHECK(
      vt_dxdy_img.size(-1) == input[0].dim() - 2 && vt_dxdy_img.size(-2) == input[0].dim() - 2,
      "mipmap_aniso_grid_sampler_2d(): expected vt_dxdy_img to have size ", input[0].dim() - 2,
      " in last two dimension, but got grid with sizes ", grid.sizes());
  TORCH_CHECK(
      vt_dxdy_img.size(1) == grid.size(1) && vt_dxdy_img.size(2) == grid.size(2) && vt_dxdy_img.device() == grid.device() &&
          vt_dxdy_img.dtype() == grid.dtype(),
      "mipmap_aniso_grid_sampler_2d(): expected vt_dxdy_img to match grid in device, dtype and spatial size");
  for (int64_t i = 1; i < mipmaps; i++) {
    TORCH_CHECK(
        input_opt.device() == input[i].options().device() && input_opt.dtype() == input[i].options().dtype() &&
            input_opt.layout() == input[i].options().layout() && input[0].dim() == input[i].dim() &&
            input[0].size(0) == input[i].size(0) && input[0].size(1) == input[i].size(1),
        "mipmap_aniso_grid_sampler_2d(): expected all inputs to have same device, dtype, layout, and first two "
        "dimensions");
  }
  for (int64_t l = 0; l < mipmaps; l++) {
    for (int64_t i = 2; i < input[l].dim(); i++) {
      TORCH_CHECK(
          input[l].size(i) > 0, "grid_sampler(): expected input to have non-empty spatial dimensions, but input has sizes ",
          input[l].sizes(), " with dimension ", i, " being empty");
    }
  }
  TORCH_CHECK(max_aniso >= 1, "mipmap_aniso_grid_sampler_2d(): expected max_aniso >= 1");
  TORCH_CHECK(
      padding_mode >= 0 && padding_mode <= 2 && (interpolation_mode == 0 || interpolation_mode == 2),
      "mipmap_aniso_grid_sampler_2d(): unsupported padding_mode / interpolation_mode");
  const drtk_dtype_t dt = dtype_of(input[0], "mipmap_aniso_grid_sampler_2d_kernel");
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(input[0].device());
  const LevelArgs lv = prep_levels(input);
  const GridArg ga = prep_grid(grid);
  const auto vt_c = vt_dxdy_img.contiguous();
  const int64_t N = input[0].size(0), C = input[0].size(1), H = grid.size(1), W = grid.size(2);
  auto out = out_empty({N, C, H, W}, input[0].options());
  check_status(
      drtk_amd_mipmap_grid_sampler_2d(
          dt, lv.ptrs.data(), lv.h.data(), lv.w.data(), lv.sn.data(), static_cast<int>(mipmaps), ga.t.data_ptr(), ga.layout, vt_c.data_ptr(), N,
          C, H, W, static_cast<int>(std::min<int64_t>(max_aniso, 1 << 20)), static_cast<int>(padding_mode),
          static_cast<int>(interpolation_mode), align_corners, force_max_ansio, clip_grad, out.data_ptr(),
          current_stream(input[0])),
      "mipmap_aniso_grid_sampler_2d");
  return out;
}

std::tuple<std::vector<Tensor>, Tensor> mipmap_grid_sampler_2d_backward_hip(
    const Tensor& grad_output, const std::vector<Tensor>& input, const Tensor& grid, const Tensor& vt_dxdy_img,
    int64_t max_aniso, int64_t padding_mode, int64_t interpolation_mode, bool align_corners, bool force_max_ansio,
    bool clip_grad) {
  const drtk_dtype_t dt = dtype_of(input[0], "mipmap_aniso_grid_sampler_2d_backward_kernel");
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(input[0].device());
  const LevelArgs lv = prep_levels(input);
  const GridArg ga = prep_grid(grid);
  const auto vt_c = vt_dxdy_img.contiguous();
  const auto go_c = grad_output.to(input[0].scalar_type()).contiguous();
  std::vector<Tensor> grad_input;
  std::vector<void*> gptrs;
  // zero-filled by the call (:1120-1123); the levels are carved out of ONE buffer, back to back (16-byte multiples), so
  // that the call zeroes them with one launch instead of one per level
  {
    int64_t total = 0;
    std::vector<int64_t> offs;
    for (const Tensor& t : input) {
      offs.push_back(total);
      total += t.numel();
    }
    const Tensor flat = out_empty({total}, input[0].options());
    for (size_t l = 0; l < input.size(); ++l) {
      grad_input.push_back(flat.narrow(0, offs[l], input[l].numel()).view(input[l].sizes()));
      gptrs.push_back(grad_input.back().data_ptr());
    }
  }
  // laid out like the grid it belongs to: the gradient of a permuted channel-first uv image arrives channel-first at
  // interpolate's backward, which would otherwise copy it (the reference allocates it contiguous, :1126)
  auto grad_grid = at::empty_strided(ga.t.sizes(), ga.t.strides(), grid.options());
  const int64_t N = input[0].size(0), C = input[0].size(1), H = grid.size(1), W = grid.size(2);
  check_status(
      drtk_amd_mipmap_grid_sampler_2d_backward(
          dt, go_c.data_ptr(), lv.ptrs.data(), lv.h.data(), lv.w.data(), lv.sn.data(), static_cast<int>(input.size()),
          ga.t.data_ptr(), ga.layout, vt_c.data_ptr(), N, C, H, W, static_cast<int>(std::min<int64_t>(max_aniso, 1 << 20)),
          static_cast<int>(padding_mode), static_cast<int>(interpolation_mode), align_corners, force_max_ansio, clip_grad,
          gptrs.data(), grad_grid.data_ptr(), ga.layout, current_stream(input[0])),
      "mipmap_aniso_grid_sampler_2d_backward");
  return {grad_input, grad_grid};
}

Tensor mipmap_grid_sampler_2d_cpu(
    at::TensorList, const Tensor&, const Tensor&, int64_t, int64_t, int64_t, bool, bool, bool) {
  no_cpu("mipmap_aniso_grid_sampler_2d");
}

Tensor mipmap_grid_sampler_2d_op(
    at::TensorList input, const Tensor& grid, const Tensor& vt_dxdy_img, int64_t max_aniso, int64_t padding_mode,
    int64_t interpolation_mode, bool align_corners, bool force_max_ansio, bool clip_grad) {
  static auto op = c10::Dispatcher::singleton()
                       .findSchemaOrThrow("mipmap_grid_sampler_ext::mipmap_grid_sampler_2d", "")
                       .typed<decltype(mipmap_grid_sampler_2d_op)>();
  return op.call(input, grid, vt_dxdy_img, max_aniso, padding_mode, interpolation_mode, align_corners, force_max_ansio, clip_grad);
}

// A torch::autograd::Function cannot take a TensorList whose members need gradients, so -- like the
// reference (mipmap_grid_sampler_module.cpp:44-181) -- the pyramid is spread over 11 optional slots.
using OptTensor = std::optional<Tensor>;
class MipmapGridSample2DFunction : public torch::autograd::Function<MipmapGridSample2DFunction> {
 public:
  static tensor_list forward(
      AutogradContext* ctx, const Tensor& grid, const Tensor& vt_dxdy_img, int64_t max_aniso, int64_t padding_mode,
      int64_t interpolation_mode, bool align_corners, bool force_max_ansio, bool clip_grad, const Tensor& input0,
      const OptTensor& i1, const OptTensor& i2, const OptTensor& i3, const OptTensor& i4, const OptTensor& i5,
      const OptTensor& i6, const OptTensor& i7, const OptTensor& i8, const OptTensor& i9, const OptTensor& i10) {
    std::vector<Tensor> input = {input0};
    for (const OptTensor* o : {&i1, &i2, &i3, &i4, &i5, &i6, &i7, &i8, &i9, &i10}) {
      if (o->has_value()) input.push_back(o->value());
    }
    ctx->set_materialize_grads(false);
    std::vector<Tensor> save_list(input.begin(), input.end());
    save_list.push_back(grid);
    save_list.push_back(vt_dxdy_img);
    ctx->save_for_backward(save_list);
    bool requires_grad = grid.requires_grad(); // :95-99
    for (const auto& inp : input) requires_grad = requires_grad || inp.requires_grad();
    ctx->saved_data["data"] = std::make_tuple(
        static_cast<int64_t>(input.size()), requires_grad, max_aniso, padding_mode, interpolation_mode, align_corners,
        force_max_ansio, clip_grad);
    at::AutoDispatchBelowADInplaceOrView g;
    return {mipmap_grid_sampler_2d_op(
        input, grid, vt_dxdy_img, max_aniso, padding_mode, interpolation_mode, align_corners, force_max_ansio, clip_grad)};
  }
  static tensor_list backward(AutogradContext* ctx, tensor_list grad_outputs) {
    int64_t mipmaps, max_aniso, padding_mode, interpolation_mode;
    bool requires_grad, align_corners, force_max_ansio, clip_grad;
    std::tie(mipmaps, requires_grad, max_aniso, padding_mode, interpolation_mode, align_corners, force_max_ansio, clip_grad) =
        ctx->saved_data["data"].to<std::tuple<int64_t, bool, int64_t, int64_t, int64_t, bool, bool, bool>>();
    tensor_list grads(19);
    if (!requires_grad || !grad_outputs[0].defined()) return grads;
    const auto saved = ctx->get_saved_variables();
    const std::vector<Tensor> input(saved.begin(), saved.begin() + mipmaps);
    auto g = mipmap_grid_sampler_2d_backward_hip(
        grad_outputs[0], input, saved[mipmaps], saved[mipmaps + 1], max_aniso, padding_mode, interpolation_mode,
        align_corners, force_max_ansio, clip_grad);
    grads[0] = std::get<1>(g); // grid; slots 1..7 (vt_dxdy_img and the scalars) stay undefined
    for (int64_t i = 0; i < mipmaps; ++i) grads[8 + i] = std::get<0>(g)[i];
    return grads;
  }
};

Tensor mipmap_grid_sampler_2d_autograd(
    at::TensorList input, const Tensor& grid, const Tensor& vt_dxdy_img, int64_t max_aniso, int64_t padding_mode,
    int64_t interpolation_mode, bool align_corners, bool force_max_ansio, bool clip_grad) {
  TORCH_CHECK(input.size() >= 1, "mipmap_aniso_grid_sampler_2d(): expected input to have at least one mipmap level");
  TORCH_CHECK(input.size() <= 11, "mipmap_aniso_grid_sampler_2d(): at most 11 mipmap levels are supported");
  auto opt = [&](size_t i) { return input.size() > i ? OptTensor(input[i]) : OptTensor(); };
  return MipmapGridSample2DFunction::apply(
      grid, vt_dxdy_img, max_aniso, padding_mode, interpolation_mode, align_corners, force_max_ansio, clip_grad, input[0],
      opt(1), opt(2), opt(3), opt(4), opt(5), opt(6), opt(7), opt(8), opt(9), opt(10))[0];
}

Tensor mipmap_grid_sampler_2d_autocast(
    at::TensorList input, const Tensor& grid, const Tensor& vt_dxdy_img, int64_t max_aniso, int64_t padding_mode,
    int64_t interpolation_mode, bool align_corners, bool force_max_ansio, bool clip_grad) {
  c10::impl::ExcludeDispatchKeyGuard no_autocast(c10::DispatchKey::Autocast);
  return mipmap_grid_sampler_2d_op(
      at::autocast::cached_cast(at::kFloat, input), at::autocast::cached_cast(at::kFloat, grid),
      at::autocast::cached_cast(at::kFloat, vt_dxdy_img), max_aniso, padding_mode, interpolation_mode, align_corners,
      force_max_ansio, clip_grad);
}

// screen_space_uv_derivative: drtk/screen_space_uv_derivative.py:15-80 as one kernel (forward only)
Tensor screen_space_uv_derivative_hip(
    const Tensor& v, const Tensor& vt, const Tensor& vi, const Tensor& vti, const Tensor& index_img,
    const Tensor& bary_img, const Tensor& mask, const Tensor& campos, const Tensor& camrot, const Tensor& focal) {
  const char* op = "screen_space_uv_derivative";
  TORCH_CHECK(v.dim() == 3 && v.size(2) == 3, op, "(): expected v to be [N,V,3], got ", v.sizes()); // geometry.py:60-61
  TORCH_CHECK(vt.dim() == 3 && vt.size(2) == 2, op, "(): expected vt to be [N,T,2], got ", vt.sizes());
  TORCH_CHECK(vt.size(0) == v.size(0), op, "(): expected vt to have the same batch size as v, got ", vt.size(0), " and ", v.size(0));
  TORCH_CHECK(vi.dim() == 2 && vi.size(1) == 3 && vti.sizes() == vi.sizes(), op, "(): expected vi and vti to be [F,3]");
  TORCH_CHECK(vi.dtype() == at::kInt && vti.dtype() == at::kInt && index_img.dtype() == at::kInt, op, "(): expected int32 vi, vti and index_img");
  TORCH_CHECK(index_img.dim() == 3 && bary_img.dim() == 4 && bary_img.size(1) == 3 && bary_img.size(0) == index_img.size(0) &&
                  bary_img.size(2) == index_img.size(1) && bary_img.size(3) == index_img.size(2),
              op, "(): expected index_img [N,H,W] and bary_img [N,3,H,W]");
  const int64_t N = index_img.size(0), H = index_img.size(1), W = index_img.size(2);
  TORCH_CHECK(v.size(0) == N && campos.sizes() == at::IntArrayRef({N, 3}) && camrot.sizes() == at::IntArrayRef({N, 3, 3}) &&
                  focal.sizes() == at::IntArrayRef({N, 2, 2}),
              op, "(): expected v, campos [N,3], camrot [N,3,3], focal [N,2,2] to share the batch size of index_img");
  TORCH_CHECK(mask.sizes() == index_img.sizes() && (mask.dtype() == at::kBool || mask.dtype() == at::kByte), op, "(): expected a bool mask [N,H,W]");
  TORCH_CHECK(v.dtype() == vt.dtype() && v.dtype() == bary_img.dtype() && v.dtype() == campos.dtype() && v.dtype() == camrot.dtype() &&
                  v.dtype() == focal.dtype(), op, "(): expected v, vt, bary_img and the camera tensors to share one dtype");
  TORCH_CHECK(v.is_cuda(), op, "(): drtk_amd implements the MI355X (HIP) path only; got CPU tensors");
  for (const Tensor* t : {&vt, &vi, &vti, &index_img, &bary_img, &mask, &campos, &camrot, &focal})
    TORCH_CHECK(t->device() == v.device(), op, "(): expected all inputs to be on same device");
  const drtk_dtype_t dt = dtype_of(v, op);
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(v.device());
  // view-shared geometry (stride-0 batch) is passed once
  const bool v_shared = N > 1 && v.stride(0) == 0, vt_shared = N > 1 && vt.stride(0) == 0;
  const auto v_c = (v_shared ? v.select(0, 0) : v).contiguous();
  const auto vt_c = (vt_shared ? vt.select(0, 0) : vt).contiguous();
  const auto vi_c = vi.contiguous(), vti_c = vti.contiguous(), idx_c = index_img.contiguous(), bary_c = bary_img.contiguous();
  const auto mask_c = mask.to(at::kByte).contiguous();
  const auto cp_c = campos.contiguous(), cr_c = camrot.contiguous(), f_c = focal.contiguous();
  auto out = out_empty({N, H, W, 2, 2}, bary_img.options());
  check_status(
      drtk_amd_screen_space_uv_derivative(
          dt, v_c.data_ptr(), v_shared ? 0 : v.size(1) * 3, vt_c.data_ptr(), vt_shared ? 0 : vt.size(1) * 2,
          vi_c.data_ptr<int32_t>(), vti_c.data_ptr<int32_t>(), idx_c.data_ptr<int32_t>(), bary_c.data_ptr(),
          mask_c.data_ptr<uint8_t>(), cp_c.data_ptr(), cr_c.data_ptr(), f_c.data_ptr(), N, v.size(1), vt.size(1), vi.size(0), H,
          W, out.data_ptr(), current_stream(v)),
      op);
  return out;
}

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

// ---------------------------------------------------------------------------------------------
// transform_pinhole -- drtk_amd extension backing the pinhole fast path of drtk_amd.transform
// (reference: pure PyTorch, drtk/transform.py:13-119).  Differentiable with respect to v only.
// ---------------------------------------------------------------------------------------------
struct TransformArgs {
  Tensor v, campos, camrot, focal, princpt;
  int64_t N, V, v_sN;
};
TransformArgs transform_prep(
    const Tensor& v, const Tensor& campos, const Tensor& camrot, const Tensor& focal, const Tensor& princpt) {
  TORCH_CHECK(v.is_cuda(), "transform(): drtk_amd implements the MI355X (HIP) path only; got CPU tensors");
  TORCH_CHECK(v.dim() == 3 && v.size(2) == 3, "transform(): expected v of shape [N, V, 3] or [1, V, 3]");
  const int64_t N = campos.size(0);
  TORCH_CHECK(campos.dim() == 2 && campos.size(1) == 3, "transform(): expected campos of shape [N, 3]");
  TORCH_CHECK(camrot.dim() == 3 && camrot.size(0) == N && camrot.size(1) == 3 && camrot.size(2) == 3,
              "transform(): expected camrot of shape [N, 3, 3]");
  TORCH_CHECK(focal.dim() == 3 && focal.size(0) == N && focal.size(1) == 2 && focal.size(2) == 2,
              "transform(): expected focal of shape [N, 2, 2]");
  TORCH_CHECK(princpt.dim() == 2 && princpt.size(0) == N && princpt.size(1) == 2,
              "transform(): expected princpt of shape [N, 2]");
  TORCH_CHECK(v.size(0) == N || v.size(0) == 1, "transform(): batch size of v must be 1 or match the cameras");
  TransformArgs a;
  const auto dt = v.scalar_type();
  a.v = v.contiguous();
  a.campos = campos.to(dt).contiguous();
  a.camrot = camrot.to(dt).contiguous();
  a.focal = focal.to(dt).contiguous();
  a.princpt = princpt.to(dt).contiguous();
  a.N = N;
  a.V = v.size(1);
  a.v_sN = (v.size(0) == 1 && N != 1) ? 0 : v.size(1) * 3;
  return a;
}

Tensor transform_pinhole_hip(
    const Tensor& v, const Tensor& campos, const Tensor& camrot, const Tensor& focal, const Tensor& princpt) {
  const drtk_dtype_t dt = dtype_of(v, "transform");
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(v.device());
  const TransformArgs a = transform_prep(v, campos, camrot, focal, princpt);
  auto v_pix = out_empty({a.N, a.V, 3}, v.options());
  check_status(
      drtk_amd_transform_pinhole(
          dt, a.v.data_ptr(), a.v_sN, a.campos.data_ptr(), a.camrot.data_ptr(), a.focal.data_ptr(),
          a.princpt.data_ptr(), a.N, a.V, v_pix.data_ptr(), nullptr, current_stream(v)),
      "transform");
  return v_pix;
}

Tensor transform_pinhole_backward_hip(
    const Tensor& v, const Tensor& campos, const Tensor& camrot, const Tensor& focal, const Tensor& princpt,
    const Tensor& grad_v_pix) {
  const drtk_dtype_t dt = dtype_of(v, "transform_backward");
  c10::hip::OptionalHIPGuardMasqueradingAsCUDA guard(v.device());
  const TransformArgs a = transform_prep(v, campos, camrot, focal, princpt);
  const auto g = grad_v_pix.to(v.scalar_type()).contiguous();
  auto grad_v = at::empty_like(a.v); // [1,V,3] (summed over views) or [N,V,3]
  check_status(
      drtk_amd_transform_pinhole_backward(
          dt, a.v.data_ptr(), a.v_sN, a.campos.data_ptr(), a.camrot.data_ptr(), a.focal.data_ptr(),
          a.princpt.data_ptr(), g.data_ptr(), a.N, a.V, grad_v.data_ptr(), current_stream(v)),
      "transform_backward");
  return grad_v;
}

Tensor transform_pinhole_cpu(const Tensor&, const Tensor&, const Tensor&, const Tensor&, const Tensor&) {
  no_cpu("transform");
}

Tensor transform_pinhole_op(
    const Tensor& v, const Tensor& campos, const Tensor& camrot, const Tensor& focal, const Tensor& princpt) {
  static auto op = c10::Dispatcher::singleton()
                       .findSchemaOrThrow("drtk_amd_ext::transform_pinhole", "")
                       .typed<decltype(transform_pinhole_op)>();
  return op.call(v, campos, camrot, focal, princpt);
}

class TransformPinholeFunction : public torch::autograd::Function<TransformPinholeFunction> {
 public:
  static tensor_list forward(
      AutogradContext* ctx, const Tensor& v, const Tensor& campos, const Tensor& camrot, const Tensor& focal,
      const Tensor& princpt) {
    ctx->set_materialize_grads(false);
    ctx->save_for_backward({v, campos, camrot, focal, princpt});
    at::AutoDispatchBelowADInplaceOrView g;
    return {transform_pinhole_op(v, campos, camrot, focal, princpt)};
  }
  static tensor_list backward(AutogradContext* ctx, tensor_list grad_outputs) {
    const auto saved = ctx->get_saved_variables();
    if (!saved[0].requires_grad() || !grad_outputs[0].defined()) return {Tensor(), Tensor(), Tensor(), Tensor(), Tensor()};
    auto gv = transform_pinhole_backward_hip(saved[0], saved[1], saved[2], saved[3], saved[4], grad_outputs[0]);
    return {gv, Tensor(), Tensor(), Tensor(), Tensor()};
  }
};

Tensor transform_pinhole_autograd(
    const Tensor& v, const Tensor& campos, const Tensor& camrot, const Tensor& focal, const Tensor& princpt) {
  return TransformPinholeFunction::apply(v, campos, camrot, focal, princpt)[0];
}

} // namespace

// ---- schemas: verbatim from the reference ------------------------------------------------------
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

TORCH_LIBRARY(interpolate_ext, m) {
  m.def("interpolate(Tensor vert_attributes, Tensor vi, Tensor index_img, Tensor bary_img) -> Tensor");
  m.def("interpolation_matrix(Tensor vi, Tensor index_img, Tensor bary_img) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def(
      "interpolation_normal_matrix(Tensor vi, Tensor index_img, Tensor bary_img, int num_vertices) -> (Tensor, Tensor, Tensor)");
  m.def(
      "interpolation_normal_matrix_values(Tensor pair_indices, Tensor index_img, Tensor bary_img, int nnz) -> Tensor");
}
TORCH_LIBRARY_IMPL(interpolate_ext, Autograd, m) {
  m.impl("interpolate", &interpolate_autograd);
  m.impl("interpolation_matrix", &interpolation_matrix_autograd);
  m.impl("interpolation_normal_matrix", &interpolation_normal_matrix_autograd);
  m.impl("interpolation_normal_matrix_values", &normal_matrix_values_autograd);
}
TORCH_LIBRARY_IMPL(interpolate_ext, Autocast, m) {
  m.impl("interpolate", interpolate_autocast);
  m.impl("interpolation_matrix", interpolation_matrix_autocast);
  m.impl("interpolation_normal_matrix", interpolation_normal_matrix_autocast);
  m.impl("interpolation_normal_matrix_values", normal_matrix_values_autocast);
}
TORCH_LIBRARY_IMPL(interpolate_ext, CUDA, m) {
  m.impl("interpolate", &interpolate_hip);
  m.impl("interpolation_matrix", &interpolation_matrix_hip);
  m.impl("interpolation_normal_matrix", &interpolation_normal_matrix_hip);
  m.impl("interpolation_normal_matrix_values", &normal_matrix_values_hip);
}
TORCH_LIBRARY_IMPL(interpolate_ext, CPU, m) {
  m.impl("interpolate", &interpolate_cpu);
  m.impl("interpolation_matrix", &interpolation_matrix_cpu);
  m.impl("interpolation_normal_matrix", &interpolation_normal_matrix_cpu);
  m.impl("interpolation_normal_matrix_values", &normal_matrix_values_cpu);
}

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

TORCH_LIBRARY(mipmap_grid_sampler_ext, m) {
  m.def(
      "mipmap_grid_sampler_2d(Tensor[] x, Tensor grid, Tensor vt_dxdy_img, int max_aniso, int padding_mode, int interpolation_mode, bool align_corners, bool force_max_ansio, bool clip_grad) -> Tensor");
}
TORCH_LIBRARY_IMPL(mipmap_grid_sampler_ext, Autograd, m) {
  m.impl("mipmap_grid_sampler_2d", &mipmap_grid_sampler_2d_autograd);
}
TORCH_LIBRARY_IMPL(mipmap_grid_sampler_ext, Autocast, m) {
  m.impl("mipmap_grid_sampler_2d", mipmap_grid_sampler_2d_autocast);
}
TORCH_LIBRARY_IMPL(mipmap_grid_sampler_ext, CUDA, m) {
  m.impl("mipmap_grid_sampler_2d", &mipmap_grid_sampler_2d_hip);
}
TORCH_LIBRARY_IMPL(mipmap_grid_sampler_ext, CPU, m) { // the reference registers no CPU kernel either
  m.impl("mipmap_grid_sampler_2d", &mipmap_grid_sampler_2d_cpu);
}

// drtk_amd's own namespace (extensions with no reference counterpart)
TORCH_LIBRARY(drtk_amd_ext, m) {
  m.def("transform_pinhole(Tensor v, Tensor campos, Tensor camrot, Tensor focal, Tensor princpt) -> Tensor");
  m.def("interpolate_masked(Tensor vert_attributes, Tensor vi, Tensor index_img, Tensor bary_img) -> Tensor");
  m.def(
      "screen_space_uv_derivative(Tensor v, Tensor vt, Tensor vi, Tensor vti, Tensor index_img, Tensor bary_img, Tensor mask, Tensor campos, Tensor camrot, Tensor focal) -> Tensor",
      &screen_space_uv_derivative_hip);
  // topology-only: the cached A^T A pattern (crow_indices, col_indices, pair_indices) on vi's device
  m.def("normal_matrix_structure(Tensor vi, int num_vertices) -> (Tensor, Tensor, Tensor)", &normal_matrix_structure_any);
  m.def("normal_matrix_cache_stats() -> int[]", &normal_matrix_cache_stats);
  m.def("normal_matrix_cache_clear() -> ()", &normal_matrix_cache_clear);
}
TORCH_LIBRARY_IMPL(drtk_amd_ext, Autograd, m) {
  m.impl("transform_pinhole", &transform_pinhole_autograd);
  m.impl("interpolate_masked", &interpolate_masked_autograd);
}
TORCH_LIBRARY_IMPL(drtk_amd_ext, CUDA, m) {
  m.impl("transform_pinhole", &transform_pinhole_hip);
  m.impl("interpolate_masked", &interpolate_masked_hip);
}
TORCH_LIBRARY_IMPL(drtk_amd_ext, CPU, m) {
  m.impl("transform_pinhole", &transform_pinhole_cpu);
  m.impl("interpolate_masked", &interpolate_masked_cpu);
}

// drtk_amd_ext::transform_pinhole -- the pinhole case of drtk/transform.py + drtk/utils/projection.py in one kernel each
// way, over drtk_amd_transform_pinhole[_backward].
#include "common.hpp"

namespace {
using namespace drtk_amd_torch;

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

TORCH_LIBRARY_FRAGMENT(drtk_amd_ext, m) {
  m.def("transform_pinhole(Tensor v, Tensor campos, Tensor camrot, Tensor focal, Tensor princpt) -> Tensor");
}
TORCH_LIBRARY_IMPL(drtk_amd_ext, Autograd, m) {
  m.impl("transform_pinhole", &transform_pinhole_autograd);
}
TORCH_LIBRARY_IMPL(drtk_amd_ext, CUDA, m) {
  m.impl("transform_pinhole", &transform_pinhole_hip);
}
TORCH_LIBRARY_IMPL(drtk_amd_ext, CPU, m) {
  m.impl("transform_pinhole", &transform_pinhole_cpu);
}

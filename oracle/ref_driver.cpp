// TEST INFRASTRUCTURE ONLY -- not part of the product path.
//
// Thin torch-op driver around the reference's own CPU kernels
// (/root/reference/src/{rasterize,render,interpolate,edge_grad}/*_kernel_cpu.cpp), which are
// compiled *from where they lie* by oracle/ref_build.py into oracle/_ref/.  Nothing from the
// reference is copied here: this file only declares the entry points (via the reference's own
// *_kernel.h headers on the include path) and forwards to them so that Python can call the
// forward AND backward CPU kernels directly, without the reference's autograd modules (those
// need the CUDA entry points to link).
//
// Namespace is selected at build time (-DREF_NS=drtk_ref_strict / drtk_ref_fast) so a strict-IEEE
// build and a build with the reference's own "-O3 -ffast-math" flags can be loaded side by side.
#include <torch/library.h>
#include <torch/types.h>

#include "edge_grad_kernel.h"
#include "interpolate_kernel.h"
#include "rasterize_kernel.h"
#include "render_kernel.h"

#ifndef REF_NS
#define REF_NS drtk_ref_strict
#endif

namespace {

std::vector<torch::Tensor>
ref_rasterize(const torch::Tensor& v, const torch::Tensor& vi, int64_t h, int64_t w) {
  return rasterize_cpu(v, vi, h, w, /*wireframe=*/false);
}

std::vector<torch::Tensor>
ref_render(const torch::Tensor& v, const torch::Tensor& vi, const torch::Tensor& index_img) {
  return render_cpu(v, vi, index_img);
}

torch::Tensor ref_render_backward(
    const torch::Tensor& v,
    const torch::Tensor& vi,
    const torch::Tensor& index_img,
    const torch::Tensor& grad_depth_img,
    const torch::Tensor& grad_bary_img) {
  return render_cpu_backward(v, vi, index_img, grad_depth_img, grad_bary_img);
}

torch::Tensor ref_interpolate(
    const torch::Tensor& vert_attributes,
    const torch::Tensor& vi,
    const torch::Tensor& index_img,
    const torch::Tensor& bary_img) {
  return interpolate_cpu(vert_attributes, vi, index_img, bary_img);
}

// The reference decides which gradients to produce from requires_grad() of the saved tensors
// (interpolate_kernel_cpu.cpp:355-356), so the flags are re-created on detached aliases here.
std::vector<torch::Tensor> ref_interpolate_backward(
    const torch::Tensor& grad_out,
    const torch::Tensor& vert_attributes,
    const torch::Tensor& vi,
    const torch::Tensor& index_img,
    const torch::Tensor& bary_img,
    bool vert_requires_grad,
    bool bary_requires_grad) {
  auto a = vert_attributes.detach();
  a.set_requires_grad(vert_requires_grad);
  auto b = bary_img.detach();
  b.set_requires_grad(bary_requires_grad);
  auto r = interpolate_cpu_backward(grad_out, a, vi, index_img, b);
  auto vg = std::get<0>(r);
  auto bg = std::get<1>(r);
  return {
      vg.defined() ? vg : torch::empty({0}, vert_attributes.options()),
      bg.defined() ? bg : torch::empty({0}, bary_img.options())};
}

torch::Tensor ref_edge_grad_backward(
    const torch::Tensor& v_pix,
    const torch::Tensor& img,
    const torch::Tensor& index_img,
    const torch::Tensor& vi,
    const torch::Tensor& grad_outputs,
    double max_dp_dr) {
  return edge_grad_estimator_cpu_backward(v_pix, img, index_img, vi, grad_outputs, max_dp_dr);
}

torch::Tensor ref_edge_grad_fwd_check(
    const torch::Tensor& v_pix,
    const torch::Tensor& v_pix_img,
    const torch::Tensor& vi,
    const torch::Tensor& img,
    const torch::Tensor& index_img,
    double max_dp_dr) {
  return edge_grad_estimator_cpu_fwd(v_pix, v_pix_img, vi, img, index_img, max_dp_dr);
}

std::vector<torch::Tensor> ref_interpolation_matrix(
    const torch::Tensor& vi, const torch::Tensor& index_img, const torch::Tensor& bary_img) {
  auto r = interpolation_matrix_cpu(vi, index_img, bary_img);
  return {std::get<0>(r), std::get<1>(r), std::get<2>(r), std::get<3>(r)};
}

torch::Tensor ref_interpolation_matrix_backward(
    const torch::Tensor& grad_values, const torch::Tensor& vi, const torch::Tensor& index_img,
    const torch::Tensor& bary_img, const torch::Tensor& row_pixels) {
  return interpolation_matrix_cpu_backward(grad_values, vi, index_img, bary_img, row_pixels);
}

torch::Tensor ref_normal_matrix_values(
    const torch::Tensor& pair_indices, const torch::Tensor& index_img, const torch::Tensor& bary_img,
    int64_t nnz) {
  return interpolation_normal_matrix_values_cpu(pair_indices, index_img, bary_img, nnz);
}

torch::Tensor ref_normal_matrix_values_backward(
    const torch::Tensor& grad_values, const torch::Tensor& pair_indices, const torch::Tensor& index_img,
    const torch::Tensor& bary_img) {
  return interpolation_normal_matrix_values_cpu_backward(grad_values, pair_indices, index_img, bary_img);
}

} // namespace

// TORCH_LIBRARY stringifies its first argument, so expand REF_NS through one more macro level.
#define DRTK_REF_LIBRARY(ns, m) TORCH_LIBRARY(ns, m)
DRTK_REF_LIBRARY(REF_NS, m) {
  m.def("rasterize(Tensor v, Tensor vi, int height, int width) -> Tensor[]", &ref_rasterize);
  m.def("render(Tensor v, Tensor vi, Tensor index_img) -> Tensor[]", &ref_render);
  m.def(
      "render_backward(Tensor v, Tensor vi, Tensor index_img, Tensor grad_depth_img, Tensor grad_bary_img) -> Tensor",
      &ref_render_backward);
  m.def(
      "interpolate(Tensor vert_attributes, Tensor vi, Tensor index_img, Tensor bary_img) -> Tensor",
      &ref_interpolate);
  m.def(
      "interpolate_backward(Tensor grad_out, Tensor vert_attributes, Tensor vi, Tensor index_img, Tensor bary_img, bool vert_requires_grad, bool bary_requires_grad) -> Tensor[]",
      &ref_interpolate_backward);
  m.def(
      "edge_grad_backward(Tensor v_pix, Tensor img, Tensor index_img, Tensor vi, Tensor grad_outputs, float max_dp_dr) -> Tensor",
      &ref_edge_grad_backward);
  m.def(
      "interpolation_matrix(Tensor vi, Tensor index_img, Tensor bary_img) -> Tensor[]",
      &ref_interpolation_matrix);
  m.def(
      "interpolation_matrix_backward(Tensor grad_values, Tensor vi, Tensor index_img, Tensor bary_img, Tensor row_pixels) -> Tensor",
      &ref_interpolation_matrix_backward);
  m.def(
      "normal_matrix_values(Tensor pair_indices, Tensor index_img, Tensor bary_img, int nnz) -> Tensor",
      &ref_normal_matrix_values);
  m.def(
      "normal_matrix_values_backward(Tensor grad_values, Tensor pair_indices, Tensor index_img, Tensor bary_img) -> Tensor",
      &ref_normal_matrix_values_backward);
  m.def(
      "edge_grad_fwd_check(Tensor v_pix, Tensor v_pix_img, Tensor vi, Tensor img, Tensor index_img, float max_dp_dr) -> Tensor",
      &ref_edge_grad_fwd_check);
}

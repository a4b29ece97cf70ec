// mipmap_grid_sampler_ext::mipmap_grid_sampler_2d -- src/mipmap_grid_sampler/mipmap_grid_sampler_module.cpp:16-266 --
// and drtk_amd_ext::screen_space_uv_derivative (forward of drtk/screen_space_uv_derivative.py:15-80 in one kernel).
#include "common.hpp"

namespace {
using namespace drtk_amd_torch;

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

} // namespace

// schema: verbatim from the reference
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

TORCH_LIBRARY_FRAGMENT(drtk_amd_ext, m) {
  m.def(
      "screen_space_uv_derivative(Tensor v, Tensor vt, Tensor vi, Tensor vti, Tensor index_img, Tensor bary_img, Tensor mask, Tensor campos, Tensor camrot, Tensor focal) -> Tensor",
      &screen_space_uv_derivative_hip);
}

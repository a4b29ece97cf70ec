// interpolate_ext::interpolation_matrix / interpolation_normal_matrix[_values] -- src/interpolate/interpolate_module.cpp
// :28-310,435-669 -- with the topology-only A^T A pattern built once per face-index tensor and cached.
#include "common.hpp"

namespace {
using namespace drtk_amd_torch;

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

} // namespace

// schemas: verbatim from the reference
TORCH_LIBRARY_FRAGMENT(interpolate_ext, m) {
  m.def("interpolation_matrix(Tensor vi, Tensor index_img, Tensor bary_img) -> (Tensor, Tensor, Tensor, Tensor)");
  m.def(
      "interpolation_normal_matrix(Tensor vi, Tensor index_img, Tensor bary_img, int num_vertices) -> (Tensor, Tensor, Tensor)");
  m.def(
      "interpolation_normal_matrix_values(Tensor pair_indices, Tensor index_img, Tensor bary_img, int nnz) -> Tensor");
}
TORCH_LIBRARY_IMPL(interpolate_ext, Autograd, m) {
  m.impl("interpolation_matrix", &interpolation_matrix_autograd);
  m.impl("interpolation_normal_matrix", &interpolation_normal_matrix_autograd);
  m.impl("interpolation_normal_matrix_values", &normal_matrix_values_autograd);
}
TORCH_LIBRARY_IMPL(interpolate_ext, Autocast, m) {
  m.impl("interpolation_matrix", interpolation_matrix_autocast);
  m.impl("interpolation_normal_matrix", interpolation_normal_matrix_autocast);
  m.impl("interpolation_normal_matrix_values", normal_matrix_values_autocast);
}
TORCH_LIBRARY_IMPL(interpolate_ext, CUDA, m) {
  m.impl("interpolation_matrix", &interpolation_matrix_hip);
  m.impl("interpolation_normal_matrix", &interpolation_normal_matrix_hip);
  m.impl("interpolation_normal_matrix_values", &normal_matrix_values_hip);
}
TORCH_LIBRARY_IMPL(interpolate_ext, CPU, m) {
  m.impl("interpolation_matrix", &interpolation_matrix_cpu);
  m.impl("interpolation_normal_matrix", &interpolation_normal_matrix_cpu);
  m.impl("interpolation_normal_matrix_values", &normal_matrix_values_cpu);
}

// drtk_amd's own namespace: the cached A^T A pattern (crow_indices, col_indices, pair_indices) on vi's device
TORCH_LIBRARY_FRAGMENT(drtk_amd_ext, m) {
  m.def("normal_matrix_structure(Tensor vi, int num_vertices) -> (Tensor, Tensor, Tensor)", &normal_matrix_structure_any);
  m.def("normal_matrix_cache_stats() -> int[]", &normal_matrix_cache_stats);
  m.def("normal_matrix_cache_clear() -> ()", &normal_matrix_cache_clear);
}

// Sparse interpolation operators: the remaining ops of interpolate_ext.
//
//   interpolation_matrix                A  [rows = foreground pixels, cols = vertices], 3 nnz/row
//       reference: interpolation_matrix_kernel        src/interpolate/interpolate_kernel.cu:301-339
//                  (CPU twin interpolate_kernel_cpu.cpp:411-497)
//   interpolation_matrix_backward       d values -> d bary                 :341-369 / cpu :499-546
//   interpolation_normal_matrix_values  values of A^T A on a cached CSR pattern: 9 products
//       bary_i * bary_j per foreground pixel added at pair_indices[n, tri, 3i+j]   :371-410 / cpu :548-626
//   ..._values_backward                 product rule                        :412-452 / cpu :628-693
//
// The pattern of A^T A (crow/col/pair_indices) depends on topology only and is built on the host by
// the torch shim (cached, like the reference's interpolate_module.cpp:62-261).
#include "common.hpp"
#include "segscatter.hpp"

namespace drtk_amd {
namespace {

// interpolate_kernel_cpu.cpp:17-30 : order of the three corners by ascending column
__device__ __forceinline__ void sorted_corner_order(const int32_t cols[3], int order[3]) {
  order[0] = 0, order[1] = 1, order[2] = 2;
  auto swap_if_less = [&](int a, int b) {
    if (cols[order[b]] < cols[order[a]]) {
      const int t = order[a];
      order[a] = order[b];
      order[b] = t;
    }
  };
  swap_if_less(0, 1);
  swap_if_less(1, 2);
  swap_if_less(0, 1);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void interpolation_matrix_kernel(
    const int32_t* __restrict__ vi, const int32_t* __restrict__ index_img, const T* __restrict__ bary_img,
    const int64_t* __restrict__ row_pixels, int64_t R, int64_t vi_sN, int64_t HW,
    int64_t* __restrict__ col_indices, T* __restrict__ values) {
  const int64_t row = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (row >= R) return;
  const int64_t flat = row_pixels[row];
  const int32_t tri = index_img[flat];
  const int64_t n = flat / HW, p = flat - n * HW;
  const int32_t* face = vi + n * vi_sN + int64_t(tri) * 3;
  const int32_t cols[3] = {face[0], face[1], face[2]};
  const T* bp = bary_img + n * 3 * HW + p;
  const T b[3] = {bp[0], bp[HW], bp[2 * HW]};
  int order[3];
  sorted_corner_order(cols, order);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    // order[] is a tiny dynamic index: select instead of indexing registers
    const int c = order[k];
    col_indices[row * 3 + k] = c == 0 ? cols[0] : (c == 1 ? cols[1] : cols[2]);
    values[row * 3 + k] = c == 0 ? b[0] : (c == 1 ? b[1] : b[2]);
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void interpolation_matrix_backward_kernel(
    const T* __restrict__ grad_values, const int32_t* __restrict__ vi,
    const int32_t* __restrict__ index_img, const int64_t* __restrict__ row_pixels, int64_t R,
    int64_t vi_sN, int64_t HW, T* __restrict__ bary_grad) {
  const int64_t row = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (row >= R) return;
  const int64_t flat = row_pixels[row];
  const int32_t tri = index_img[flat];
  const int64_t n = flat / HW, p = flat - n * HW;
  const int32_t* face = vi + n * vi_sN + int64_t(tri) * 3;
  const int32_t cols[3] = {face[0], face[1], face[2]};
  int order[3];
  sorted_corner_order(cols, order);
  T* gp = bary_grad + n * 3 * HW + p;
#pragma unroll
  for (int k = 0; k < 3; ++k) gp[int64_t(order[k]) * HW] = grad_values[row * 3 + k];
}

// values[pair[n,tri,3i+j]] += b_i * b_j.  Round 6 (rounds 2-5: the products staged in LDS, summed per run of equal triangle
// along a row and flushed with nine single-float atomics PER RUN -- 24 M scattered atomic requests on the bench views, which
// is what the kernel cost: 1.10 ms, 0.07 of the HBM peak on its 19 B/px): render backward's scheme.  A workgroup owns a
// 64 x 16 pixel tile, each wave 4 adjacent rows; lane = pixel; the SIX distinct products of a pixel (b_i b_j == b_j b_i
// exactly) stay in registers, a segmented scan over the 16-lane rows leaves each run's sums in its last lane, which adds
// them to a wave-private table keyed by TRIANGLE (double accumulators); the table is flushed once per 64 x 4 pixels with
// the nine atomics of each triangle it holds -- a triangle costs nine requests per wave tile instead of nine per run.
template <typename T>
__global__ __launch_bounds__(kBlock, sizeof(T) == 4 ? 8 : 4) void normal_matrix_values_kernel(
    const int32_t* __restrict__ pair_indices, const int32_t* __restrict__ index_img,
    const T* __restrict__ bary_img, int64_t pair_sN, int H, int W, int tiles_x, T* __restrict__ values, int strip) {
  constexpr int kWaves = kBlock / kWave;
  __shared__ int32_t t_keys[kWaves][kTableSlots];
  __shared__ TableAcc t_vals[kWaves][kTableSlots * 6];
  const int64_t HW = int64_t(H) * W;
  const int n = blockIdx.y;
  const int tile = tile_index(strip);
  const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
  const int lane = threadIdx.x & (kWave - 1);
  const int x = txi * kWave + lane;
  const int32_t* pr_n = pair_indices + int64_t(n) * pair_sN;
  table_init(t_keys[wave]);
  for (int i = lane; i < kTableSlots * 6; i += kWave) t_vals[wave][i] = 0;
  wave_lds_sync();
  auto load_tr = [&](int pass) -> int32_t {
    const int yy = tyi * kTileRows + wave * (kTileRows / kWaves) + pass;
    return (x < W && yy < H) ? index_img[int64_t(n) * HW + int64_t(yy) * W + x] : -1;
  };
  int32_t tr_next = load_tr(0);
#pragma unroll 1
  for (int pass = 0; pass < kTileRows / kWaves; ++pass) {
    const int y = tyi * kTileRows + wave * (kTileRows / kWaves) + pass;
    const int32_t tr = tr_next;
    if (pass + 1 < kTileRows / kWaves) tr_next = load_tr(pass + 1);
    if (__ballot(tr != -1) == 0) continue;
    T g[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) g[j] = T(0);
    if (tr != -1) {
      const T* bp = bary_img + int64_t(n) * 3 * HW + int64_t(y) * W + x;
      const T b0 = bp[0], b1 = bp[HW], b2 = bp[2 * HW];
      g[0] = b0 * b0, g[1] = b0 * b1, g[2] = b0 * b2, g[3] = b1 * b1, g[4] = b1 * b2, g[5] = b2 * b2;
    }
    int dist;
    bool tail;
    run_rows16(tr, dist, tail);
    run_sums_rows16<T, 6>(g, dist);
    if (tail && tr != -1) {
      const int slot = table_slot(t_keys[wave], tr);
      if (slot >= 0) {
#pragma unroll
        for (int j = 0; j < 6; ++j) lds_atomic_add(t_vals[wave] + slot * 6 + j, static_cast<TableAcc>(g[j]));
      } else { // the table has no room for this triangle: its nine entries directly
        const int32_t* pr = pr_n + int64_t(tr) * 9;
        constexpr int kSym[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
#pragma unroll
        for (int j = 0; j < 9; ++j) atomic_add_global(values + pr[j], g[kSym[j]]);
      }
    }
  }
  wave_lds_sync();
  for (int e = lane; e < kTableSlots * 9; e += kWave) {
    const int s = e / 9, j = e - s * 9;
    const int32_t key = t_keys[wave][s];
    if (key >= 0) {
      const int i3 = j / 3, j3 = j - i3 * 3;
      const int lo = i3 < j3 ? i3 : j3, hi = i3 < j3 ? j3 : i3;
      const int sym = lo == 0 ? hi : (lo == 1 ? 2 + hi : 5); // (0,0) (0,1) (0,2) (1,1) (1,2) (2,2) -> 0 .. 5
      const T xv = static_cast<T>(t_vals[wave][s * 6 + sym]);
      if (xv != T(0)) atomic_add_global(values + pr_n[int64_t(key) * 9 + j], xv);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void normal_matrix_values_backward_kernel(
    const T* __restrict__ grad_values, const int32_t* __restrict__ pair_indices,
    const int32_t* __restrict__ index_img, const T* __restrict__ bary_img, int64_t pair_sN, int64_t HW,
    T* __restrict__ bary_grad) {
  const int n = blockIdx.y;
  const int64_t pix = int64_t(blockIdx.x) * kBlock + threadIdx.x;
  if (pix >= HW) return;
  const int32_t tr = index_img[int64_t(n) * HW + pix];
  T o0 = T(0), o1 = T(0), o2 = T(0);
  if (tr != -1) {
    const int32_t* pr = pair_indices + int64_t(n) * pair_sN + int64_t(tr) * 9;
    const T* bp = bary_img + int64_t(n) * 3 * HW + pix;
    const T b0 = bp[0], b1 = bp[HW], b2 = bp[2 * HW];
    const T g00 = grad_values[pr[0]], g01 = grad_values[pr[1]], g02 = grad_values[pr[2]];
    const T g10 = grad_values[pr[3]], g11 = grad_values[pr[4]], g12 = grad_values[pr[5]];
    const T g20 = grad_values[pr[6]], g21 = grad_values[pr[7]], g22 = grad_values[pr[8]];
    o0 = T(2) * g00 * b0 + (g01 + g10) * b1 + (g02 + g20) * b2;
    o1 = (g10 + g01) * b0 + T(2) * g11 * b1 + (g12 + g21) * b2;
    o2 = (g20 + g02) * b0 + (g21 + g12) * b1 + T(2) * g22 * b2;
  }
  T* gp = bary_grad + int64_t(n) * 3 * HW + pix;
  gp[0] = o0, gp[HW] = o1, gp[2 * HW] = o2;
}

bool bad(int64_t N, int64_t F, int64_t H, int64_t W) {
  return N < 0 || F < 0 || H < 0 || W < 0 || H * W >= (int64_t(1) << 31);
}

} // namespace
} // namespace drtk_amd

using namespace drtk_amd;

#define DRTK_DISPATCH(dtype, CALL_F32, CALL_F64) \
  switch (dtype) {                               \
    case DRTK_F32: {                             \
      using T = float;                           \
      CALL_F32;                                  \
      break;                                     \
    }                                            \
    case DRTK_F64: {                             \
      using T = double;                          \
      CALL_F64;                                  \
      break;                                     \
    }                                            \
    default:                                     \
      return DRTK_ERR_INVALID_ARGUMENT;          \
  }

extern "C" int drtk_amd_interpolation_matrix(
    drtk_dtype_t dtype, const int32_t* vi, const int32_t* index_img, const void* bary_img,
    const int64_t* row_pixels, int64_t R, int64_t N, int64_t F, int64_t vi_sN, int64_t H, int64_t W,
    int64_t* col_indices, void* values, drtk_stream_t stream) {
  if (bad(N, F, H, W) || R < 0 || (vi_sN != 0 && vi_sN != F * 3)) return DRTK_ERR_INVALID_ARGUMENT;
  if (R == 0) return DRTK_OK;
  if (!vi || !index_img || !bary_img || !row_pixels || !col_indices || !values) return DRTK_ERR_INVALID_ARGUMENT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(ceil_div(R, kBlock)));
#define CALL DRTK_LAUNCH((interpolation_matrix_kernel<T>), grid, dim3(kBlock), 0, s, vi, index_img, static_cast<const T*>(bary_img), row_pixels, R, vi_sN, H * W, col_indices, static_cast<T*>(values))
  DRTK_DISPATCH(dtype, CALL, CALL)
#undef CALL
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}

extern "C" int drtk_amd_interpolation_matrix_backward(
    drtk_dtype_t dtype, const void* grad_values, const int32_t* vi, const int32_t* index_img,
    const int64_t* row_pixels, int64_t R, int64_t N, int64_t F, int64_t vi_sN, int64_t H, int64_t W,
    void* bary_grad, drtk_stream_t stream) {
  if (bad(N, F, H, W) || R < 0 || (vi_sN != 0 && vi_sN != F * 3)) return DRTK_ERR_INVALID_ARGUMENT;
  if (dtype != DRTK_F32 && dtype != DRTK_F64) return DRTK_ERR_INVALID_ARGUMENT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t es = dtype == DRTK_F32 ? 4 : 8;
  if (N * H * W > 0) {
    if (!bary_grad) return DRTK_ERR_INVALID_ARGUMENT;
    if (fill_bytes_async(bary_grad, 0, es * 3 * N * H * W, s) != DRTK_OK) return DRTK_ERR_LAUNCH; // cpu :512
  }
  if (R == 0) return DRTK_OK;
  if (!grad_values || !vi || !index_img || !row_pixels) return DRTK_ERR_INVALID_ARGUMENT;
  const dim3 grid(static_cast<unsigned>(ceil_div(R, kBlock)));
#define CALL DRTK_LAUNCH((interpolation_matrix_backward_kernel<T>), grid, dim3(kBlock), 0, s, static_cast<const T*>(grad_values), vi, index_img, row_pixels, R, vi_sN, H * W, static_cast<T*>(bary_grad))
  DRTK_DISPATCH(dtype, CALL, CALL)
#undef CALL
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}

extern "C" int drtk_amd_interpolation_normal_matrix_values(
    drtk_dtype_t dtype, const int32_t* pair_indices, const int32_t* index_img, const void* bary_img,
    int64_t N, int64_t F, int64_t pair_sN, int64_t H, int64_t W, int64_t nnz, void* values,
    drtk_stream_t stream) {
  // (the two normal-matrix kernels take the view from blockIdx.y and accumulate into ONE values array: no slicing here)
  if (bad(N, F, H, W) || N > kMaxViewsPerLaunch || nnz < 0 || (pair_sN != 0 && pair_sN != F * 9)) return DRTK_ERR_INVALID_ARGUMENT;
  if (dtype != DRTK_F32 && dtype != DRTK_F64) return DRTK_ERR_INVALID_ARGUMENT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t es = dtype == DRTK_F32 ? 4 : 8;
  if (nnz > 0) {
    if (!values) return DRTK_ERR_INVALID_ARGUMENT;
    if (fill_bytes_async(values, 0, es * nnz, s) != DRTK_OK) return DRTK_ERR_LAUNCH; // cpu :586
  }
  if (N * H * W == 0 || nnz == 0) return DRTK_OK;
  if (!pair_indices || !index_img || !bary_img) return DRTK_ERR_INVALID_ARGUMENT;
  const int tiles_x = static_cast<int>(ceil_div(W, kWave)), tiles_y = static_cast<int>(ceil_div(H, kTileRows));
  const dim3 grid(static_cast<unsigned>(int64_t(tiles_x) * tiles_y), static_cast<unsigned>(N));
#define CALL DRTK_LAUNCH((normal_matrix_values_kernel<T>), grid, dim3(kBlock), 0, s, pair_indices, index_img, static_cast<const T*>(bary_img), pair_sN, (int)H, (int)W, tiles_x, static_cast<T*>(values), xcd_strip(int64_t(tiles_x) * (16 / kTileRows)))
  DRTK_DISPATCH(dtype, CALL, CALL)
#undef CALL
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}

extern "C" int drtk_amd_interpolation_normal_matrix_values_backward(
    drtk_dtype_t dtype, const void* grad_values, const int32_t* pair_indices, const int32_t* index_img,
    const void* bary_img, int64_t N, int64_t F, int64_t pair_sN, int64_t H, int64_t W, void* bary_grad,
    drtk_stream_t stream) {
  if (bad(N, F, H, W) || N > kMaxViewsPerLaunch || (pair_sN != 0 && pair_sN != F * 9)) return DRTK_ERR_INVALID_ARGUMENT;
  if (dtype != DRTK_F32 && dtype != DRTK_F64) return DRTK_ERR_INVALID_ARGUMENT;
  if (N * H * W == 0) return DRTK_OK;
  if (!bary_grad) return DRTK_ERR_INVALID_ARGUMENT;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (F == 0) {
    // No faces: every pixel is background and its gradient is zero.  The pair table and grad_values are empty
    // then and have no storage -- a null pointer is not a missing argument here (same idiom as
    // drtk_amd_interpolation_matrix_backward above: write the output, return when there is nothing to scatter).
    const size_t es = dtype == DRTK_F32 ? 4 : 8;
    if (fill_bytes_async(bary_grad, 0, es * 3 * N * H * W, s) != DRTK_OK) return DRTK_ERR_LAUNCH;
    return DRTK_OK;
  }
  if (!grad_values || !pair_indices || !index_img || !bary_img) return DRTK_ERR_INVALID_ARGUMENT;
  const dim3 grid(static_cast<unsigned>(ceil_div(H * W, kBlock)), static_cast<unsigned>(N));
#define CALL DRTK_LAUNCH((normal_matrix_values_backward_kernel<T>), grid, dim3(kBlock), 0, s, static_cast<const T*>(grad_values), pair_indices, index_img, static_cast<const T*>(bary_img), pair_sN, H * W, static_cast<T*>(bary_grad))
  DRTK_DISPATCH(dtype, CALL, CALL)
#undef CALL
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}

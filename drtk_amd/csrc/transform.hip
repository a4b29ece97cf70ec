// transform (pinhole) -- the step in front of the hot path.
//
// Reference: drtk/transform.py:13-119 -> drtk/utils/projection.py:33-53,486-540 (pure PyTorch:
// v_cam = camrot @ (v - campos); z clamped away from 0 by 1e-8; v_pix.xy = focal @ (v_cam.xy / z) +
// princpt; v_pix.z = v_cam.z).  In eager PyTorch that is ~10 launches forward and ~25 backward per
// step (two of them batched GEMMs); here it is one kernel each way.  World-space vertices shared by
// all views ([1,V,3]) are broadcast in the kernel, and the backward sums their gradient over the
// views in registers (one thread per vertex walks the N views) -- no atomics, deterministic, and the
// tensor that later goes into the cross-GPU all-reduce comes out already reduced over local views.
#include "common.hpp"

namespace drtk_amd {
namespace {

template <typename T>
struct Cam {
  T pos[3], rot[9], focal[4], pp[2];
};

template <typename T>
__device__ __forceinline__ Cam<T> load_cam(
    const T* __restrict__ campos, const T* __restrict__ camrot, const T* __restrict__ focal,
    const T* __restrict__ princpt, int n) {
  Cam<T> c;
#pragma unroll
  for (int i = 0; i < 3; ++i) c.pos[i] = campos[n * 3 + i];
#pragma unroll
  for (int i = 0; i < 9; ++i) c.rot[i] = camrot[n * 9 + i];
#pragma unroll
  for (int i = 0; i < 4; ++i) c.focal[i] = focal[n * 4 + i];
  c.pp[0] = princpt[n * 2 + 0];
  c.pp[1] = princpt[n * 2 + 1];
  return c;
}

// projection.py:47-48 : z < 0 ? min(z, -1e-8) : max(z, 1e-8)
template <typename T>
__device__ __forceinline__ T clamp_z(T z, bool& clamped) {
  const T e = T(1e-8);
  const T zc = z < T(0) ? (z < -e ? z : -e) : (z > e ? z : e);
  clamped = zc != z;
  return zc;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void transform_kernel(
    const T* __restrict__ v, int64_t v_sN, const T* __restrict__ campos, const T* __restrict__ camrot,
    const T* __restrict__ focal, const T* __restrict__ princpt, int V, T* __restrict__ v_pix,
    T* __restrict__ v_cam_out) {
  const int n = blockIdx.y;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= V) return;
  const Cam<T> c = load_cam<T>(campos, camrot, focal, princpt, n);
  const T* p = v + int64_t(n) * v_sN + int64_t(i) * 3;
  const T dx = p[0] - c.pos[0], dy = p[1] - c.pos[1], dz = p[2] - c.pos[2];
  const T cx = c.rot[0] * dx + c.rot[1] * dy + c.rot[2] * dz;
  const T cy = c.rot[3] * dx + c.rot[4] * dy + c.rot[5] * dz;
  const T cz = c.rot[6] * dx + c.rot[7] * dy + c.rot[8] * dz;
  bool clamped;
  const T zc = clamp_z(cz, clamped);
  const T px = cx / zc, py = cy / zc;
  T* o = v_pix + (int64_t(n) * V + i) * 3;
  o[0] = c.focal[0] * px + c.focal[1] * py + c.pp[0];
  o[1] = c.focal[2] * px + c.focal[3] * py + c.pp[1];
  o[2] = cz;
  if (v_cam_out) {
    T* q = v_cam_out + (int64_t(n) * V + i) * 3;
    q[0] = cx, q[1] = cy, q[2] = cz;
  }
}

// grad_v[n or 0, i, :] ; SHARED: v is [1,V,3] and the gradient is summed over the N views here.
template <typename T, bool SHARED>
__global__ __launch_bounds__(kBlock) void transform_backward_kernel(
    const T* __restrict__ v, const T* __restrict__ campos, const T* __restrict__ camrot,
    const T* __restrict__ focal, const T* __restrict__ princpt, const T* __restrict__ grad_v_pix,
    int N, int V, T* __restrict__ grad_v) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= V) return;
  const int n_begin = SHARED ? 0 : blockIdx.y, n_end = SHARED ? N : blockIdx.y + 1;
  T ax = T(0), ay = T(0), az = T(0);
  for (int n = n_begin; n < n_end; ++n) {
    const Cam<T> c = load_cam<T>(campos, camrot, focal, princpt, n);
    const T* p = v + (SHARED ? int64_t(0) : int64_t(n) * V * 3) + int64_t(i) * 3;
    const T dx = p[0] - c.pos[0], dy = p[1] - c.pos[1], dz = p[2] - c.pos[2];
    const T cx = c.rot[0] * dx + c.rot[1] * dy + c.rot[2] * dz;
    const T cy = c.rot[3] * dx + c.rot[4] * dy + c.rot[5] * dz;
    const T cz = c.rot[6] * dx + c.rot[7] * dy + c.rot[8] * dz;
    bool clamped;
    const T zc = clamp_z(cz, clamped);
    const T* g = grad_v_pix + (int64_t(n) * V + i) * 3;
    // v_pix.xy = focal @ proj + pp  ->  d proj = focal^T g.xy
    const T gpx = c.focal[0] * g[0] + c.focal[2] * g[1];
    const T gpy = c.focal[1] * g[0] + c.focal[3] * g[1];
    const T gcx = gpx / zc, gcy = gpy / zc;
    const T gzc = -(gpx * cx + gpy * cy) / (zc * zc);
    const T gcz = (clamped ? T(0) : gzc) + g[2];
    // v_cam = R (v - campos)  ->  d v = R^T d v_cam
    ax += c.rot[0] * gcx + c.rot[3] * gcy + c.rot[6] * gcz;
    ay += c.rot[1] * gcx + c.rot[4] * gcy + c.rot[7] * gcz;
    az += c.rot[2] * gcx + c.rot[5] * gcy + c.rot[8] * gcz;
  }
  T* o = grad_v + (SHARED ? int64_t(0) : int64_t(blockIdx.y) * V * 3) + int64_t(i) * 3;
  o[0] = ax, o[1] = ay, o[2] = az;
}

template <typename T>
int transform_impl(const T* v, int64_t v_sN, const T* campos, const T* camrot, const T* focal, const T* princpt,
                   int64_t N, int64_t V, T* v_pix, T* v_cam, hipStream_t stream) {
  if (N * V == 0) return DRTK_OK;
  const dim3 grid(static_cast<unsigned>(ceil_div(V, kBlock)), static_cast<unsigned>(N));
  DRTK_LAUNCH((transform_kernel<T>), grid, dim3(kBlock), 0, stream, v, v_sN, campos, camrot, focal, princpt, (int)V, v_pix, v_cam);
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}

template <typename T>
int transform_backward_impl(const T* v, int64_t v_sN, const T* campos, const T* camrot, const T* focal,
                            const T* princpt, const T* grad_v_pix, int64_t N, int64_t V, T* grad_v, hipStream_t stream) {
  if (V == 0 || N == 0) return DRTK_OK;
  if (v_sN == 0) {
    const dim3 grid(static_cast<unsigned>(ceil_div(V, kBlock)), 1);
    DRTK_LAUNCH((transform_backward_kernel<T, true>), grid, dim3(kBlock), 0, stream, v, campos, camrot, focal, princpt, grad_v_pix, (int)N, (int)V, grad_v);
  } else {
    const dim3 grid(static_cast<unsigned>(ceil_div(V, kBlock)), static_cast<unsigned>(N));
    DRTK_LAUNCH((transform_backward_kernel<T, false>), grid, dim3(kBlock), 0, stream, v, campos, camrot, focal, princpt, grad_v_pix, (int)N, (int)V, grad_v);
  }
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}

} // namespace
} // namespace drtk_amd

using namespace drtk_amd;

extern "C" int drtk_amd_transform_pinhole(
    drtk_dtype_t dtype, const void* v, int64_t v_sN, const void* campos, const void* camrot,
    const void* focal, const void* princpt, int64_t N, int64_t V, void* v_pix, void* v_cam,
    drtk_stream_t stream) {
  if (N < 0 || V < 0 || V >= (int64_t(1) << 31) || (v_sN != 0 && v_sN != V * 3)) return DRTK_ERR_INVALID_ARGUMENT;
  if (N * V > 0 && (!v || !campos || !camrot || !focal || !princpt || !v_pix)) return DRTK_ERR_INVALID_ARGUMENT;
  if (dtype != DRTK_F32 && dtype != DRTK_F64) return DRTK_ERR_INVALID_ARGUMENT;
  const size_t es = dtype_size(dtype);
  DRTK_FOR_VIEW_SLICES(N, n0, n, drtk_amd_transform_pinhole(
      dtype, advance(v, n0 * v_sN, es), v_sN, advance(campos, n0 * 3, es), advance(camrot, n0 * 9, es), advance(focal, n0 * 4, es),
      advance(princpt, n0 * 2, es), n, V, advance(v_pix, n0 * V * 3, es), advance(v_cam, n0 * V * 3, es), stream))
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case DRTK_F32:
      return transform_impl<float>(static_cast<const float*>(v), v_sN, static_cast<const float*>(campos), static_cast<const float*>(camrot), static_cast<const float*>(focal), static_cast<const float*>(princpt), N, V, static_cast<float*>(v_pix), static_cast<float*>(v_cam), s);
    case DRTK_F64:
      return transform_impl<double>(static_cast<const double*>(v), v_sN, static_cast<const double*>(campos), static_cast<const double*>(camrot), static_cast<const double*>(focal), static_cast<const double*>(princpt), N, V, static_cast<double*>(v_pix), static_cast<double*>(v_cam), s);
    default:
      return DRTK_ERR_INVALID_ARGUMENT;
  }
}

extern "C" int drtk_amd_transform_pinhole_backward(
    drtk_dtype_t dtype, const void* v, int64_t v_sN, const void* campos, const void* camrot,
    const void* focal, const void* princpt, const void* grad_v_pix, int64_t N, int64_t V,
    void* grad_v, drtk_stream_t stream) {
  if (N < 0 || V < 0 || V >= (int64_t(1) << 31) || (v_sN != 0 && v_sN != V * 3)) return DRTK_ERR_INVALID_ARGUMENT;
  if (N * V > 0 && (!v || !campos || !camrot || !focal || !princpt || !grad_v_pix || !grad_v)) return DRTK_ERR_INVALID_ARGUMENT;
  if (dtype != DRTK_F32 && dtype != DRTK_F64) return DRTK_ERR_INVALID_ARGUMENT;
  if (v_sN != 0) { // per-view vertices: views are independent (shared vertices: ONE launch sums the views in-kernel, any N)
    const size_t es = dtype_size(dtype);
    DRTK_FOR_VIEW_SLICES(N, n0, n, drtk_amd_transform_pinhole_backward(
        dtype, advance(v, n0 * v_sN, es), v_sN, advance(campos, n0 * 3, es), advance(camrot, n0 * 9, es), advance(focal, n0 * 4, es),
        advance(princpt, n0 * 2, es), advance(grad_v_pix, n0 * V * 3, es), n, V, advance(grad_v, n0 * V * 3, es), stream))
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (dtype) {
    case DRTK_F32:
      return transform_backward_impl<float>(static_cast<const float*>(v), v_sN, static_cast<const float*>(campos), static_cast<const float*>(camrot), static_cast<const float*>(focal), static_cast<const float*>(princpt), static_cast<const float*>(grad_v_pix), N, V, static_cast<float*>(grad_v), s);
    case DRTK_F64:
      return transform_backward_impl<double>(static_cast<const double*>(v), v_sN, static_cast<const double*>(campos), static_cast<const double*>(camrot), static_cast<const double*>(focal), static_cast<const double*>(princpt), static_cast<const double*>(grad_v_pix), N, V, static_cast<double*>(grad_v), s);
    default:
      return DRTK_ERR_INVALID_ARGUMENT;
  }
}

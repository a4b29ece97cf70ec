// screen_space_uv_derivative -- per-pixel Jacobian of the texture coordinates wrt the pixel position,
// [[du/dx, dv/dx], [du/dy, dv/dy]], the `vt_dxdy_img` input of mipmap_grid_sample.
//
// Reference: drtk/screen_space_uv_derivative.py:15-80, a PyTorch composite over
//   face_dpdt            drtk/utils/geometry.py:18-82     (dp/dt)^T = ((dt/db)^T)^-1 (dp/db)^T per face
//   interpolate x2       of the per-face Jacobian (6 channels) and of the face's vertex positions
//   project_points_grad  drtk/utils/projection.py:650-709  pinhole Jacobian applied to both rows
//   linalg.inv_ex + mask
// which materialises ~40 floats per pixel in intermediate images.  Here it is ONE kernel: lane = pixel,
// reads index (4 B) + bary (12 B), gathers the triangle's 3 positions and 3 uvs, writes 16 B.
// The arithmetic follows the composite's steps (including the x*b0 + x*b1 + x*b2 form that `interpolate` gives a
// per-face constant) with ONE algebraic change: the composite inverts the UV edge matrix A, pushes A^-1 dpdb through
// the (linear) interpolation and pinhole Jacobian to J = A^-1 G, and inverts again; J^-1 = G^-1 A is evaluated here
// directly.  Same function, one 2x2 inverse (closed-form adjugate, compensated determinant) instead of two, and no
// rounding error amplified by cond(A): on UV slivers the two-inverse form -- in f32, the composite's LU as well as a
// closed form -- is 10-30x further from the f64 result than this one (tests/fuzz_next_ops.py, seeds 450324, 760460).
// A face with zero UV area, for which the reference raises, gets the finite limit.
// Pinhole cameras only, like project_points_grad itself (distortion raises NotImplementedError there).
#include "common.hpp"

namespace drtk_amd {
namespace {

// The per-pixel arithmetic: p[k] / t[k] = positions / uvs of the pixel's triangle, b* its barycentrics, R / cp / K the view's
// camera.  Shared by the two kernels below (same operations, same order).
template <typename T>
__device__ __forceinline__ void uv_jacobian(
    const T (&p)[3][3], const T (&t)[3][2], T b0, T b1, T b2, const T* __restrict__ R, const T* __restrict__ cp,
    const T* __restrict__ K, T& o00, T& o01, T& o10, T& o11) {
  // face_dpdt's two edge matrices: A = dtdb_t = [[t1-t0],[t2-t0]] (rows), dpdb_t = [[p1-p0],[p2-p0]]
  const T a = t[1][0] - t[0][0], b = t[1][1] - t[0][1];
  const T c = t[2][0] - t[0][0], d = t[2][1] - t[0][1];
  // the interpolate() of the positions: sum b_k p_k
  T P[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) P[j] = p[0][j] * b0 + p[1][j] * b1 + p[2][j] * b2;
  // project_points_grad (pinhole)
  const T dx = P[0] - cp[0], dy = P[1] - cp[1], dz = P[2] - cp[2];
  const T cx = R[0] * dx + R[1] * dy + R[2] * dz;
  const T cy = R[3] * dx + R[4] * dy + R[5] * dz;
  T z = R[6] * dx + R[7] * dy + R[8] * dz;
  const T e = T(1e-8);
  z = z < T(0) ? (z < -e ? z : -e) : (z > e ? z : e);
  const T rzz = T(1) / (z * z);
  // G[k][j] = d p_pix[j] along the position edge k (p_{k+1} - p_0), carrying the interpolate() of a per-face constant
  // (x*b0 + x*b1 + x*b2).  The composite's J = A^-1 G (every step after dp/dt = A^-1 dpdb_t is linear in its rows), so
  // its result J^-1 = G^-1 A: the UV edge matrix is never inverted.
  T G[2][2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    T g[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const T ek = p[k + 1][j] - p[0][j];
      g[j] = ek * b0 + ek * b1 + ek * b2;
    }
    const T gx = R[0] * g[0] + R[1] * g[1] + R[2] * g[2];
    const T gy = R[3] * g[0] + R[4] * g[1] + R[5] * g[2];
    const T gz = R[6] * g[0] + R[7] * g[1] + R[8] * g[2];
    const T px = (gx * z - cx * gz) * rzz;
    const T py = (gy * z - cy * gz) * rzz;
    G[k][0] = K[0] * px + K[1] * py;
    G[k][1] = K[2] * px + K[3] * py;
  }
  // det G by Kahan's difference of products (the rounding error of one product recovered with an fma): 1.5 ulp
  const T w = G[0][1] * G[1][0];
  const T dj = fma(G[0][0], G[1][1], -w) + fma(-G[0][1], G[1][0], w);
  // one reciprocal instead of four IEEE divisions (a float division is a ten-instruction sequence); below ~1e-30
  // (1e-200 in double) the reciprocal itself overflows where the quotients may still be finite: true divisions
  // there -- a wave takes that path only if one of its faces is that degenerate
  const T kTinyDet = sizeof(T) == 4 ? T(1e-30) : T(1e-200);
  const T rdj = T(1) / dj;
  // vt_dxdy = G^-1 A: [i][j] = d t[j] / d p_pix[i]
  const T n00 = fma(G[1][1], a, -(G[0][1] * c)), n01 = fma(G[1][1], b, -(G[0][1] * d));
  const T n10 = fma(G[0][0], c, -(G[1][0] * a)), n11 = fma(G[0][0], d, -(G[1][0] * b));
  o00 = n00 * rdj, o01 = n01 * rdj, o10 = n10 * rdj, o11 = n11 * rdj;
  if (fabs(dj) < kTinyDet) o00 = n00 / dj, o01 = n01 / dj, o10 = n10 / dj, o11 = n11 / dj;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void uv_derivative_kernel(
    const T* __restrict__ v, int64_t v_sN, const T* __restrict__ vt, int64_t vt_sN,
    const int32_t* __restrict__ vi, const int32_t* __restrict__ vti, const int32_t* __restrict__ index_img,
    const T* __restrict__ bary_img, const uint8_t* __restrict__ mask, const T* __restrict__ campos,
    const T* __restrict__ camrot, const T* __restrict__ focal, int64_t HW, T* __restrict__ out, int strip) {
  const int n = blockIdx.y;
  const int64_t pix = int64_t(tile_index(strip)) * kBlock + threadIdx.x;
  if (pix >= HW) return;
  const int32_t tr = index_img[int64_t(n) * HW + pix];
  T o00 = T(0), o01 = T(0), o10 = T(0), o11 = T(0);
  const bool keep = tr >= 0 && (mask == nullptr || mask[int64_t(n) * HW + pix] != 0);
  if (keep) {
    const T* bp = bary_img + int64_t(n) * 3 * HW + pix;
    const T b0 = bp[0], b1 = bp[HW], b2 = bp[2 * HW];
    const int32_t* f = vi + int64_t(tr) * 3;
    const int32_t* ft = vti + int64_t(tr) * 3;
    const T* v_n = v + int64_t(n) * v_sN;
    const T* vt_n = vt + int64_t(n) * vt_sN;
    T p[3][3], t[3][2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const T* q = v_n + int64_t(f[k]) * 3;
      p[k][0] = q[0], p[k][1] = q[1], p[k][2] = q[2];
      const T* r = vt_n + int64_t(ft[k]) * 2;
      t[k][0] = r[0], t[k][1] = r[1];
    }
    uv_jacobian<T>(p, t, b0, b1, b2, camrot + int64_t(n) * 9, campos + int64_t(n) * 3, focal + int64_t(n) * 4, o00, o01, o10, o11);
  }
  T* o = out + (int64_t(n) * HW + pix) * 4;
  if constexpr (sizeof(T) == 4) {
    *reinterpret_cast<float4*>(o) = make_float4(o00, o01, o10, o11);
  } else {
    o[0] = o00, o[1] = o01, o[2] = o10, o[3] = o11;
  }
}

// (Round 4, measured and not kept: four consecutive pixels per lane for float -- index / barycentrics as element-aligned
// 16-byte accesses, the face ids of all four pixels as one batch, then their 36 + 24 vertex / uv values as one batch, 64
// contiguous bytes stored per lane.  The one-pixel kernel's waves wait 77 % of their cycles on that chain of three dependent
// round trips, at 8 waves per SIMD; the batched one has 114 VGPRs, 4 waves per SIMD, and is SLOWER: 0.424 vs 0.343 ms
// through the C ABI on the textured benchmark's 2 x 4096^2 pixels -- twice the loads per wave in flight do not make up for
// half the waves.  Parity was green: fuzz_next_ops 3000.)

} // namespace
} // namespace drtk_amd

using namespace drtk_amd;

extern "C" int drtk_amd_screen_space_uv_derivative(
    drtk_dtype_t dtype, const void* v, int64_t v_sN, const void* vt, int64_t vt_sN, const int32_t* vi,
    const int32_t* vti, const int32_t* index_img, const void* bary_img, const uint8_t* mask, const void* campos,
    const void* camrot, const void* focal, int64_t N, int64_t V, int64_t T_, int64_t F, int64_t H, int64_t W,
    void* out, drtk_stream_t stream) {
  if (N < 0 || V < 0 || T_ < 0 || F < 0 || H < 0 || W < 0 || H * W >= (int64_t(1) << 31) ||
      (v_sN != 0 && v_sN != V * 3) || (vt_sN != 0 && vt_sN != T_ * 2) || (dtype != DRTK_F32 && dtype != DRTK_F64))
    return DRTK_ERR_INVALID_ARGUMENT;
  if (N * H * W == 0) return DRTK_OK;
  if (!index_img || !bary_img || !out || !campos || !camrot || !focal || (F > 0 && (!vi || !vti || !v || !vt)))
    return DRTK_ERR_INVALID_ARGUMENT;
  {
    const size_t es = dtype_size(dtype);
    DRTK_FOR_VIEW_SLICES(N, n0, n, drtk_amd_screen_space_uv_derivative(
        dtype, advance(v, n0 * v_sN, es), v_sN, advance(vt, n0 * vt_sN, es), vt_sN, vi, vti, advance_typed(index_img, n0 * H * W),
        advance(bary_img, n0 * 3 * H * W, es), advance_typed(mask, n0 * H * W), advance(campos, n0 * 3, es), advance(camrot, n0 * 9, es),
        advance(focal, n0 * 4, es), n, V, T_, F, H, W, advance(out, n0 * 4 * H * W, es), stream))
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid(static_cast<unsigned>(ceil_div(H * W, kBlock)), static_cast<unsigned>(N));
  if (dtype == DRTK_F32) {
    DRTK_LAUNCH(
        (uv_derivative_kernel<float>), grid, dim3(kBlock), 0, s, static_cast<const float*>(v), v_sN,
        static_cast<const float*>(vt), vt_sN, vi, vti, index_img, static_cast<const float*>(bary_img), mask,
        static_cast<const float*>(campos), static_cast<const float*>(camrot), static_cast<const float*>(focal), H * W,
        static_cast<float*>(out), xcd_strip(ceil_div(16 * W, kBlock)));
  } else {
    DRTK_LAUNCH(
        (uv_derivative_kernel<double>), grid, dim3(kBlock), 0, s, static_cast<const double*>(v), v_sN,
        static_cast<const double*>(vt), vt_sN, vi, vti, index_img, static_cast<const double*>(bary_img), mask,
        static_cast<const double*>(campos), static_cast<const double*>(camrot), static_cast<const double*>(focal),
        H * W, static_cast<double*>(out), xcd_strip(ceil_div(16 * W, kBlock)));
  }
  DRTK_RETURN_IF_LAUNCH_FAILED();
  return DRTK_OK;
}
